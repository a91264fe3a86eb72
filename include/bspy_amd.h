/*
 * bspy_amd.h - C ABI of libbspy_amd.so: MI355X (gfx950) batched B-spline evaluation.
 *
 * The reference (ericbrec/BSpy 5.0.1) is pure Python and has no FFI of its own; its
 * boundary for this path is the Python API.  Each entry point
 * below names the reference interface it replaces (paths relative to the reference
 * checkout).  The reference-side binding a maintainer would add is the ctypes stub
 * shown in INTEGRATION.md; bspy_amd/_native.py is that stub in full.
 *
 * Conventions
 *  - plain C types only; no exceptions, no Python objects, no torch types.
 *  - every function returns a bsk_status (0 = ok).  bsk_last_error() returns a
 *    thread-local, human readable message for the last non-zero status.
 *  - dtype: BSK_F32 or BSK_F64.  It is the arithmetic type of the whole call:
 *    knots, coefficients, parameters and results all have it (the Python layer
 *    promotes mixed inputs, see bspy_amd/spline.py).
 *  - parameter points are SoA: one pointer per independent variable, each to n
 *    contiguous values.  Results are SoA: out[d * n + i] (evaluate/derivative),
 *    out[(d * nInd + j) * n + i] (jacobian).
 *  - mem: BSK_HOST buffers are copied to/from the device by the library (PCIe in the
 *    call); BSK_DEVICE buffers are device pointers on the spline's device and the call
 *    only enqueues work on `stream` (a hipStream_t, NULL = default stream).
 *  - the caller owns every buffer it passes; the library owns the device tables
 *    behind a bsk_spline handle.  Handles are not thread safe; distinct handles are
 *    independent.
 *  - a handle also owns scratch workspaces (jacobian rows of a non-fused normal, derivative passes of a
 *    non-fused curvature, grid basis rows, the cell-order pipeline's records).  They are shared by every call
 *    on the handle, so ONE handle is used from ONE stream at a time (create a second handle for a second
 *    stream).  A workspace grows (free + allocate) only when a call needs more than any call before it; such
 *    growth cannot be recorded by a stream capture and returns BSK_ERR_INVALID with a message saying so - run
 *    the call once outside the capture.  With that, every BSK_DEVICE call only enqueues kernels on `stream`
 *    (the cell-order pipeline of large tables sizes its workspace per batch and declines under capture: the
 *    call then runs the gather kernel).
 */
#ifndef BSPY_AMD_H
#define BSPY_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BSK_VERSION 1

#define BSK_MAX_NIND 8     /* independent variables per spline */
#define BSK_MAX_ORDER 16   /* polynomial order (degree + 1) per variable */

typedef enum { BSK_F32 = 0, BSK_F64 = 1 } bsk_dtype;
typedef enum { BSK_HOST = 0, BSK_DEVICE = 1 } bsk_mem;

typedef enum {
    BSK_OK = 0,
    BSK_ERR_INVALID = 1,      /* bad argument (message says which) */
    BSK_ERR_DOMAIN = 2,       /* a parameter lies outside the spline's domain; *first_bad = its flat index */
    BSK_ERR_HIP = 3,          /* HIP runtime failure */
    BSK_ERR_NO_DEVICE = 4,    /* no usable gfx950 device */
    BSK_ERR_UNSUPPORTED = 5   /* nInd / order beyond BSK_MAX_* */
} bsk_status;

typedef struct bsk_spline_s *bsk_spline;

/* Library / device information. */
int bsk_version(void);
const char *bsk_last_error(void);
bsk_status bsk_device_count(int *count);

/*
 * Spline handle = the device-resident tables of one reference `Spline` object.
 * Replaces: Spline.__init__ storage, bspy/spline.py:46-76 (validation stays in Python).
 *   knots[iv]  : order[iv] + nCoef[iv] values, non-decreasing
 *   coefs      : C-contiguous (nDep, nCoef[0], ..., nCoef[nInd-1])
 * The data is copied; bsk_spline_update re-uploads after the Python object's
 * knots/coefs were mutated (same shapes).
 */
bsk_status bsk_spline_create(bsk_dtype dtype, int device, int nInd, int nDep,
                             const int *order, const int *nCoef,
                             const void *const *knots, const void *coefs,
                             bsk_spline *out);
bsk_status bsk_spline_update(bsk_spline s, const void *const *knots, const void *coefs);
bsk_status bsk_spline_destroy(bsk_spline s);

/*
 * Batched evaluate / derivative.
 * Replaces: Spline.evaluate / Spline.derivative batched through np.frompyfunc,
 *   bspy/spline.py:936-949 and :757-770, i.e. one call of
 *   bspy/_spline_evaluation.py:140-164 (evaluate) / :109-133 (derivative) per point,
 *   including its span search and de Boor recursion (:4-27).
 *   wrt        : nInd derivative orders, or NULL for plain evaluation
 *   uvw[iv]    : n parameter values of variable iv
 *   out        : nDep * n results, out[d * n + i]
 *   first_bad  : (may be NULL) receives -1, or the index of the first point outside
 *                the inclusive domain [knots[order-1], knots[nCoef]] (status
 *                BSK_ERR_DOMAIN; the reference raises ValueError for that point,
 *                _spline_evaluation.py:148-152).  NaN parameters pass, as there.
 * With mem == BSK_DEVICE the call is asynchronous and never reports BSK_ERR_DOMAIN
 * itself: use bsk_domain_status() after the work was enqueued.
 */
bsk_status bsk_evaluate(bsk_spline s, const int *wrt, const void *const *uvw, int64_t n,
                        bsk_mem mem, void *out, void *stream, int64_t *first_bad);

/*
 * Batched jacobian: all first partial derivatives in one pass.
 * Replaces: Spline.jacobian / tangent_space, bspy/_spline_evaluation.py:205-213
 *   (nInd calls of derivative per point; single point only in the reference).
 *   out        : nDep * nInd * n results, out[(d * nInd + j) * n + i]
 */
bsk_status bsk_jacobian(bsk_spline s, const void *const *uvw, int64_t n,
                        bsk_mem mem, void *out, void *stream, int64_t *first_bad);

/*
 * Batched normal (next row of the scope table).
 * Replaces: Spline.normal, bspy/spline.py:1648-1682 -> bspy/_spline_evaluation.py:215-246
 *   (single point in the reference).  Needs |nInd - nDep| == 1 and max(nInd, nDep) <= 4.
 *   normalize  : unit length (the reference's default) or area-scaled cofactor vector
 *   negate     : the reference's metadata["negateNormal"]
 *   out        : max(nInd, nDep) * n values, out[i * n + p]
 */
bsk_status bsk_normal(bsk_spline s, const void *const *uvw, int64_t n, bsk_mem mem, int normalize, int negate,
                      void *out, void *stream, int64_t *first_bad);

/*
 * Batched curvature (next row).
 * Replaces: Spline.curvature, bspy/_spline_evaluation.py:80-107 (single point there).
 *   curves (nInd 1, nDep >= 2): signed curvature in 2-D, unsigned otherwise;
 *   surfaces in 3-D (nInd 2, nDep 3): Gaussian curvature.
 *   out : n values
 */
bsk_status bsk_curvature(bsk_spline s, const void *const *uvw, int64_t n, bsk_mem mem, void *out, void *stream,
                         int64_t *first_bad);

/*
 * Tensor-product grid evaluation: parameters are the outer product of per-variable
 * vectors (the reference's broadcast call s(u[:, None], v[None, :]), bspy/spline.py:941-945).
 *   grid[iv]   : ngrid[iv] values of variable iv
 *   out        : nDep * prod(ngrid) results, C order (d, i0, i1, ...)
 *   first_bad  : flat index into the broadcast shape (i0, i1, ...) of the first offender
 */
bsk_status bsk_evaluate_grid(bsk_spline s, const int *wrt, const void *const *grid, const int64_t *ngrid,
                             bsk_mem mem, void *out, void *stream, int64_t *first_bad);

/*
 * Tessellation of a BATCH of surface patches in 3-D on one parameter grid: positions and, optionally,
 * normals of every patch from one launch (next row).
 * Replaces: per patch the broadcast call s(u[:, None], v[None, :]) (bspy/spline.py:941-945) and
 *   Spline.normal at every grid point (bspy/spline.py:1648-1682 -> bspy/_spline_evaluation.py:215-246);
 *   it is the compute-side counterpart of the reference's GLSL tessellation shaders
 *   (bspy/splineOpenGLFrame.py:671-714), e.g. for the 32 patches of examples/teapot.py.
 *   splines    : count handles with nInd 2, nDep 3 that share dtype, device, orders, nCoef and knots
 *   grid[iv]   : ngrid[iv] values of variable iv (the grid is shared by all patches)
 *   positions  : count * 3 * n0 * n1 values, positions[((p * 3 + d) * n0 + i0) * n1 + i1]
 *   normals    : NULL, or the same shape: cross product of the two partial derivatives
 *                (normalize / negate as in bsk_normal)
 *   first_bad  : flat index i0 * n1 + i1 of the first grid point outside the domain
 */
bsk_status bsk_tessellate(const bsk_spline *splines, int count, const void *const *grid, const int64_t *ngrid,
                          bsk_mem mem, int normalize, int negate, void *positions, void *normals, void *stream,
                          int64_t *first_bad);

/*
 * Synchronise `stream` and report whether any BSK_DEVICE call on this handle since the
 * last bsk_domain_status() met an out-of-domain parameter (*first_bad = smallest such
 * index, else -1).  Resets the record.
 */
bsk_status bsk_domain_status(bsk_spline s, void *stream, int64_t *first_bad);

/*
 * Batched B-spline basis values (no spline object).
 * Replaces: Spline.bspline_values, bspy/spline.py:207-252 -> bspy/_spline_evaluation.py:4-27.
 *   knot_in    : NULL to search the span of every u (reference: knot=None), else n
 *                explicit span indices
 *   ix_out     : n span indices ("rightmost knot of the segment")
 *   basis_out  : n * order values, basis_out[i * order + k] multiplies coefficient
 *                ix_out[i] - order + k
 * Host buffers only (this is the low-rate public helper; the evaluation kernels have
 * the same recursion fused in).
 */
bsk_status bsk_bspline_values(bsk_dtype dtype, int device, const void *knots, int nknots, int order,
                              const void *u, int64_t n, int derivative_order, int taylor_coefs,
                              const int32_t *knot_in, int32_t *ix_out, void *basis_out);

/*
 * Multi-device evaluation in ONE process (north_star: "sharding the evaluation-point batch with an RCCL
 * all-gather of results"; the reference has no multi-device code).  A bsk_multi holds one
 * replica of the spline tables and one stream per device.  A call cuts the n points into contiguous shards
 * of ceil(n / ndev) points (bsk_multi_shard_plan: start[d] .. start[d + 1]), evaluates every shard on its
 * device through bsk_evaluate / bsk_jacobian, and exchanges results only on request:
 *   mem == BSK_HOST    uvw[iv] = n host values; out[0] = host (rows, n); every device copies its shard over
 *                      its own PCIe link; `gather` is ignored (the host holds the whole result).
 *   mem == BSK_DEVICE  uvw[d * nInd + iv] = device d's shard of variable iv (on device d);
 *     gather == 0      out[d] = device d's compact (rows, shard) block: no collective at all.
 *     gather != 0      out[d] = (rows, ndev * chunk) on device d, chunk = ceil(n / ndev): every device
 *                      receives every shard - one grouped RCCL exchange (ncclGroupStart, one ncclAllGather
 *                      per device and row, ncclGroupEnd); columns >= n of the last chunk are padding.
 *                      librccl.so is opened on first use; BSK_ERR_UNSUPPORTED when it cannot be loaded.
 *   rows = nDep (evaluate / derivative) or nDep * nInd (jacobian).
 *   first_bad = global index of the first out-of-domain point (BSK_ERR_DOMAIN), else -1.  The call returns
 *   when every device has finished.
 */
typedef struct bsk_multi_s *bsk_multi;
bsk_status bsk_multi_create(bsk_dtype dtype, int ndev, const int *devices /* NULL = 0 .. ndev-1 */, int nInd, int nDep,
                            const int *order, const int *nCoef, const void *const *knots, const void *coefs,
                            bsk_multi *out);
bsk_status bsk_multi_destroy(bsk_multi m);
bsk_status bsk_multi_shard_plan(bsk_multi m, int64_t n, int64_t *start /* ndev + 1 entries */);
bsk_status bsk_multi_stream(bsk_multi m, int d, void **stream /* hipStream_t of device slot d */);
bsk_status bsk_multi_evaluate(bsk_multi m, const int *wrt, const void *const *uvw, int64_t n, bsk_mem mem,
                              void *const *out, int gather, int64_t *first_bad);
bsk_status bsk_multi_jacobian(bsk_multi m, const void *const *uvw, int64_t n, bsk_mem mem, void *const *out,
                              int gather, int64_t *first_bad);

/*
 * Diagnostics (no reference counterpart; used by bench.py, tools/ and the tests).
 *   bsk_last_kernel : family name of the kernel the most recent point call on this handle launched
 *                     ("eval_uni", "eval_rowrot", "eval_stream", "cell-order pipeline (...)", ...), so
 *                     that measurements and tests name the kernel that actually ran.
 */
const char *bsk_last_kernel(bsk_spline s);

/*
 * BSK_INTERNAL: measurement hooks of this repository's bench.py / tools/.  They are exported by the library but are
 * NOT part of the product ABI (no reference counterpart, no stability promise): a binding of the reference never
 * needs them, and they are only declared when the including file asks for them.
 *   bsk_debug_probe       : memory-side floor of the LDS-resident kernels' launch geometry: streams two fp64
 *                           parameter arrays in and three result arrays out with trivial arithmetic
 *                           (mode 0: 8 B per lane per access, mode 1: 16 B), `blocks_per_cu` workgroups of
 *                           `threads` lanes per CU with `lds_bytes` of LDS allocated.  tools/probe_stream.py.
 *   bsk_debug_stage_times : per-kernel durations of the cell-order pipeline (HIP events between its kernels on the
 *                           call's stream).  enable != 0 records the FOLLOWING calls on this handle; a call with
 *                           ms / names / count returns the stages of the last recorded call (names are static
 *                           strings, at most `cap` entries).
 */
#ifdef BSK_INTERNAL
bsk_status bsk_debug_probe(bsk_spline s, int mode, int blocks_per_cu, int threads, int64_t lds_bytes,
                           const void *u, const void *v, int64_t n, void *out, void *stream);
bsk_status bsk_debug_stage_times(bsk_spline s, int enable, float *ms, const char **names, int cap, int *count);
#endif

#ifdef __cplusplus
}
#endif
#endif /* BSPY_AMD_H */
