// Host-side declarations shared by the translation units of libbspy_amd.so: error reporting, the
// spline handle, and small launch helpers.  (The kernels of the large-table paths - control-point
// major gather and the cell-order pipeline - are two thirds of the device code; they are compiled
// in their own translation unit, bsk_gather_tu.hip, in parallel with bsk_api.hip.)
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <string>
#include <tuple>
#include <vector>

#include "bsk_common.hpp"
#include "bsk_tile.hpp"      // TileDesc

using namespace bsk;

// ------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------
inline thread_local std::string g_err;

inline bsk_status fail(bsk_status st, const std::string &msg)
{
    g_err = msg;
    return st;
}

#define HIPCHK(expr)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(BSK_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));      \
    } while (0)

// ------------------------------------------------------------------------------------
// handle
// ------------------------------------------------------------------------------------
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        hipError_t e = hipMalloc(&p, bytes);
        if (e == hipSuccess) cap = bytes;
        return e;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

struct bsk_spline_s {
    bsk_dtype dtype;
    int device;
    int nInd, nDep;
    int order[MAXI], ncoef[MAXI];
    bool same_order;         // every variable has the same order
    bool aos_small = false;  // coef_aos exists although the table fits LDS: mixed-order surfaces run eval_slab2 in one pass
    size_t esize;
    Desc<float> d32;
    Desc<double> d64;
    TileDesc<float> t32;
    TileDesc<double> t64;
    void *tab = nullptr;     // device axis table
    void *coef = nullptr;    // device coefficients
    void *coef_aos = nullptr;  // control-point-major copy (tables too large for LDS, nDep <= 4)
    unsigned *lut = nullptr; // device span-search bucket tables
    int variant = 0;         // BSK_VARIANT pins the kernel family: 0 auto, 1 eval_fixed, 4 eval_stream, 9 eval_rowrot, 7 no cell-order evaluation
    unsigned long long *bad = nullptr;  // device out-of-domain record
    int num_cu = 256;
    size_t lds_max = 160 * 1024;
    DevBuf in_ws, out_ws, aux_ws;       // staging for BSK_HOST calls and grid tables
    DevBuf bin_ws;                      // cell-order evaluation (bsk_binned.hpp)
    unsigned *ticket = nullptr;         // zero-initialised counters of the cell-order pipeline's "last workgroup" hand-overs
    bool bin_reuse = false;             // the batch in bin_ws is already sorted (later derivative passes of a jacobian)
    DevBuf curv_ws;                     // derivative passes of the non-fused curvature
    std::vector<unsigned char> tab_host; // host copy of the axis table (bsk_tessellate compares knots of a batch)
    void *pin = nullptr;                // pinned, device-mapped host buffer of the small-call path (run_small)
    size_t pin_cap = 0;
    struct HostPipe *pipe = nullptr;    // pinned staging, streams, events of the large BSK_HOST path (bsk_api.hip)
    // uniform-knot surface path (bsk_uniform.hpp): prebuilt LDS image (domain knots + unclamped coefficients)
    bool uni = false;
    UniDesc<float> u32;
    UniDesc<double> u64;
    DevBuf uni_img;
    // the same front end for curves, surfaces of other orders and volumes (eval_stream_uni / jac_stream_uni): shares uni_img
    bool uniN = false;
    UniDescN<float> un32;
    UniDescN<double> un64;
    const char *last_kernel = "";       // family of the most recent point-kernel launch (bsk_last_kernel)
    // measurement hook (bsk_debug_stage_times): events between the kernels of the cell-order pipeline
    static constexpr int MAX_STAGES = 12;
    bool stage_timing = false;
    hipEvent_t stage_ev[MAX_STAGES] = {};
    const char *stage_name[MAX_STAGES] = {};
    int stage_count = 0;
};

// bsk_debug_stage_times: an event behind every kernel of a multi-kernel pipeline (only while the hook is enabled)
inline void stage_mark(bsk_spline s, hipStream_t st, const char *name, bool first = false)
{
    if (!s->stage_timing) return;
    if (first) s->stage_count = 0;
    if (s->stage_count >= bsk_spline_s::MAX_STAGES) return;
    hipEvent_t &e = s->stage_ev[s->stage_count];
    if (!e && hipEventCreateWithFlags(&e, hipEventDefault) != hipSuccess) { e = nullptr; return; }
    if (hipEventRecord(e, st) != hipSuccess) return;
    s->stage_name[s->stage_count++] = name;
}

template <typename T>
Desc<T> &desc_of(bsk_spline s);
template <>
inline Desc<float> &desc_of<float>(bsk_spline s) { return s->d32; }
template <>
inline Desc<double> &desc_of<double>(bsk_spline s) { return s->d64; }
template <typename T>
UniDesc<T> &uni_of(bsk_spline s);
template <>
inline UniDesc<float> &uni_of<float>(bsk_spline s) { return s->u32; }
template <>
inline UniDesc<double> &uni_of<double>(bsk_spline s) { return s->u64; }
template <typename T>
UniDescN<T> &uniN_of(bsk_spline s);
template <>
inline UniDescN<float> &uniN_of<float>(bsk_spline s) { return s->un32; }
template <>
inline UniDescN<double> &uniN_of<double>(bsk_spline s) { return s->un64; }
template <typename T>
TileDesc<T> &tile_of(bsk_spline s);
template <>
inline TileDesc<float> &tile_of<float>(bsk_spline s) { return s->t32; }
template <>
inline TileDesc<double> &tile_of<double>(bsk_spline s) { return s->t64; }


// Raise a kernel's dynamic-LDS limit above 64 KiB.  hipFuncSetAttribute is a driver call
// (tens of microseconds): it is issued once per kernel, device and size, not per launch.
template <typename K>
inline hipError_t allow_lds(K kernel, size_t bytes)
{
    if (bytes <= 64 * 1024) return hipSuccess;
    static thread_local std::vector<std::tuple<const void *, int, size_t>> done;
    int dev = 0;
    (void)hipGetDevice(&dev);
    const void *fn = reinterpret_cast<const void *>(kernel);
    for (const auto &e : done)
        if (std::get<0>(e) == fn && std::get<1>(e) == dev && std::get<2>(e) >= bytes) return hipSuccess;
    hipError_t err = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (err == hipSuccess) done.emplace_back(fn, dev, bytes);
    return err;
}


// Large-table evaluation (bsk_gather_tu.hip): control-point-major gather, or the cell-order pipeline
// for batches of >= 2^18 points.  `mixed`: variables of different orders (kernels run at the largest).
// Returns BSK_ERR_UNSUPPORTED when the shape is not covered (the caller falls back).
template <typename T>
bsk_status gather_or_binned_any(bsk_spline s, bool mixed, const Params<T> &prm, long long n, T *out, long long ostride,
                                const Wrt &w, hipStream_t st);

// Fused jacobian of large batches on L2-resident tables (bsk_gather_tu.hip, eval_cellsort<..., JAC>);
// BSK_ERR_UNSUPPORTED when the shape is not covered.
template <typename T>
bsk_status cellsort_jacobian_any(bsk_spline s, const Params<T> &prm, long long n, T *out, hipStream_t st);

// Surfaces whose table is streamed through LDS (bsk_slab_tu.hip, eval_slab2); BSK_ERR_UNSUPPORTED when not covered.
template <typename T>
bsk_status slab2_any(bsk_spline s, bool mixed, const Params<T> &prm, long long n, T *out, long long ostride, const Wrt &w,
                     hipStream_t st);
