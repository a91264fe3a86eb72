// Kernels of libbspy_amd (gfx950).  One lane = one parameter point; the parameter batch
// and the results stream through HBM fully coalesced (SoA), the spline's tables are
// staged once per workgroup into LDS and the workgroups are persistent (grid-stride).
#pragma once
#include "bsk_device.hpp"

namespace bsk {

constexpr int BLOCK_MAX = 1024;

// Copy the axis table (and, when it fits, the coefficient table) into LDS.
template <typename T>
__device__ __forceinline__ void stage_tables(T *stab, T *scoef, const T *__restrict__ gtab, int tab_len,
                                             const T *__restrict__ gcoef, int coef_len, bool with_coefs)
{
    for (int i = threadIdx.x; i < tab_len; i += blockDim.x) stab[i] = gtab[i];
    if (with_coefs)
        for (int i = threadIdx.x; i < coef_len; i += blockDim.x) scoef[i] = gcoef[i];
    __syncthreads();
}

// ---------------------------------------------------------------------------------
// evaluate / derivative, all variables of the same compile-time order O.
//   LDSC: coefficient table staged in LDS (else gathered from global memory / L2).
// out[d * ostride + n]
// ---------------------------------------------------------------------------------
template <typename T, int NIND, int O, bool LDSC>
__global__ __launch_bounds__(BLOCK_MAX) void eval_fixed(const Desc<T> d, const T *__restrict__ gtab,
                                                        const T *__restrict__ gcoef, const Params<T> prm,
                                                        const long long N, T *__restrict__ out,
                                                        const long long ostride, const Wrt wrt,
                                                        unsigned long long *bad)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T *stab = reinterpret_cast<T *>(smem);
    T *scoef = stab + ((d.tab_len + 1) & ~1);
    stage_tables(stab, scoef, gtab, d.tab_len, gcoef, d.coef_len, LDSC);

    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += stride) {
        T b[NIND][O];
        int base = 0;
        bool outside = false;
#pragma unroll
        for (int iv = 0; iv < NIND; ++iv) {
            const T u = prm.p[iv][n];
            outside |= (u < d.lo[iv]) | (u > d.hi[iv]);
            const T *tab = stab + d.off[iv];
            const int ix = find_span<T>(tab, O, d.ncoef[iv], d.steps[iv], u);
            basis_fixed<T, O>(tab, d.nk[iv], ix, u, wrt.w[iv], b[iv]);
            base += (ix - O) * d.cstride[iv + 1];
        }
        if (outside) record_bad(bad, n);
        for (int dep = 0; dep < d.nDep; ++dep) {
            T r;
            if constexpr (LDSC) {
                const T *c = scoef + dep * d.cstride[0] + base;
                if constexpr (NIND == 1) r = contract1<T, O>(c, b[0]);
                else if constexpr (NIND == 2) r = contract2<T, O>(c, d.cstride[1], b[0], b[1]);
                else r = contract3<T, O>(c, d.cstride[1], d.cstride[2], b[0], b[1], b[2]);
            } else {
                const T *__restrict__ c = gcoef + dep * d.cstride[0] + base;
                if constexpr (NIND == 1) r = contract1<T, O>(c, b[0]);
                else if constexpr (NIND == 2) r = contract2<T, O>(c, d.cstride[1], b[0], b[1]);
                else r = contract3<T, O>(c, d.cstride[1], d.cstride[2], b[0], b[1], b[2]);
            }
            out[dep * ostride + n] = r;
        }
    }
}

// Window contraction for eval_mixed, any number of variables: variable IV outermost, the last
// variable innermost - the reference's `myCoefs @ bValues[iv]` from the last variable to the first
// (bspy/_spline_evaluation.py:162-163).  Entries below pad[iv] carry weight zero and are not read.
template <typename T, int NIND, int OMAX, int IV>
__device__ __forceinline__ T mixed_contract(const T *c, const int (&cstride)[MAXI + 1], const int (&pad)[NIND],
                                            const T (&b)[NIND][OMAX])
{
    if constexpr (IV == NIND) {
        return *c;
    } else {
        T acc = T(0);
#pragma unroll
        for (int a = 0; a < OMAX; ++a)
            if (a >= pad[IV]) acc += mixed_contract<T, NIND, OMAX, IV + 1>(c + a * cstride[IV + 1], cstride, pad, b) * b[IV][a];
        return acc;
    }
}

// ---------------------------------------------------------------------------------
// evaluate / derivative, variables of DIFFERENT orders (CAD surfaces are often order (2, k) or
// (3, k); the reference's fixture examples/TomsNasty.json is order (4, 5)): every variable is run
// as order OMAX = max(order) with right-aligned basis arrays (basis_bounded) - the window starts at
// ix - OMAX and its first OMAX - order entries per variable carry weight zero and are not loaded.
// Same structure as eval_fixed otherwise.  out[d * ostride + n]
// ---------------------------------------------------------------------------------
template <typename T, int NIND, int OMAX, bool LDSC>
__global__ __launch_bounds__(BLOCK_MAX) void eval_mixed(const Desc<T> d, const T *__restrict__ gtab,
                                                        const T *__restrict__ gcoef, const Params<T> prm,
                                                        const long long N, T *__restrict__ out,
                                                        const long long ostride, const Wrt wrt,
                                                        unsigned long long *bad)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T *stab = reinterpret_cast<T *>(smem);
    T *scoef = stab + ((d.tab_len + 1) & ~1);
    stage_tables(stab, scoef, gtab, d.tab_len, gcoef, d.coef_len, LDSC);
    int pad[NIND];                                            // leading zero-weight entries per variable
#pragma unroll
    for (int iv = 0; iv < NIND; ++iv) pad[iv] = OMAX - d.order[iv];

    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += stride) {
        T b[NIND][OMAX];
        int base = 0;
        bool outside = false;
#pragma unroll
        for (int iv = 0; iv < NIND; ++iv) {
            const T u = prm.p[iv][n];
            outside |= (u < d.lo[iv]) | (u > d.hi[iv]);
            const T *tab = stab + d.off[iv];
            const int ix = find_span<T>(tab, d.order[iv], d.ncoef[iv], d.steps[iv], u);
            basis_bounded<T, OMAX>(tab, d.nk[iv], d.order[iv], ix, u, wrt.w[iv], b[iv]);
            base += (ix - OMAX) * d.cstride[iv + 1];          // may point before the table: padded entries are never read
        }
        if (outside) record_bad(bad, n);
        for (int dep = 0; dep < d.nDep; ++dep) {
            const T *c = (LDSC ? scoef : gcoef) + dep * d.cstride[0] + base;
            const T r = mixed_contract<T, NIND, OMAX, 0>(c, d.cstride, pad, b);
            nt_store(&out[dep * ostride + n], r);
        }
    }
}

// ---------------------------------------------------------------------------------
// fused jacobian for variables of different orders (and orders beyond jac_fixed): eval_mixed's
// right-aligned windows with a value and a first-derivative basis per variable; partial j contracts
// the window with db of variable j and b of the others.  Tables and coefficients in LDS.
// out[(dep * NIND + j) * N + n]
// ---------------------------------------------------------------------------------
template <typename T, int NIND, int OMAX, int IV, int J>
__device__ __forceinline__ T mixed_contract_d(const T *c, const int (&cstride)[MAXI + 1], const int (&pad)[NIND],
                                              const T (&b)[NIND][OMAX], const T (&db)[NIND][OMAX])
{
    if constexpr (IV == NIND) {
        return *c;
    } else {
        T acc = T(0);
#pragma unroll
        for (int a = 0; a < OMAX; ++a)
            if (a >= pad[IV])
                acc += mixed_contract_d<T, NIND, OMAX, IV + 1, J>(c + a * cstride[IV + 1], cstride, pad, b, db) *
                       (IV == J ? db[IV][a] : b[IV][a]);
        return acc;
    }
}

template <typename T, int NIND, int OMAX, int J>
__device__ __forceinline__ void mixed_partials(const T *c, const int (&cstride)[MAXI + 1], const int (&pad)[NIND],
                                               const T (&b)[NIND][OMAX], const T (&db)[NIND][OMAX], T *o, long long N)
{
    if constexpr (J < NIND) {
        nt_store(&o[J * N], mixed_contract_d<T, NIND, OMAX, 0, J>(c, cstride, pad, b, db));
        mixed_partials<T, NIND, OMAX, J + 1>(c, cstride, pad, b, db, o, N);
    }
}

template <typename T, int NIND, int OMAX>
__global__ __launch_bounds__(BLOCK_MAX) void jac_mixed(const Desc<T> d, const T *__restrict__ gtab,
                                                       const T *__restrict__ gcoef, const Params<T> prm,
                                                       const long long N, T *__restrict__ out,
                                                       unsigned long long *bad)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T *stab = reinterpret_cast<T *>(smem);
    T *scoef = stab + ((d.tab_len + 1) & ~1);
    stage_tables(stab, scoef, gtab, d.tab_len, gcoef, d.coef_len, true);
    int pad[NIND];
#pragma unroll
    for (int iv = 0; iv < NIND; ++iv) pad[iv] = OMAX - d.order[iv];

    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += stride) {
        T b[NIND][OMAX], db[NIND][OMAX];
        int base = 0;
        bool outside = false;
#pragma unroll
        for (int iv = 0; iv < NIND; ++iv) {
            const T u = prm.p[iv][n];
            outside |= (u < d.lo[iv]) | (u > d.hi[iv]);
            const T *tab = stab + d.off[iv];
            const int ix = find_span<T>(tab, d.order[iv], d.ncoef[iv], d.steps[iv], u);
            basis_bounded<T, OMAX>(tab, d.nk[iv], d.order[iv], ix, u, 0, b[iv]);
            basis_bounded<T, OMAX>(tab, d.nk[iv], d.order[iv], ix, u, 1, db[iv]);
            base += (ix - OMAX) * d.cstride[iv + 1];
        }
        if (outside) record_bad(bad, n);
        for (int dep = 0; dep < d.nDep; ++dep)
            mixed_partials<T, NIND, OMAX, 0>(scoef + dep * d.cstride[0] + base, d.cstride, pad, b, db,
                                             out + (long long)dep * NIND * N + n, N);
    }
}

// ---------------------------------------------------------------------------------
// fused jacobian: every first partial derivative from one span search and one
// recursion per variable, coefficients read once.  out[(dep * NIND + j) * N + n]
// ---------------------------------------------------------------------------------
template <typename T, int NIND, int O, bool LDSC>
__global__ __launch_bounds__(BLOCK_MAX) void jac_fixed(const Desc<T> d, const T *__restrict__ gtab,
                                                       const T *__restrict__ gcoef, const Params<T> prm,
                                                       const long long N, T *__restrict__ out,
                                                       unsigned long long *bad)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T *stab = reinterpret_cast<T *>(smem);
    T *scoef = stab + ((d.tab_len + 1) & ~1);
    stage_tables(stab, scoef, gtab, d.tab_len, gcoef, d.coef_len, LDSC);

    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += stride) {
        T b[NIND][O], db[NIND][O];
        int base = 0;
        bool outside = false;
#pragma unroll
        for (int iv = 0; iv < NIND; ++iv) {
            const T u = prm.p[iv][n];
            outside |= (u < d.lo[iv]) | (u > d.hi[iv]);
            const T *tab = stab + d.off[iv];
            const int ix = find_span<T>(tab, O, d.ncoef[iv], d.steps[iv], u);
            basis_value_and_d1<T, O>(tab, d.nk[iv], ix, u, b[iv], db[iv]);
            base += (ix - O) * d.cstride[iv + 1];
        }
        if (outside) record_bad(bad, n);
        for (int dep = 0; dep < d.nDep; ++dep) {
            const T *c = (LDSC ? (const T *)scoef : gcoef) + dep * d.cstride[0] + base;
            T *o = out + (long long)dep * NIND * N + n;
            if constexpr (NIND == 1) {
                o[0] = contract1<T, O>(c, db[0]);
            } else if constexpr (NIND == 2) {
                const int s0 = d.cstride[1];
                T j0 = T(0), j1 = T(0);
#pragma unroll
                for (int a = 0; a < O; ++a) {
                    T t = T(0), td = T(0);
#pragma unroll
                    for (int k = 0; k < O; ++k) {
                        const T cv = c[a * s0 + k];
                        t += cv * b[1][k];
                        td += cv * db[1][k];
                    }
                    j0 += t * db[0][a];
                    j1 += td * b[0][a];
                }
                o[0] = j0;
                o[N] = j1;
            } else {
                const int s0 = d.cstride[1], s1 = d.cstride[2];
                T j0 = T(0), j1 = T(0), j2 = T(0);
#pragma unroll
                for (int a = 0; a < O; ++a) {
                    T s = T(0), sb = T(0), sc = T(0);
#pragma unroll
                    for (int k = 0; k < O; ++k) {
                        T t = T(0), td = T(0);
#pragma unroll
                        for (int m = 0; m < O; ++m) {
                            const T cv = c[a * s0 + k * s1 + m];
                            t += cv * b[2][m];
                            td += cv * db[2][m];
                        }
                        s += t * b[1][k];
                        sb += t * db[1][k];
                        sc += td * b[1][k];
                    }
                    j0 += s * db[0][a];
                    j1 += sb * b[0][a];
                    j2 += sc * b[0][a];
                }
                o[0] = j0;
                o[N] = j1;
                o[2 * N] = j2;
            }
        }
    }
}

// ---------------------------------------------------------------------------------
// generic fallback: any nInd <= MAXI, any per-variable order <= MAXO.
// Tables are read from global memory (L1/L2 resident), basis rows live in private memory.
// ---------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void eval_generic(const Desc<T> d, const T *__restrict__ gtab,
                                                    const T *__restrict__ gcoef, const Params<T> prm,
                                                    const long long N, T *__restrict__ out,
                                                    const long long ostride, const Wrt wrt,
                                                    unsigned long long *bad)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += stride) {
        T b[MAXI * MAXO];
        int base = 0, win = 1;
        bool outside = false;
        for (int iv = 0; iv < d.nInd; ++iv) {
            const T u = prm.p[iv][n];
            outside |= (u < d.lo[iv]) | (u > d.hi[iv]);
            const T *tab = gtab + d.off[iv];
            const int O = d.order[iv];
            const int ix = find_span<T>(tab, O, d.ncoef[iv], d.steps[iv], u);
            basis_runtime<T>(tab, d.nk[iv], O, ix, u, wrt.w[iv], false, b + iv * MAXO);
            base += (ix - O) * d.cstride[iv + 1];
            win *= O;
        }
        if (outside) record_bad(bad, n);
        for (int dep = 0; dep < d.nDep; ++dep) {
            const T *__restrict__ c = gcoef + dep * d.cstride[0] + base;
            // odometer over the window, last variable fastest; the weight of an element is
            // the product of its per-variable basis values
            int idx[MAXI];
            for (int iv = 0; iv < d.nInd; ++iv) idx[iv] = 0;
            T acc = T(0);
            for (int w = 0; w < win; ++w) {
                int off = 0;
                T wgt = T(1);
                for (int iv = 0; iv < d.nInd; ++iv) {
                    off += idx[iv] * d.cstride[iv + 1];
                    wgt *= b[iv * MAXO + idx[iv]];
                }
                acc += c[off] * wgt;
                for (int iv = d.nInd - 1; iv >= 0; --iv) {
                    if (++idx[iv] < d.order[iv]) break;
                    idx[iv] = 0;
                }
            }
            out[dep * ostride + n] = acc;
        }
    }
}

// ---------------------------------------------------------------------------------
// batched bspline_values: span index + basis row per parameter (reference
// bspy/_spline_evaluation.py:4-27).  Also stage 1 of the tensor-product grid path.
//   tab: axis table of ONE variable (knots + reciprocal rows), global memory.
//   bad_flag (grid path): set ix negative-encoded?  no: out-of-domain entries are
//   reported through `outside` (one byte per parameter) when it is not NULL.
// ---------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void basis_rows(const T *__restrict__ tab, int nk, int order, int ncoef,
                                                  int steps, const T *__restrict__ u, long long n, int wrt,
                                                  int taylor, const int *__restrict__ knot_in,
                                                  int *__restrict__ ix_out, T *__restrict__ basis_out,
                                                  T lo, T hi, unsigned char *__restrict__ outside)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const T x = u[i];
    T b[MAXO];
    const int ix = knot_in ? knot_in[i] : find_span<T>(tab, order, ncoef, steps, x);
    basis_runtime<T>(tab, nk, order, ix, x, wrt, taylor != 0, b);
    ix_out[i] = ix;
    for (int k = 0; k < order; ++k) basis_out[i * order + k] = b[k];
    if (outside) outside[i] = (x < lo) | (x > hi);
}

struct GridDims {
    long long n[MAXI];     // grid points per variable
    long long goff[MAXI];  // offset of variable iv in ixs / outside; rows offset = goff * order (see roff)
    long long roff[MAXI];  // offset of variable iv in rows
};

// Stage 1 of the tensor-product grid for ALL variables in one launch (blockIdx.y = variable):
// span index, basis row and out-of-domain flag of every grid parameter.
template <typename T>
struct GridAxes {
    const T *u[MAXI];      // grid parameters of variable iv (device)
    int wrt[MAXI];
};

template <typename T>
__global__ __launch_bounds__(256) void basis_rows_grid(const Desc<T> d, const T *__restrict__ gtab, const GridAxes<T> ax,
                                                       const GridDims g, int *__restrict__ ixs, T *__restrict__ rows,
                                                       unsigned char *__restrict__ outside)
{
    const int iv = blockIdx.y;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= g.n[iv]) return;
    const T *tab = gtab + d.off[iv];
    const int order = d.order[iv];
    const T x = ax.u[iv][i];
    T b[MAXO];
    const int ix = find_span<T>(tab, order, d.ncoef[iv], d.steps[iv], x);
    basis_runtime<T>(tab, d.nk[iv], order, ix, x, ax.wrt[iv], false, b);
    ixs[g.goff[iv] + i] = ix;
    for (int k = 0; k < order; ++k) rows[g.roff[iv] + i * order + k] = b[k];
    outside[g.goff[iv] + i] = (x < d.lo[iv]) | (x > d.hi[iv]);
}

// ---------------------------------------------------------------------------------
// tensor-product grid, stage 2: one lane per output point, last grid variable fastest
// (coalesced stores).  Basis rows come from stage 1 (tiny, cache resident): the per-point
// work is the window contraction only.
//   ixs / rows: concatenated per variable; goff[iv] = first parameter of variable iv.
// out[dep * total + flat]
// ---------------------------------------------------------------------------------

template <typename T>
__global__ __launch_bounds__(256) void grid_generic(const Desc<T> d, const T *__restrict__ gcoef,
                                                    const GridDims g, const int *__restrict__ ixs,
                                                    const T *__restrict__ rows,
                                                    const unsigned char *__restrict__ outside,
                                                    const long long total, T *__restrict__ out,
                                                    unsigned long long *bad)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x; n < total; n += stride) {
        long long rem = n;
        const T *brow[MAXI];
        int base = 0, win = 1;
        bool out_of_domain = false;
        for (int iv = d.nInd - 1; iv >= 0; --iv) {
            const long long gi = rem % g.n[iv];
            rem /= g.n[iv];
            const int O = d.order[iv];
            brow[iv] = rows + g.roff[iv] + gi * O;
            base += (ixs[g.goff[iv] + gi] - O) * d.cstride[iv + 1];
            out_of_domain |= outside[g.goff[iv] + gi] != 0;
            win *= O;
        }
        if (out_of_domain) record_bad(bad, n);
        for (int dep = 0; dep < d.nDep; ++dep) {
            const T *__restrict__ c = gcoef + dep * d.cstride[0] + base;
            int idx[MAXI];
            for (int iv = 0; iv < d.nInd; ++iv) idx[iv] = 0;
            T acc = T(0);
            for (int w = 0; w < win; ++w) {
                int off = 0;
                T wgt = T(1);
                for (int iv = 0; iv < d.nInd; ++iv) {
                    off += idx[iv] * d.cstride[iv + 1];
                    wgt *= brow[iv][idx[iv]];
                }
                acc += c[off] * wgt;
                for (int iv = d.nInd - 1; iv >= 0; --iv) {
                    if (++idx[iv] < d.order[iv]) break;
                    idx[iv] = 0;
                }
            }
            out[dep * total + n] = acc;
        }
    }
}

// Surface fast path of the grid: order O x O, coefficient window in registers per
// (i0 row); one lane per output point along the last variable.
template <typename T, int O>
__global__ __launch_bounds__(256) void grid_surface(const Desc<T> d, const T *__restrict__ gcoef,
                                                    const GridDims g, const int *__restrict__ ixs,
                                                    const T *__restrict__ rows,
                                                    const unsigned char *__restrict__ outside,
                                                    T *__restrict__ out, unsigned long long *bad)
{
    // blockIdx.y = grid row (variable 0), x covers variable 1
    const long long n1 = g.n[1], total = g.n[0] * g.n[1];
    const long long i0 = blockIdx.y;
    const int ix0 = ixs[g.goff[0] + i0];
    const bool bad0 = outside[g.goff[0] + i0] != 0;
    T b0[O];
#pragma unroll
    for (int a = 0; a < O; ++a) b0[a] = rows[g.roff[0] + i0 * O + a];
    const int s0 = d.cstride[1];
    for (long long i1 = (long long)blockIdx.x * blockDim.x + threadIdx.x; i1 < n1;
         i1 += (long long)gridDim.x * blockDim.x) {
        const int ix1 = ixs[g.goff[1] + i1];
        T b1[O];
#pragma unroll
        for (int k = 0; k < O; ++k) b1[k] = rows[g.roff[1] + i1 * O + k];
        const long long flat = i0 * n1 + i1;
        if (bad0 | (outside[g.goff[1] + i1] != 0)) record_bad(bad, flat);
        const int base = (ix0 - O) * s0 + (ix1 - O);
        for (int dep = 0; dep < d.nDep; ++dep) {
            const T *__restrict__ c = gcoef + dep * d.cstride[0] + base;
            out[dep * total + flat] = contract2<T, O>(c, s0, b0, b1);
        }
    }
}

// Row-factored surface grid, compile-time order O for both variables, nDep <= 4 per pass.
// One workgroup per grid row i0: its basis b0 is the same for every point of the row, so the
// workgroup first contracts the FIRST variable for all columns of the coefficient table,
//     rowc[dep][c] = sum_a b0[a] * C[dep][ix0 - O + a][c],     c in [0, nCoef1),
// into LDS (nDep * nCoef1 values), and every grid point of the row then needs O values of rowc
// per dependent variable: O multiply-adds instead of O * O, read from LDS addresses that
// neighbouring lanes share (broadcast).  Lanes own VEC = 16 / sizeof(T) consecutive columns and
// store 16 bytes per dependent variable when the rows are aligned.  The kernel is bound by its
// stores (12 B fp32 / 24 B fp64 per point for nDep 3).  out[dep * total + i0 * n1 + i1]
// MIXED: the two variables have different orders o0, o1 <= O (O = the larger): loops run to O with
// wave-uniform tests; one common order compiles them out.
template <typename T, int O, bool MIXED>
__global__ __launch_bounds__(256) void grid_rows(const Desc<T> d, const T *__restrict__ gcoef, const GridDims g,
                                                 const int *__restrict__ ixs, const T *__restrict__ rows,
                                                 const unsigned char *__restrict__ outside, T *__restrict__ out,
                                                 unsigned long long *bad, const int vec_ok)
{
    extern __shared__ __attribute__((aligned(16))) char smem_g[];
    T *rowc = reinterpret_cast<T *>(smem_g);                  // [nDep][nCoef1]
    constexpr int VEC = 16 / (int)sizeof(T);
    const long long n1 = g.n[1], total = g.n[0] * g.n[1];
    const int nc1 = d.ncoef[1], s0 = d.cstride[1];
    const int o0 = MIXED ? d.order[0] : O, o1 = MIXED ? d.order[1] : O;
    for (long long i0 = blockIdx.x; i0 < g.n[0]; i0 += gridDim.x) {
        const int ix0 = ixs[g.goff[0] + i0];
        const bool bad0 = outside[g.goff[0] + i0] != 0;
        T b0[O];
#pragma unroll
        for (int a = 0; a < O; ++a) b0[a] = (!MIXED || a < o0) ? rows[g.roff[0] + i0 * o0 + a] : T(0);
        __syncthreads();                                       // the previous row's readers are done
        for (int e = threadIdx.x; e < d.nDep * nc1; e += blockDim.x) {
            const int dep = e / nc1, c = e - dep * nc1;
            const T *__restrict__ col = gcoef + dep * d.cstride[0] + (ix0 - o0) * s0 + c;
            T acc = T(0);
#pragma unroll
            for (int a = 0; a < O; ++a)
                if (!MIXED || a < o0) acc += b0[a] * col[a * s0];
            rowc[e] = acc;
        }
        __syncthreads();
        if (vec_ok) {
            for (long long c0 = (long long)threadIdx.x * VEC; c0 < n1; c0 += (long long)blockDim.x * VEC) {
                int ix1[VEC];
                T b1[VEC][O];
                bool anybad = bad0;
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    ix1[v] = ixs[g.goff[1] + c0 + v] - o1;
#pragma unroll
                    for (int k = 0; k < O; ++k) b1[v][k] = (!MIXED || k < o1) ? rows[g.roff[1] + (c0 + v) * o1 + k] : T(0);
                    anybad |= outside[g.goff[1] + c0 + v] != 0;
                }
                if (anybad) {
#pragma unroll
                    for (int v = 0; v < VEC; ++v)
                        if (bad0 | (outside[g.goff[1] + c0 + v] != 0)) record_bad(bad, i0 * n1 + c0 + v);
                }
                for (int dep = 0; dep < d.nDep; ++dep) {
                    T res[VEC];
#pragma unroll
                    for (int v = 0; v < VEC; ++v) {
                        const T *rc = rowc + dep * nc1 + ix1[v];
                        T acc = T(0);
#pragma unroll
                        for (int k = 0; k < O; ++k)
                            if (!MIXED || k < o1) acc += rc[k] * b1[v][k];
                        res[v] = acc;
                    }
                    T *o = out + dep * total + i0 * n1 + c0;
                    // 16-byte non-temporal store: the grid is written once and not read here
                    typedef T vec_t __attribute__((ext_vector_type(VEC)));
                    vec_t w;
#pragma unroll
                    for (int v = 0; v < VEC; ++v) w[v] = res[v];
                    __builtin_nontemporal_store(w, reinterpret_cast<vec_t *>(o));
                }
            }
        } else {
            for (long long i1 = threadIdx.x; i1 < n1; i1 += blockDim.x) {
                const int ix1 = ixs[g.goff[1] + i1] - o1;
                T b1[O];
#pragma unroll
                for (int k = 0; k < O; ++k) b1[k] = (!MIXED || k < o1) ? rows[g.roff[1] + i1 * o1 + k] : T(0);
                if (bad0 | (outside[g.goff[1] + i1] != 0)) record_bad(bad, i0 * n1 + i1);
                for (int dep = 0; dep < d.nDep; ++dep) {
                    const T *rc = rowc + dep * nc1 + ix1;
                    T acc = T(0);
#pragma unroll
                    for (int k = 0; k < O; ++k)
                        if (!MIXED || k < o1) acc += rc[k] * b1[k];
                    __builtin_nontemporal_store(acc, &out[dep * total + i0 * n1 + i1]);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------
// tess_rows: tessellation of a BATCH of surface patches in 3-D on one parameter grid - positions
// and (NORMALS) normals of every patch in one launch (the compute-side counterpart
// of the reference's GLSL tessellation shaders, bspy/splineOpenGLFrame.py:671-714; per patch it
// replaces s(u[:, None], v[None, :]) and Spline.normal at every grid point,
// bspy/_spline_evaluation.py:215-246).  The patches share orders, nCoef and knots (the 32 Bezier
// patches of the Utah teapot), so the basis rows of the grid lines are computed once; grid_rows'
// row factorisation is applied per (patch, grid row): rowc = sum_a b0[a] C[.., a, :] and, for the
// normals, drowc = sum_a b0'[a] C[.., a, :]; a point then needs O multiply-adds per output:
//   position = rowc . b1,   S_u = drowc . b1,   S_v = rowc . b1',   normal = S_u x S_v.
// blockIdx.y = patch.  pos / nrm [((patch * 3 + dep) * n0 + i0) * n1 + i1]
// ---------------------------------------------------------------------------------
constexpr int TESS_MAX_PATCHES = 64;
constexpr int TESS_R = 4;               // grid rows per barrier pair of the hoisted (positions-only) form
template <typename T>
struct PatchCoefs {
    const T *c[TESS_MAX_PATCHES];
};

template <typename T>
__device__ __forceinline__ void tess_normal(const T (&su)[3], const T (&sv)[3], int normalize, int negate, T (&n)[3])
{
    // cofactors of the 3 x 2 tangent space, as in jac_rowrot<.., NORMAL>
    T nx = su[1] * sv[2] - sv[1] * su[2];
    T ny = -(su[0] * sv[2] - sv[0] * su[2]);
    T nz = su[0] * sv[1] - sv[0] * su[1];
    if (negate) { nx = -nx; ny = -ny; nz = -nz; }
    if (normalize) {
        const T len = sqrt(nx * nx + ny * ny + nz * nz);
        nx = nx / len; ny = ny / len; nz = nz / len;
    }
    n[0] = nx; n[1] = ny; n[2] = nz;
}

// MIXED: the two variables have different orders o0, o1 <= O (as grid_rows).
// HIT: hoisted vector steps per lane of the positions-only form - 2 at 256 lanes, 1 at 512 lanes (half the hoisted
// registers: twice the waves per SIMD)
template <typename T, int O, bool NORMALS, bool MIXED, int HIT = 2>
__global__ __launch_bounds__(HIT == 1 ? 512 : 256) void tess_rows(const Desc<T> d, const PatchCoefs<T> pc, const GridDims g,
                                                 const int *__restrict__ ixs, const T *__restrict__ rows,
                                                 const T *__restrict__ drows, const unsigned char *__restrict__ outside,
                                                 T *__restrict__ pos, T *__restrict__ nrm, unsigned long long *bad,
                                                 const int vec_ok, const int normalize, const int negate, const int tess_r)
{
    extern __shared__ __attribute__((aligned(16))) char smem_g[];
    const int nc1 = d.ncoef[1], s0 = d.cstride[1];
    const int o0 = MIXED ? d.order[0] : O, o1 = MIXED ? d.order[1] : O;
    T *rowc = reinterpret_cast<T *>(smem_g);                  // [3][nCoef1]
    T *drowc = rowc + 3 * nc1;                                // [3][nCoef1]  (NORMALS)
    constexpr int VEC = 16 / (int)sizeof(T);
    typedef T vec_t __attribute__((ext_vector_type(VEC)));
    const long long n1 = g.n[1], total = g.n[0] * g.n[1];
    const T *__restrict__ gcoef = pc.c[blockIdx.y];
    T *ppos = pos + (long long)blockIdx.y * 3 * total;
    T *pnrm = NORMALS ? nrm + (long long)blockIdx.y * 3 * total : nullptr;
    // The columns a lane works on are the same for every grid row of this workgroup: when the row is short
    // enough (two vector steps per lane) their span indices and basis rows of the second variable are taken
    // into registers ONCE, and a row then costs LDS reads of the contracted row, multiply-adds and stores only
    // (before: 21 B of L2 reads per 12 B written).
    // (positions only: with normals the second basis row per column costs the registers that four waves per
    // SIMD need - measured 0.78 -> 1.07 ms at 256 lanes, 0.78 -> 0.89 ms in the 512-lane form at 256 registers)
    const bool hoist = !NORMALS && vec_ok && !MIXED && n1 <= (long long)blockDim.x * VEC * HIT;
    T hb1[HIT][VEC][O], hdb1[NORMALS ? HIT : 1][NORMALS ? VEC : 1][O];
    int hix1[HIT][VEC];
    bool hbad[HIT][VEC];
    if (hoist) {
#pragma unroll
        for (int it = 0; it < HIT; ++it) {
            const long long c0 = ((long long)it * blockDim.x + threadIdx.x) * VEC;
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                const long long c = c0 + v < n1 ? c0 + v : n1 - 1;
                hix1[it][v] = ixs[g.goff[1] + c] - O;
                hbad[it][v] = outside[g.goff[1] + c] != 0;
#pragma unroll
                for (int k = 0; k < O; ++k) {
                    hb1[it][v][k] = rows[g.roff[1] + c * O + k];
                    if constexpr (NORMALS) hdb1[it][v][k] = drows[g.roff[1] + c * O + k];
                }
            }
        }
    }
    if (hoist) {
        // TESS_R grid rows per barrier pair: the two dependent global round trips in front of a row's contraction (span
        // index, then the control points it selects) and the barriers are paid once per 4 rows = 96 KB of stores; a lane
        // whose columns share a span reads its contracted control points once; 512 lanes with one hoisted step each
        // (70 registers, 7 waves per SIMD) instead of 256 with two (106 registers).  tools/tess_sweep.py, one placement
        // of the 1.6 GB result, interleaved medians: 0.341 (round-2 form) -> 0.329 (256 lanes) -> 0.312 - 0.316 ms (512
        // lanes, 2 - 4 workgroups per CU).  The PLACEMENT of the result moves every variant by +- 12 % from run to run
        // (0.25 - 0.33 ms for one build); a linear fill of the same bytes runs at 6.9 TB/s (tools/write_floor.py)
        const int rsz = 3 * nc1;
        // (every patch starting at another row unit - the planes lie a power of two apart - measured no different)
        for (long long ib = (long long)blockIdx.x * tess_r; ib < g.n[0]; ib += (long long)gridDim.x * tess_r) {
            const int nr = (int)((g.n[0] - ib) < tess_r ? (g.n[0] - ib) : tess_r);
            __syncthreads();                                   // the previous rows' readers are done
            for (int e = threadIdx.x; e < nr * rsz; e += blockDim.x) {
                const int r = e / rsz, e2 = e - r * rsz, dep = e2 / nc1, c = e2 - dep * nc1;
                const long long i0 = ib + r;
                const int ix0 = ixs[g.goff[0] + i0];
                const T *__restrict__ col = gcoef + dep * d.cstride[0] + (ix0 - O) * s0 + c;
                T acc = T(0);
#pragma unroll
                for (int a = 0; a < O; ++a) acc += rows[g.roff[0] + i0 * O + a] * col[a * s0];
                rowc[r * rsz + e2] = acc;
            }
            __syncthreads();
            for (int r = 0; r < nr; ++r) {
                const long long i0 = ib + r;
                const bool bad0 = outside[g.goff[0] + i0] != 0;
                const T *rowr = rowc + r * rsz;
#pragma unroll
                for (int it = 0; it < HIT; ++it) {
                    const long long c0 = ((long long)it * blockDim.x + threadIdx.x) * VEC;
                    if (c0 < n1) {
                        T P[3][VEC];
#pragma unroll
                        for (int v = 0; v < VEC; ++v)
                            if (blockIdx.y == 0 && (bad0 | hbad[it][v])) record_bad(bad, i0 * n1 + c0 + v);
                        if (hix1[it][0] == hix1[it][VEC - 1]) {
                            // the lane's columns lie in one span (always, for Bezier patches): its 3 x O contracted control
                            // points are read once for the VEC columns (LDS instructions per point: 3 O -> 3 O / VEC)
                            T rcv[3][O];
#pragma unroll
                            for (int dep = 0; dep < 3; ++dep)
#pragma unroll
                                for (int k = 0; k < O; ++k) rcv[dep][k] = rowr[dep * nc1 + hix1[it][0] + k];
#pragma unroll
                            for (int v = 0; v < VEC; ++v)
#pragma unroll
                                for (int dep = 0; dep < 3; ++dep) {
                                    T p = T(0);
#pragma unroll
                                    for (int k = 0; k < O; ++k) p += rcv[dep][k] * hb1[it][v][k];
                                    P[dep][v] = p;
                                }
                        } else {
#pragma unroll
                            for (int v = 0; v < VEC; ++v) {
#pragma unroll
                                for (int dep = 0; dep < 3; ++dep) {
                                    const T *rc = rowr + dep * nc1 + hix1[it][v];
                                    T p = T(0);
#pragma unroll
                                    for (int k = 0; k < O; ++k) p += rc[k] * hb1[it][v][k];
                                    P[dep][v] = p;
                                }
                            }
                        }
#pragma unroll
                        for (int dep = 0; dep < 3; ++dep) {
                            vec_t w;
#pragma unroll
                            for (int v = 0; v < VEC; ++v) w[v] = P[dep][v];
                            __builtin_nontemporal_store(w, reinterpret_cast<vec_t *>(ppos + dep * total + i0 * n1 + c0));   // (plain stores: 0.302 -> 0.319 ms)
                        }
                    }
                }
            }
        }
        return;
    }
    for (long long i0 = blockIdx.x; i0 < g.n[0]; i0 += gridDim.x) {
        const int ix0 = ixs[g.goff[0] + i0];
        const bool bad0 = outside[g.goff[0] + i0] != 0;
        T b0[O], db0[O];
#pragma unroll
        for (int a = 0; a < O; ++a) {
            const bool on = !MIXED || a < o0;
            b0[a] = on ? rows[g.roff[0] + i0 * o0 + a] : T(0);
            db0[a] = (NORMALS && on) ? drows[g.roff[0] + i0 * o0 + a] : T(0);
        }
        __syncthreads();                                       // the previous row's readers are done
        for (int e = threadIdx.x; e < 3 * nc1; e += blockDim.x) {
            const int dep = e / nc1, c = e - dep * nc1;
            const T *__restrict__ col = gcoef + dep * d.cstride[0] + (ix0 - o0) * s0 + c;
            T acc = T(0), dacc = T(0);
#pragma unroll
            for (int a = 0; a < O; ++a) {
                if (!MIXED || a < o0) {
                    const T cv = col[a * s0];
                    acc += b0[a] * cv;
                    dacc += db0[a] * cv;
                }
            }
            rowc[e] = acc;
            if constexpr (NORMALS) drowc[e] = dacc;
        }
        __syncthreads();
        const long long step = vec_ok ? VEC : 1;
        for (long long c0 = (long long)threadIdx.x * step; c0 < n1; c0 += (long long)blockDim.x * step) {
            T P[3][VEC], Nn[3][VEC];
            const int nv = vec_ok ? VEC : 1;
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                if (v < nv) {
                    const int ix1 = ixs[g.goff[1] + c0 + v] - o1;
                    T b1[O], db1[O];
#pragma unroll
                    for (int k = 0; k < O; ++k) {
                        const bool on = !MIXED || k < o1;
                        b1[k] = on ? rows[g.roff[1] + (c0 + v) * o1 + k] : T(0);
                        db1[k] = (NORMALS && on) ? drows[g.roff[1] + (c0 + v) * o1 + k] : T(0);
                    }
                    if (blockIdx.y == 0 && (bad0 | (outside[g.goff[1] + c0 + v] != 0))) record_bad(bad, i0 * n1 + c0 + v);
                    T su[3], sv[3];
#pragma unroll
                    for (int dep = 0; dep < 3; ++dep) {
                        const T *rc = rowc + dep * nc1 + ix1;
                        const T *drc = drowc + dep * nc1 + ix1;
                        T p = T(0), u_ = T(0), v_ = T(0);
#pragma unroll
                        for (int k = 0; k < O; ++k) {
                            if (!MIXED || k < o1) {
                                p += rc[k] * b1[k];
                                if constexpr (NORMALS) { u_ += drc[k] * b1[k]; v_ += rc[k] * db1[k]; }
                            }
                        }
                        P[dep][v] = p; su[dep] = u_; sv[dep] = v_;
                    }
                    if constexpr (NORMALS) {
                        T nn[3];
                        tess_normal<T>(su, sv, normalize, negate, nn);
                        Nn[0][v] = nn[0]; Nn[1][v] = nn[1]; Nn[2][v] = nn[2];
                    }
                }
            }
#pragma unroll
            for (int dep = 0; dep < 3; ++dep) {
                T *o = ppos + dep * total + i0 * n1 + c0;
                if (vec_ok) {
                    vec_t w;
#pragma unroll
                    for (int v = 0; v < VEC; ++v) w[v] = P[dep][v];
                    __builtin_nontemporal_store(w, reinterpret_cast<vec_t *>(o));
                } else {
                    __builtin_nontemporal_store(P[dep][0], o);
                }
                if constexpr (NORMALS) {
                    T *q = pnrm + dep * total + i0 * n1 + c0;
                    if (vec_ok) {
                        vec_t w;
#pragma unroll
                        for (int v = 0; v < VEC; ++v) w[v] = Nn[dep][v];
                        __builtin_nontemporal_store(w, reinterpret_cast<vec_t *>(q));
                    } else {
                        __builtin_nontemporal_store(Nn[dep][0], q);
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------
// normal from a batched jacobian (next row: reference bspy/_spline_evaluation.py:215-246).
// |nInd - nDep| == 1, big = max(nInd, nDep) <= 4.  jac[(dep * nInd + j) * N + n] (the layout
// bsk_jacobian writes); the tangent space is taken with its larger dimension first,
// normal[i] = sign * (-1)^i * det(tangent space without row i); out[i * N + n].
// ---------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ T det_small(const T (&a)[9], int m)
{
    if (m == 0) return T(1);
    if (m == 1) return a[0];
    if (m == 2) return a[0] * a[3] - a[1] * a[2];
    return a[0] * (a[4] * a[8] - a[5] * a[7]) - a[1] * (a[3] * a[8] - a[5] * a[6]) + a[2] * (a[3] * a[7] - a[4] * a[6]);
}

template <typename T>
__device__ __forceinline__ void normal_from_tangents(const T (&tan)[4][3], int big, bool normalize, bool negate, T (&nrm)[4])
{
    const int small = big - 1;
    T sumsq = T(0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        nrm[i] = T(0);
        if (i < big) {
            T sub[9];
            int q = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (r < big && r != i) {
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                        if (c < small) sub[q++] = tan[r][c];
                }
            }
            T v = det_small<T>(sub, small);
            if (i & 1) v = -v;
            if (negate) v = -v;
            nrm[i] = v;
            sumsq += v * v;
        }
    }
    if (normalize) {
        const T len = sqrt(sumsq);
#pragma unroll
        for (int i = 0; i < 4; ++i) nrm[i] = nrm[i] / len;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void normal_epilogue(const T *__restrict__ jac, int nInd, int nDep, long long N,
                                                       int normalize, int negate, T *__restrict__ out)
{
    const int big = nInd > nDep ? nInd : nDep, small = big - 1;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += stride) {
        T tan[4][3];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                tan[r][c] = T(0);
                if (r < big && c < small) {
                    // jacobian entry (dep, j): dep = row of the (nDep x nInd) matrix
                    const int dep = nInd > nDep ? c : r, j = nInd > nDep ? r : c;
                    tan[r][c] = jac[((long long)dep * nInd + j) * N + n];
                }
            }
        T nrm[4];
        normal_from_tangents<T>(tan, big, normalize != 0, negate != 0, nrm);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < big) out[(long long)i * N + n] = nrm[i];
    }
}

// ---------------------------------------------------------------------------------
// curvature epilogues (next row: reference bspy/_spline_evaluation.py:80-107) on derivative
// buffers produced by the evaluation kernels (each (nDep, N), SoA).
// ---------------------------------------------------------------------------------
// curves: fp = first, fpp = second derivative; signed in 2-D, unsigned otherwise
template <typename T>
__global__ __launch_bounds__(256) void curvature_curve(const T *__restrict__ fp, const T *__restrict__ fpp, int nDep,
                                                       long long N, T *__restrict__ out)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += stride) {
        T fpfp = T(0), fpfpp = T(0), fppfpp = T(0);
        for (int d = 0; d < nDep; ++d) {
            const T a = fp[(long long)d * N + n], b = fpp[(long long)d * N + n];
            fpfp += a * a;
            fpfpp += a * b;
            fppfpp += b * b;
        }
        const T denom = fpfp * sqrt(fpfp);                       // fpDotFp ** 1.5
        T num;
        if (nDep == 2) num = fp[n] * fpp[N + n] - fp[N + n] * fpp[n];
        else num = sqrt(fppfpp * fpfp - fpfpp * fpfpp);
        out[n] = num / denom;
    }
}

// surfaces in 3-D: Gaussian curvature from su, sv, suu, suv, svv and the unit normal
template <typename T>
__global__ __launch_bounds__(256) void curvature_surface(const T *__restrict__ su, const T *__restrict__ sv,
                                                         const T *__restrict__ suu, const T *__restrict__ suv,
                                                         const T *__restrict__ svv, const T *__restrict__ nrm,
                                                         long long N, T *__restrict__ out)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += stride) {
        T E = T(0), F = T(0), G = T(0), L = T(0), M = T(0), Nn = T(0);
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const long long k = (long long)d * N + n;
            const T a = su[k], b = sv[k], c = nrm[k];
            E += a * a; F += a * b; G += b * b;
            L += suu[k] * c; M += suv[k] * c; Nn += svv[k] * c;
        }
        out[n] = (L * Nn - M * M) / (E * G - F * F);
    }
}

}  // namespace bsk
