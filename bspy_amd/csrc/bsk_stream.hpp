// eval_stream: the LDS-table evaluation kernel tuned for latency hiding (gfx950).
//
// One lane = one point; table image (axis tables, span-search bucket tables, coefficients)
// in LDS; persistent workgroups.  The per-point dependency chain is kept short: (1) the next tile's parameters are fetched from HBM before the
// current tile is evaluated, (2) the span searches of all variables advance in lock step
// (one LDS round trip per step for all of them), (3) every knot / reciprocal of all
// variables is requested before the first recursion level runs, (4) the coefficient window
// of the next dependent variable is requested before the current one is contracted.
// All LDS table reads are explicit ds_read_b64 / ds_read_b32 (see bsk_tile.hpp for why).
// Surfaces of order 2 / 4 use the bank-conflict-aware variant of this kernel, bsk_rowrot.hpp.
#pragma once
#include "bsk_tile.hpp"

namespace bsk {

// s_waitcnt lgkmcnt(min(CNT, 15)) tied to the first N values of v.  Clamping only makes the
// wait stronger, never weaker.
template <int CNT, int N, typename T, int CAP>
__device__ __forceinline__ void lds_wait_c(T (&v)[CAP])
{
    lds_wait_n<(CNT > 15 ? 15 : CNT), N>(v);
}

// Tie every element of a [R][C] block to the wait that precedes this call (empty asm: no
// instruction, only a scheduling dependency), so no consumer can be hoisted above the wait.
template <typename T, int C>
__device__ __forceinline__ void lds_tie_row(T (&v)[C])
{
    static_assert(C >= 1 && C <= 8, "row length");
    if constexpr (C == 1) asm volatile("" : "+v"(v[0]) :: "memory");
    else if constexpr (C == 2) asm volatile("" : "+v"(v[0]), "+v"(v[1]) :: "memory");
    else if constexpr (C == 3) asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]) :: "memory");
    else if constexpr (C == 4) asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]) :: "memory");
    else if constexpr (C == 5) asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]) :: "memory");
    else if constexpr (C == 6) asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]) :: "memory");
    else if constexpr (C == 7)
        asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]) :: "memory");
    else
        asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) :: "memory");
}

template <typename T, int R, int C, int A = 0>
__device__ __forceinline__ void lds_tie_block(T (&c)[R][C])
{
    if constexpr (A < R) {
        lds_tie_row<T, C>(c[A]);
        lds_tie_block<T, R, C, A + 1>(c);
    }
}

template <typename T, int R, int C>
__device__ __forceinline__ void block_issue(unsigned addr, unsigned rstride, T (&c)[R][C])
{
#pragma unroll
    for (int a = 0; a < R; ++a) lds_issue_n<T, C, C>(addr + a * rstride, c[a]);
}

// Wait for a whole block of which YOUNGER reads were issued afterwards, then tie it.
template <int YOUNGER, typename T, int R, int C>
__device__ __forceinline__ void block_wait(T (&c)[R][C])
{
    lds_wait_c<YOUNGER, C>(c[0]);
    lds_tie_block<T, R, C, 1>(c);
}

template <typename T, int O>
__device__ __forceinline__ T slab_fma(const T (&c)[O][O], const T (&b_out)[O], const T (&b_in)[O])
{
    T acc = T(0);
#pragma unroll
    for (int a = 0; a < O; ++a) {
        T t = T(0);
#pragma unroll
        for (int k = 0; k < O; ++k) t += c[a][k] * b_in[k];
        acc += t * b_out[a];
    }
    return acc;
}

template <typename T, int O>
__device__ __forceinline__ T row_fma(const T (&c)[1][O], const T (&b)[O])
{
    T acc = T(0);
#pragma unroll
    for (int k = 0; k < O; ++k) acc += c[0][k] * b[k];
    return acc;
}

// Span search of all variables in lock step through the bucket tables.
template <typename T, int NIND>
__device__ __forceinline__ void find_spans(unsigned tab_a, unsigned lut_a, const Desc<T> &d, const TileDesc<T> &td,
                                           int steps, const T (&u)[NIND], int (&ix)[NIND])
{
    unsigned e[NIND];
#pragma unroll
    for (int iv = 0; iv < NIND; ++iv) {
        int b = (int)((u[iv] - d.lo[iv]) * td.lut_scale[iv]);
        b = b < 0 ? 0 : (b > td.lut_m[iv] - 1 ? td.lut_m[iv] - 1 : b);
        asm volatile("ds_read_b32 %0, %1" : "=v"(e[iv]) : "v"(lut_a + 4u * (unsigned)(td.lut_off[iv] + b)) : "memory");
    }
    lds_wait_n<0, NIND>(e);
    int l[NIND], h[NIND];
#pragma unroll
    for (int iv = 0; iv < NIND; ++iv) { l[iv] = (int)(e[iv] & 0xffffu); h[iv] = (int)(e[iv] >> 16); }
    for (int s = 0; s < steps; ++s) {
        T km[NIND];
#pragma unroll
        for (int iv = 0; iv < NIND; ++iv)
            km[iv] = LdsRead<T>::template at<0>(tab_a + (unsigned)(d.off[iv] + ((l[iv] + h[iv]) >> 1)) * (unsigned)sizeof(T));
        lds_wait_n<0, NIND>(km);
#pragma unroll
        for (int iv = 0; iv < NIND; ++iv) {
            const int mid = (l[iv] + h[iv]) >> 1;
            const bool open = l[iv] < h[iv];
            const bool right = open && (km[iv] <= u[iv]);
            const bool left = open && !right;
            l[iv] = right ? mid + 1 : l[iv];
            h[iv] = left ? mid : h[iv];
        }
    }
#pragma unroll
    for (int iv = 0; iv < NIND; ++iv) ix[iv] = (u[iv] != u[iv]) ? d.ncoef[iv] : l[iv];
}

// Levels of one variable when AFTER more reads (of later variables) follow its own.
template <typename T, int O, int AFTER, bool DERIV, int D>
__device__ __forceinline__ void basis_levels_c(T u, int wrt, const T (&kn)[O], T (&rc)[O][O], T (&b)[O])
{
    if constexpr (D < O) {
        constexpr int younger = (O * (O - 1) - D * (D + 1)) / 2 + AFTER;
        lds_wait_c<younger, D>(rc[D]);
        if (!DERIV || D < O - wrt) {
            if constexpr (D == 1) {
                // first level on b = (0, .., 0, 1): written out, hipcc may not fold 0 + x * 1
                const T alpha = (u - kn[O - 2]) * rc[1][0];
                b[O - 2] = T(1) - alpha;
                b[O - 1] = alpha;
            } else {
#pragma unroll
                for (int j = 0; j < D; ++j) {
                    const int bi = O - D + j;
                    const T alpha = (u - kn[(O - 1) - D + j]) * rc[D][j];
                    b[bi - 1] += (T(1) - alpha) * b[bi];
                    b[bi] *= alpha;
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < D; ++j) {
                const int bi = O - D + j;
                const T alpha = T(D) * rc[D][j];
                b[bi - 1] -= alpha * b[bi];
                b[bi] *= alpha;
            }
        }
        basis_levels_c<T, O, AFTER, DERIV, D + 1>(u, wrt, kn, rc, b);
    }
}

template <typename T, int NIND, int O, bool DERIV, int IV>
__device__ __forceinline__ void bases_compute(const T (&u)[NIND], const Wrt &wrt, T (&kn)[NIND][O],
                                              T (&rc)[NIND][O][O], T (&b)[NIND][O])
{
    if constexpr (IV < NIND) {
        constexpr int per_axis = (O - 1) + (O * (O - 1)) / 2;
        constexpr int after = (NIND - 1 - IV) * per_axis;
#pragma unroll
        for (int k = 0; k < O; ++k) b[IV][k] = T(0);
        b[IV][O - 1] = T(1);
        if constexpr (O > 1) {
            lds_wait_c<(O * (O - 1)) / 2 + after, O - 1>(kn[IV]);
            basis_levels_c<T, O, after, DERIV, 1>(u[IV], DERIV ? wrt.w[IV] : 0, kn[IV], rc[IV], b[IV]);
        }
        if (DERIV && wrt.w[IV] >= O) {
#pragma unroll
            for (int k = 0; k < O; ++k) b[IV][k] = T(0);
        }
        bases_compute<T, NIND, O, DERIV, IV + 1>(u, wrt, kn, rc, b);
    }
}

template <typename T, int NIND, int O, bool DERIV>
__device__ __forceinline__ void bases_all(unsigned tab_a, const Desc<T> &d, const int (&ix)[NIND], const T (&u)[NIND],
                                          const Wrt &wrt, T (&b)[NIND][O])
{
    T kn[NIND][O];
    T rc[NIND][O][O];
    if constexpr (O > 1) {
#pragma unroll
        for (int iv = 0; iv < NIND; ++iv) {
            const unsigned ta = tab_a + (unsigned)d.off[iv] * (unsigned)sizeof(T);
            lds_issue_n<T, O - 1, O>(ta + (unsigned)(ix[iv] - (O - 1)) * (unsigned)sizeof(T), kn[iv]);
            basis_issue<T, O, 1>(ta, d.nk[iv], ix[iv], rc[iv]);
        }
    }
    bases_compute<T, NIND, O, DERIV, 0>(u, wrt, kn, rc, b);
}

// -------------------------------------------------------------------------------------
// Helpers of the rank rotation used by bsk_rowrot.hpp (orders 2 and 4): separately rounded
// product / sum for the rotation-invariant tree, and the rotation of a basis row by rank.
// -------------------------------------------------------------------------------------
// Separately rounded product / sum: `#pragma clang fp contract(off)` keeps hipcc (default
// -ffp-contract=fast) from fusing them into FMAs, which would break the symmetry of the tree.
template <typename T>
__device__ __forceinline__ T mul_rn(T a, T b)
{
#pragma clang fp contract(off)
    return a * b;
}
template <typename T>
__device__ __forceinline__ T add_rn(T a, T b)
{
#pragma clang fp contract(off)
    return a + b;
}

// br[j] = b[(j + rho) mod O] (values only; see rotate_basis for the select staging)
template <typename T, int O>
__device__ __forceinline__ void rotate_basis_values(const T (&b)[O], int rho, T (&br)[O])
{
    static_assert(O == 2 || O == 4, "rank rotation covers orders 2 and 4");
    int r0i = rho & 1, r1i = rho & 2;
    asm volatile("" : "+v"(r0i), "+v"(r1i));
    const bool r0 = r0i != 0, r1 = r1i != 0;
    if constexpr (O == 2) {
        T v0 = b[0], v1 = b[1];
        asm volatile("" : "+v"(v0), "+v"(v1));
        br[0] = r0 ? v1 : v0;
        br[1] = r0 ? v0 : v1;
    } else {
        T v0 = b[0], v1 = b[1], v2 = b[2], v3 = b[3];
        asm volatile("" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
        const T t0 = r0 ? v1 : v0, t1 = r0 ? v2 : v1, t2 = r0 ? v3 : v2, t3 = r0 ? v0 : v3;
        br[0] = r1 ? t2 : t0;
        br[1] = r1 ? t3 : t1;
        br[2] = r1 ? t0 : t2;
        br[3] = r1 ? t1 : t3;
    }
}

// -------------------------------------------------------------------------------------
// DERIV = false: plain evaluation (every derivative order zero), no per-level branches.
constexpr int STREAM_BLOCK = TILE;  // 16 waves per CU (4 per SIMD, 128 VGPRs)

template <typename T, int NIND, int O, bool DERIV>
__global__ __launch_bounds__(STREAM_BLOCK) void eval_stream(const Desc<T> d, const TileDesc<T> td,
                                                    const T *__restrict__ gtab, const unsigned *__restrict__ glut,
                                                    const T *__restrict__ gcoef, const Params<T> prm,
                                                    const long long N, T *__restrict__ out, const long long ostride,
                                                    const Wrt wrt, unsigned long long *bad)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned tab_a = (unsigned)(size_t)smem;
    const unsigned lut_a = tab_a + td.tab_bytes;
    const unsigned coef_a = lut_a + td.lut_bytes;
    {
        T *stab = reinterpret_cast<T *>(smem);
        unsigned *slut = reinterpret_cast<unsigned *>(smem + td.tab_bytes);
        T *scoef = reinterpret_cast<T *>(smem + td.tab_bytes + td.lut_bytes);
        for (int i = threadIdx.x; i < d.tab_len; i += blockDim.x) stab[i] = gtab[i];
        for (int i = threadIdx.x; i < td.lut_len; i += blockDim.x) slut[i] = glut[i];
        for (int i = threadIdx.x; i < d.coef_len; i += blockDim.x) scoef[i] = gcoef[i];
    }
    __syncthreads();

    int steps = 0;
#pragma unroll
    for (int iv = 0; iv < NIND; ++iv) steps = td.lut_steps[iv] > steps ? td.lut_steps[iv] : steps;
    const unsigned dstride = (unsigned)d.cstride[0] * (unsigned)sizeof(T);
    const long long stride = (long long)gridDim.x * blockDim.x;
    long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x;

    // Next iteration's parameters are fetched one iteration ahead.  The fallback value lives in a
    // register: `cond ? prm[i] : d.lo[iv]` lets hipcc select between the two ADDRESSES and emit
    // a flat_load (generic address space) with s_waitcnt vmcnt(0) lgkmcnt(0).  Deeper prefetch
    // rings were measured slower (register moves; 16 waves already hide the HBM latency).
    T lo_r[NIND];
#pragma unroll
    for (int iv = 0; iv < NIND; ++iv) lo_r[iv] = d.lo[iv];
    T un[NIND];
#pragma unroll
    for (int iv = 0; iv < NIND; ++iv) {
        un[iv] = lo_r[iv];
        if (n < N) un[iv] = prm.p[iv][n];
    }
    // no load may be pending at the loop header (hipcc would then await every prefetch right after issuing it)
#pragma unroll
    for (int iv = 0; iv < NIND; ++iv) asm volatile("" : "+v"(un[iv]));

    for (; n < N; n += stride) {
        T u[NIND];
        bool outside = false;
#pragma unroll
        for (int iv = 0; iv < NIND; ++iv) {
            u[iv] = un[iv];
            outside |= (u[iv] < lo_r[iv]) | (u[iv] > d.hi[iv]);
            un[iv] = lo_r[iv];
        }
        if (n + stride < N) {
#pragma unroll
            for (int iv = 0; iv < NIND; ++iv) un[iv] = prm.p[iv][n + stride];
        }
        if (outside) record_bad(bad, n);

        int ix[NIND];
        find_spans<T, NIND>(tab_a, lut_a, d, td, steps, u, ix);
        T b[NIND][O];
        bases_all<T, NIND, O, DERIV>(tab_a, d, ix, u, wrt, b);
        // Take the prefetched parameters HERE, before this iteration's stores are issued: the wait
        // hipcc inserts then covers the loads only (issued half an iteration ago).  At the end of
        // the iteration it would be vmcnt(0) behind a run-time number of stores, i.e. every
        // iteration would wait for its own stores to be acknowledged.
#pragma unroll
        for (int iv = 0; iv < NIND; ++iv) asm volatile("" : "+v"(un[iv]));
        unsigned caddr = coef_a;
#pragma unroll
        for (int iv = 0; iv < NIND; ++iv) caddr += (unsigned)((ix[iv] - O) * d.cstride[iv + 1]) * (unsigned)sizeof(T);

        if constexpr (NIND <= 2) {
            // one block (row or slab) per dependent variable
            constexpr int R = NIND == 1 ? 1 : O;
            const unsigned rstride = NIND == 1 ? 0u : (unsigned)d.cstride[1] * (unsigned)sizeof(T);
            for (int dep = 0; dep < d.nDep; ++dep) {
                T c[R][O];
                block_issue<T, R, O>(caddr, rstride, c);
                block_wait<0>(c);
                T r;
                if constexpr (NIND == 1) r = row_fma<T, O>(c, b[0]);
                else r = slab_fma<T, O>(c, b[0], b[1]);
                nt_store(&out[dep * ostride + n], r);
                caddr += dstride;
            }
        } else {
            // three variables: O slabs per dependent variable.  Single buffered: an asm read's
            // destination must never be spilled between issue and wait, so the live window
            // stays at one slab (O * O values).
            const unsigned s0 = (unsigned)d.cstride[1] * (unsigned)sizeof(T);
            const unsigned s1 = (unsigned)d.cstride[2] * (unsigned)sizeof(T);
            for (int dep = 0; dep < d.nDep; ++dep) {
                T acc = T(0);
#pragma unroll
                for (int a = 0; a < O; ++a) {
                    T c[O][O];
                    block_issue<T, O, O>(caddr + (unsigned)a * s0, s1, c);
                    block_wait<0>(c);
                    acc += b[0][a] * slab_fma<T, O>(c, b[1], b[2]);
                }
                nt_store(&out[dep * ostride + n], acc);
                caddr += dstride;
            }
        }
    }
}

// -------------------------------------------------------------------------------------
// jac_stream: fused jacobian on the LDS table image (the eval_stream of bsk_jacobian).
// One span search and one recursion per variable give the value basis b and the
// first-derivative basis db (they share every level but the last; the reference repeats the
// whole evaluation nInd times, bspy/_spline_evaluation.py:205-213); each coefficient window
// is read once and contracted against both.  out[(dep * NIND + j) * N + n]
// -------------------------------------------------------------------------------------
// Value levels D .. LAST-1 of one variable (counted waits as basis_levels_c).
template <typename T, int O, int AFTER, int D, int LAST>
__device__ __forceinline__ void basis_levels_upto(T u, const T (&kn)[O], T (&rc)[O][O], T (&b)[O])
{
    if constexpr (D < LAST) {
        constexpr int younger = (O * (O - 1) - D * (D + 1)) / 2 + AFTER;
        lds_wait_c<younger, D>(rc[D]);
#pragma unroll
        for (int j = 0; j < D; ++j) {
            const int bi = O - D + j;
            const T alpha = (u - kn[(O - 1) - D + j]) * rc[D][j];
            b[bi - 1] += (T(1) - alpha) * b[bi];
            b[bi] *= alpha;
        }
        basis_levels_upto<T, O, AFTER, D + 1, LAST>(u, kn, rc, b);
    }
}

template <typename T, int NIND, int O, int IV>
__device__ __forceinline__ void bases_d1_compute(const T (&u)[NIND], T (&kn)[NIND][O], T (&rc)[NIND][O][O],
                                                 T (&b)[NIND][O], T (&db)[NIND][O])
{
    if constexpr (IV < NIND) {
        constexpr int per_axis = (O - 1) + (O * (O - 1)) / 2;
        constexpr int after = (NIND - 1 - IV) * per_axis;
#pragma unroll
        for (int k = 0; k < O; ++k) { b[IV][k] = T(0); db[IV][k] = T(0); }
        b[IV][O - 1] = T(1);
        if constexpr (O > 1) {
            lds_wait_c<(O * (O - 1)) / 2 + after, O - 1>(kn[IV]);
            // levels 1 .. O-2 are shared by b and db; the last level is run twice
            T lvl[O];
#pragma unroll
            for (int k = 0; k < O; ++k) lvl[k] = b[IV][k];
            basis_levels_upto<T, O, after, 1, O - 1>(u[IV], kn[IV], rc[IV], lvl);      // order O-1 basis
#pragma unroll
            for (int k = 0; k < O; ++k) { b[IV][k] = lvl[k]; db[IV][k] = lvl[k]; }
            constexpr int D = O - 1;
            lds_wait_c<after, D>(rc[IV][D]);
#pragma unroll
            for (int j = 0; j < D; ++j) {
                const int bi = O - D + j;
                const T r = rc[IV][D][j];
                const T alpha = (u[IV] - kn[IV][(O - 1) - D + j]) * r;
                const T dalpha = T(D) * r;
                b[IV][bi - 1] += (T(1) - alpha) * b[IV][bi];
                b[IV][bi] *= alpha;
                db[IV][bi - 1] -= dalpha * db[IV][bi];
                db[IV][bi] *= dalpha;
            }
        }
        bases_d1_compute<T, NIND, O, IV + 1>(u, kn, rc, b, db);
    }
}

template <typename T, int NIND, int O>
__global__ __launch_bounds__(STREAM_BLOCK) void jac_stream(const Desc<T> d, const TileDesc<T> td,
                                                           const T *__restrict__ gtab, const unsigned *__restrict__ glut,
                                                           const T *__restrict__ gcoef, const Params<T> prm,
                                                           const long long N, T *__restrict__ out,
                                                           unsigned long long *bad)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned tab_a = (unsigned)(size_t)smem;
    const unsigned lut_a = tab_a + td.tab_bytes;
    const unsigned coef_a = lut_a + td.lut_bytes;
    {
        T *stab = reinterpret_cast<T *>(smem);
        unsigned *slut = reinterpret_cast<unsigned *>(smem + td.tab_bytes);
        T *scoef = reinterpret_cast<T *>(smem + td.tab_bytes + td.lut_bytes);
        for (int i = threadIdx.x; i < d.tab_len; i += blockDim.x) stab[i] = gtab[i];
        for (int i = threadIdx.x; i < td.lut_len; i += blockDim.x) slut[i] = glut[i];
        for (int i = threadIdx.x; i < d.coef_len; i += blockDim.x) scoef[i] = gcoef[i];
    }
    __syncthreads();
    int steps = 0;
#pragma unroll
    for (int iv = 0; iv < NIND; ++iv) steps = td.lut_steps[iv] > steps ? td.lut_steps[iv] : steps;
    const unsigned dstride = (unsigned)d.cstride[0] * (unsigned)sizeof(T);
    const long long stride = (long long)gridDim.x * blockDim.x;
    long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    T lo_r[NIND];
#pragma unroll
    for (int iv = 0; iv < NIND; ++iv) lo_r[iv] = d.lo[iv];
    T un[NIND];
#pragma unroll
    for (int iv = 0; iv < NIND; ++iv) { un[iv] = lo_r[iv]; if (n < N) un[iv] = prm.p[iv][n]; }
#pragma unroll
    for (int iv = 0; iv < NIND; ++iv) asm volatile("" : "+v"(un[iv]));   // see eval_stream

    for (; n < N; n += stride) {
        T u[NIND];
        bool outside = false;
#pragma unroll
        for (int iv = 0; iv < NIND; ++iv) {
            u[iv] = un[iv];
            outside |= (u[iv] < lo_r[iv]) | (u[iv] > d.hi[iv]);
            un[iv] = lo_r[iv];
        }
        if (n + stride < N) {
#pragma unroll
            for (int iv = 0; iv < NIND; ++iv) un[iv] = prm.p[iv][n + stride];
        }
        if (outside) record_bad(bad, n);

        int ix[NIND];
        find_spans<T, NIND>(tab_a, lut_a, d, td, steps, u, ix);
        T b[NIND][O], db[NIND][O];
        {
            T kn[NIND][O];
            T rc[NIND][O][O];
            if constexpr (O > 1) {
#pragma unroll
                for (int iv = 0; iv < NIND; ++iv) {
                    const unsigned ta = tab_a + (unsigned)d.off[iv] * (unsigned)sizeof(T);
                    lds_issue_n<T, O - 1, O>(ta + (unsigned)(ix[iv] - (O - 1)) * (unsigned)sizeof(T), kn[iv]);
                    basis_issue<T, O, 1>(ta, d.nk[iv], ix[iv], rc[iv]);
                }
            }
            bases_d1_compute<T, NIND, O, 0>(u, kn, rc, b, db);
        }
#pragma unroll
        for (int iv = 0; iv < NIND; ++iv) asm volatile("" : "+v"(un[iv]));   // see eval_stream: before the stores
        unsigned caddr = coef_a;
#pragma unroll
        for (int iv = 0; iv < NIND; ++iv) caddr += (unsigned)((ix[iv] - O) * d.cstride[iv + 1]) * (unsigned)sizeof(T);

        for (int dep = 0; dep < d.nDep; ++dep) {
            T *o = out + (long long)dep * NIND * N + n;
            if constexpr (NIND == 1) {
                T c[1][O];
                block_issue<T, 1, O>(caddr, 0u, c);
                block_wait<0>(c);
                nt_store(&o[0], row_fma<T, O>(c, db[0]));
            } else if constexpr (NIND == 2) {
                T c[O][O];
                block_issue<T, O, O>(caddr, (unsigned)d.cstride[1] * (unsigned)sizeof(T), c);
                block_wait<0>(c);
                T j0 = T(0), j1 = T(0);
#pragma unroll
                for (int a = 0; a < O; ++a) {
                    T t = T(0), tdv = T(0);
#pragma unroll
                    for (int k = 0; k < O; ++k) { t += c[a][k] * b[1][k]; tdv += c[a][k] * db[1][k]; }
                    j0 += t * db[0][a];
                    j1 += tdv * b[0][a];
                }
                nt_store(&o[0], j0);
                nt_store(&o[N], j1);
            } else {
                const unsigned s0 = (unsigned)d.cstride[1] * (unsigned)sizeof(T);
                const unsigned s1 = (unsigned)d.cstride[2] * (unsigned)sizeof(T);
                T j0 = T(0), j1 = T(0), j2 = T(0);
#pragma unroll
                for (int a = 0; a < O; ++a) {
                    T c[O][O];
                    block_issue<T, O, O>(caddr + (unsigned)a * s0, s1, c);
                    block_wait<0>(c);
                    T sv = T(0), sb = T(0), sc = T(0);
#pragma unroll
                    for (int k = 0; k < O; ++k) {
                        T t = T(0), tdv = T(0);
#pragma unroll
                        for (int m = 0; m < O; ++m) { t += c[k][m] * b[2][m]; tdv += c[k][m] * db[2][m]; }
                        sv += t * b[1][k];
                        sb += t * db[1][k];
                        sc += tdv * b[1][k];
                    }
                    j0 += sv * db[0][a];
                    j1 += sb * b[0][a];
                    j2 += sc * b[0][a];
                }
                nt_store(&o[0], j0);
                nt_store(&o[N], j1);
                nt_store(&o[2 * N], j2);
            }
            caddr += dstride;
        }
    }
}

// -------------------------------------------------------------------------------------
// probe_stream: memory-side floor of the evaluation kernels' launch geometry (one persistent
// workgroup per CU with `lds` bytes of LDS allocated): streams two parameter arrays in and
// three result arrays out, trivial arithmetic.  MODE 0: 8 bytes per lane per access (one
// point per lane), MODE 1: 16 bytes per lane (two adjacent points).  Diagnostic only
// (bsk_debug_probe); not part of the evaluation path.
// -------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(TILE) void probe_stream(const double *__restrict__ u, const double *__restrict__ v,
                                                     const long long N, double *__restrict__ out, const long long ostride)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (threadIdx.x == 0) smem[0] = 1;
    __syncthreads();
    const long long stride = (long long)gridDim.x * blockDim.x;
    if constexpr (MODE == 0) {
        for (long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += stride) {
            const double a = u[n], b = v[n];
            out[n] = a + b;
            out[ostride + n] = a - b;
            out[2 * ostride + n] = a * b;
        }
    } else {
        const long long np = N / 2;
        for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < np; q += stride) {
            const double2 a = *reinterpret_cast<const double2 *>(u + 2 * q), b = *reinterpret_cast<const double2 *>(v + 2 * q);
            double2 r0, r1, r2;
            r0.x = a.x + b.x; r0.y = a.y + b.y;
            r1.x = a.x - b.x; r1.y = a.y - b.y;
            r2.x = a.x * b.x; r2.y = a.y * b.y;
            *reinterpret_cast<double2 *>(out + 2 * q) = r0;
            *reinterpret_cast<double2 *>(out + ostride + 2 * q) = r1;
            *reinterpret_cast<double2 *>(out + 2 * ostride + 2 * q) = r2;
        }
    }
}

}  // namespace bsk
