// eval_stream: the LDS-table evaluation kernel tuned for latency hiding (gfx950).
//
// One lane = one point; table image (axis tables, span-search bucket tables, coefficients)
// in LDS; persistent workgroups.  Compared with eval_tile<PERM = false> the per-point
// dependency chain is cut: (1) the next tile's parameters are fetched from HBM before the
// current tile is evaluated, (2) the span searches of all variables advance in lock step
// (one LDS round trip per step for all of them), (3) every knot / reciprocal of all
// variables is requested before the first recursion level runs, (4) the coefficient window
// of the next dependent variable is requested before the current one is contracted.
// All LDS table reads are explicit ds_read_b64 / ds_read_b32 (see bsk_tile.hpp for why).
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
    static_assert(C >= 1 && C <= 6, "row length");
    if constexpr (C == 1) asm volatile("" : "+v"(v[0]) :: "memory");
    else if constexpr (C == 2) asm volatile("" : "+v"(v[0]), "+v"(v[1]) :: "memory");
    else if constexpr (C == 3) asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]) :: "memory");
    else if constexpr (C == 4) asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]) :: "memory");
    else if constexpr (C == 5) asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]) :: "memory");
    else asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]) :: "memory");
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
#pragma unroll
            for (int j = 0; j < D; ++j) {
                const int bi = O - D + j;
                const T alpha = (u - kn[(O - 1) - D + j]) * rc[D][j];
                b[bi - 1] += (T(1) - alpha) * b[bi];
                b[bi] *= alpha;
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
// DERIV = false: plain evaluation (every derivative order zero), no per-level branches.
template <typename T, int NIND, int O, bool DERIV>
__global__ __launch_bounds__(TILE) void eval_stream(const Desc<T> d, const TileDesc<T> td,
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

    T un[NIND];
#pragma unroll
    for (int iv = 0; iv < NIND; ++iv) un[iv] = n < N ? prm.p[iv][n] : d.lo[iv];

    for (; n < N; n += stride) {
        T u[NIND];
        bool outside = false;
#pragma unroll
        for (int iv = 0; iv < NIND; ++iv) {
            u[iv] = un[iv];
            outside |= (u[iv] < d.lo[iv]) | (u[iv] > d.hi[iv]);
        }
        {   // next tile's parameters: in flight while this tile is evaluated
            const long long nn = n + stride;
#pragma unroll
            for (int iv = 0; iv < NIND; ++iv) un[iv] = nn < N ? prm.p[iv][nn] : d.lo[iv];
        }
        if (outside) record_bad(bad, n);

        int ix[NIND];
        find_spans<T, NIND>(tab_a, lut_a, d, td, steps, u, ix);
        T b[NIND][O];
        bases_all<T, NIND, O, DERIV>(tab_a, d, ix, u, wrt, b);
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
                out[dep * ostride + n] = r;
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
                out[dep * ostride + n] = acc;
                caddr += dstride;
            }
        }
    }
}

}  // namespace bsk
