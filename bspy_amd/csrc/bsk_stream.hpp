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
// Rank rotation (orders 2 and 4): lanes of a half-wave whose windows start in the same LDS
// bank class would hit the same bank at every read.  Each lane therefore walks the O columns
// of its window rows starting at column (rank mod O), rank = its index among the lanes of its
// half-wave with the same class (one LDS atomic on a per-wave counter row).  Equal-class
// lanes then sit on different banks at every step; Monte Carlo of the bank model: expected
// conflict multiplicity 3.57 -> 2.41 for random spans.  The products of a row are summed
// with a rotation-invariant tree ((p0 + p2) + (p1 + p3), products rounded separately), so
// the result does not depend on the rank: runs stay bitwise reproducible.
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

template <typename T, int O>
__device__ __forceinline__ void rotate_basis(const T (&b)[O], int rho, T (&br)[O], unsigned (&co)[O])
{
    static_assert(O == 2 || O == 4, "rank rotation covers orders 2 and 4");
    // the two rotate-by-1 / rotate-by-2 select stages; the conditions are laundered through
    // empty asm so hipcc keeps them as 2 x O selects instead of a dynamic register index
    int r0i = rho & 1, r1i = rho & 2;
    asm volatile("" : "+v"(r0i), "+v"(r1i));
    const bool r0 = r0i != 0, r1 = r1i != 0;
    if constexpr (O == 2) {
        T v0 = b[0], v1 = b[1];
        asm volatile("" : "+v"(v0), "+v"(v1));      // plain values, not elements of an indexable vector
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
#pragma unroll
    for (int j = 0; j < O; ++j) co[j] = (unsigned)((j + rho) & (O - 1)) * (unsigned)sizeof(T);
}

// Slab with rotated columns: c[a][j] = window[a][(j + rho) mod O]; br is the rotated basis.
template <typename T, int O>
__device__ __forceinline__ void slab_rot_issue(unsigned addr, unsigned rstride, const unsigned (&co)[O], T (&c)[O][O])
{
#pragma unroll
    for (int a = 0; a < O; ++a) {
        const unsigned ra = addr + (unsigned)a * rstride;
#pragma unroll
        for (int j = 0; j < O; ++j) c[a][j] = LdsRead<T>::template at<0>(ra + co[j]);
    }
}

template <typename T, int O>
__device__ __forceinline__ T slab_rot_sum(const T (&c)[O][O], const T (&b_out)[O], const T (&br)[O])
{
    T acc = T(0);
#pragma unroll
    for (int a = 0; a < O; ++a) {
#pragma clang fp contract(off)
        T t;
        if constexpr (O == 2) {
            t = add_rn<T>(mul_rn<T>(c[a][0], br[0]), mul_rn<T>(c[a][1], br[1]));
        } else {
            const T p0 = mul_rn<T>(c[a][0], br[0]), p1 = mul_rn<T>(c[a][1], br[1]);
            const T p2 = mul_rn<T>(c[a][2], br[2]), p3 = mul_rn<T>(c[a][3], br[3]);
            t = add_rn<T>(add_rn<T>(p0, p2), add_rn<T>(p1, p3));
        }
        acc = add_rn<T>(acc, mul_rn<T>(t, b_out[a]));
    }
    return acc;
}

// Half-slab units (O / 2 rows) for the software pipeline of the rotated contraction.
template <typename T, int O>
__device__ __forceinline__ void half_rot_issue(unsigned addr, unsigned rstride, const unsigned (&co)[O], T (&c)[O / 2][O])
{
#pragma unroll
    for (int a = 0; a < O / 2; ++a) {
        const unsigned ra = addr + (unsigned)a * rstride;
#pragma unroll
        for (int j = 0; j < O; ++j) c[a][j] = LdsRead<T>::template at<0>(ra + co[j]);
    }
}

template <typename T, int O, int A0>
__device__ __forceinline__ T half_rot_sum(T acc, const T (&c)[O / 2][O], const T (&b_out)[O], const T (&br)[O])
{
#pragma unroll
    for (int a = 0; a < O / 2; ++a) {
#pragma clang fp contract(off)
        T t;
        if constexpr (O == 2) {
            t = add_rn<T>(mul_rn<T>(c[a][0], br[0]), mul_rn<T>(c[a][1], br[1]));
        } else {
            const T p0 = mul_rn<T>(c[a][0], br[0]), p1 = mul_rn<T>(c[a][1], br[1]);
            const T p2 = mul_rn<T>(c[a][2], br[2]), p3 = mul_rn<T>(c[a][3], br[3]);
            t = add_rn<T>(add_rn<T>(p0, p2), add_rn<T>(p1, p3));
        }
        acc = add_rn<T>(acc, mul_rn<T>(t, b_out[A0 + a]));
    }
    return acc;
}

template <typename T, int O>
__device__ __forceinline__ T slab_rot(unsigned addr, unsigned rstride, const unsigned (&co)[O], const T (&b_out)[O],
                                      const T (&br)[O])
{
    T c[O][O];
    slab_rot_issue<T, O>(addr, rstride, co, c);
    block_wait<0>(c);
    return slab_rot_sum<T, O>(c, b_out, br);
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
    constexpr bool ROT = (NIND == 2) && (O == 2 || O == 4);   // (3 variables: the rotated slabs spill)
    // per-wave class counters of the rank rotation: [wave][half-wave][class]
    unsigned *s_rc = reinterpret_cast<unsigned *>(smem + td.tab_bytes + td.lut_bytes + td.coef_bytes) + (threadIdx.x & ~63);
    const int lane = threadIdx.x & 63;

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

        if constexpr (ROT) {
            int base = 0;
#pragma unroll
            for (int iv = 0; iv < NIND; ++iv) base += (ix[iv] - O) * d.cstride[iv + 1];
            s_rc[lane] = 0u;
            const int rho = (int)atomicAdd(&s_rc[(lane & 32) + (base & 31)], 1u) & (O - 1);
            T br[O];
            unsigned co[O];
            rotate_basis<T, O>(b[NIND - 1], rho, br, co);
            if constexpr (NIND == 2) {
                // The rotated column offsets are run-time values, so hipcc cannot pair these
                // reads into ds_read2_b64: plain loads, scheduled and waited for by the compiler
                // (no asm destinations live across other code).
                const unsigned rstride = (unsigned)d.cstride[1] * (unsigned)sizeof(T);
                const char *cw = smem + (caddr - tab_a);
                auto one_dep = [&](int dep) {
                    T acc = T(0);
#pragma unroll
                    for (int a = 0; a < O; ++a) {
#pragma clang fp contract(off)
                        const char *row = cw + (unsigned)a * rstride;
                        T t;
                        if constexpr (O == 2) {
                            const T p0 = mul_rn<T>(*reinterpret_cast<const T *>(row + co[0]), br[0]);
                            const T p1 = mul_rn<T>(*reinterpret_cast<const T *>(row + co[1]), br[1]);
                            t = add_rn<T>(p0, p1);
                        } else {
                            const T p0 = mul_rn<T>(*reinterpret_cast<const T *>(row + co[0]), br[0]);
                            const T p1 = mul_rn<T>(*reinterpret_cast<const T *>(row + co[1]), br[1]);
                            const T p2 = mul_rn<T>(*reinterpret_cast<const T *>(row + co[2]), br[2]);
                            const T p3 = mul_rn<T>(*reinterpret_cast<const T *>(row + co[3]), br[3]);
                            t = add_rn<T>(add_rn<T>(p0, p2), add_rn<T>(p1, p3));
                        }
                        acc = add_rn<T>(acc, mul_rn<T>(t, b[0][a]));
                    }
                    out[dep * ostride + n] = acc;
                    cw += dstride;
                };
                if (d.nDep == 3) {          // the common case, unrolled so loads run ahead of the sums
                    one_dep(0); one_dep(1); one_dep(2);
                } else {
                    for (int dep = 0; dep < d.nDep; ++dep) one_dep(dep);
                }
            } else {
                const unsigned s0 = (unsigned)d.cstride[1] * (unsigned)sizeof(T);
                const unsigned s1 = (unsigned)d.cstride[2] * (unsigned)sizeof(T);
                for (int dep = 0; dep < d.nDep; ++dep) {
                    T acc = T(0);
#pragma unroll
                    for (int a = 0; a < O; ++a) acc += b[0][a] * slab_rot<T, O>(caddr + (unsigned)a * s0, s1, co, b[1], br);
                    out[dep * ostride + n] = acc;
                    caddr += dstride;
                }
            }
        } else if constexpr (NIND <= 2) {
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

// -------------------------------------------------------------------------------------
// eval_perm: eval_stream with the points of every 1024-point tile re-assigned to lanes by
// the LDS bank class of their coefficient window (class = window offset mod 32).  Each
// half-wave then holds (at most) one point of every class, so its 32 lanes read 32 different
// bank pairs: the coefficient reads - 48 of the 70 table reads of a bicubic point - and the
// table reads of the last variable become conflict free.
//
// Two barriers per tile, everything else double buffered by tile parity p:
//   A  (lane = original point)  parameters (prefetched), span search, class, rank in class
//                               by LDS atomic; parameters and spans staged in LDS
//   -- barrier --
//   B  (every wave redundantly) class counts -> holes / overflow prefix by wave shuffles;
//                               point -> slot (rank * 32 + class), overflow points -> list
//   -- barrier --
//   C  (lane = slot)            fetch the assigned point, recursion, contraction, results
//                               stored straight to the point's own position (scattered
//                               8-byte stores inside the tile's 8 KB window per variable)
// Slots of short classes are filled with overflow points of long classes (those lanes may
// conflict; ~7 % of a random tile).
// -------------------------------------------------------------------------------------
template <typename T, int NIND, int O, bool DERIV>
__global__ __launch_bounds__(TILE) void eval_perm(const Desc<T> d, const TileDesc<T> td,
                                                  const T *__restrict__ gtab, const unsigned *__restrict__ glut,
                                                  const T *__restrict__ gcoef, const Params<T> prm,
                                                  const long long N, T *__restrict__ out, const long long ostride,
                                                  const Wrt wrt, unsigned long long *bad, const int dbg)
{
    // dbg (timing-only ablations, results wrong): 1 = coalesced stores to the slot's own index,
    // 2 = identity assignment (src = tid), 4 = no barriers (only with 2)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned tab_a = (unsigned)(size_t)smem;
    const unsigned lut_a = tab_a + td.tab_bytes;
    const unsigned coef_a = lut_a + td.lut_bytes;
    char *stage = smem + td.tab_bytes + td.lut_bytes + td.coef_bytes;
    T *s_u = reinterpret_cast<T *>(stage);                                          // [2][NIND][TILE]
    unsigned *s_ix = reinterpret_cast<unsigned *>(s_u + 2 * NIND * TILE);           // [2][TILE] packed spans
    unsigned short *s_perm = reinterpret_cast<unsigned short *>(s_ix + 2 * TILE);   // [2][TILE]
    unsigned short *s_ovf = s_perm + 2 * TILE;                                      // [2][TILE]
    int *s_cnt = reinterpret_cast<int *>(s_ovf + 2 * TILE);                         // [2][NCLASS]
    const int tid = threadIdx.x;
    {
        T *stab = reinterpret_cast<T *>(smem);
        unsigned *slut = reinterpret_cast<unsigned *>(smem + td.tab_bytes);
        T *scoef = reinterpret_cast<T *>(smem + td.tab_bytes + td.lut_bytes);
        for (int i = tid; i < d.tab_len; i += TILE) stab[i] = gtab[i];
        for (int i = tid; i < td.lut_len; i += TILE) slut[i] = glut[i];
        for (int i = tid; i < d.coef_len; i += TILE) scoef[i] = gcoef[i];
        if (tid < 2 * NCLASS) s_cnt[tid] = 0;
    }
    __syncthreads();

    int steps = 0;
#pragma unroll
    for (int iv = 0; iv < NIND; ++iv) steps = td.lut_steps[iv] > steps ? td.lut_steps[iv] : steps;
    const unsigned dstride = (unsigned)d.cstride[0] * (unsigned)sizeof(T);
    const long long ntiles = (N + TILE - 1) / TILE;
    const int lane32 = tid & (NCLASS - 1);
    const int row = tid >> 5;

    long long tile = blockIdx.x;
    T un[NIND];
#pragma unroll
    for (int iv = 0; iv < NIND; ++iv) un[iv] = (tile < ntiles && tile * TILE + tid < N) ? prm.p[iv][tile * TILE + tid] : d.lo[iv];

    for (int p = 0; tile < ntiles; tile += gridDim.x, p ^= 1) {
        const long long n0 = tile * TILE;
        const long long n = n0 + tid;
        const bool valid = n < N;
        T *su = s_u + p * NIND * TILE;
        unsigned *six = s_ix + p * TILE;
        unsigned short *sperm = s_perm + p * TILE;
        unsigned short *sovf = s_ovf + p * TILE;
        int *scnt = s_cnt + p * NCLASS;

        // ---- A: this lane's own point
        T u[NIND];
        bool outside = false;
#pragma unroll
        for (int iv = 0; iv < NIND; ++iv) {
            u[iv] = un[iv];
            outside |= (u[iv] < d.lo[iv]) | (u[iv] > d.hi[iv]);
        }
        {
            const long long nn = (tile + gridDim.x) * TILE + tid;
#pragma unroll
            for (int iv = 0; iv < NIND; ++iv) un[iv] = nn < N ? prm.p[iv][nn] : d.lo[iv];
        }
        if (valid && outside) record_bad(bad, n);
        int ix[NIND];
        find_spans<T, NIND>(tab_a, lut_a, d, td, steps, u, ix);
        int base = 0;
        unsigned packed = 0;
#pragma unroll
        for (int iv = 0; iv < NIND; ++iv) {
            base += (ix[iv] - O) * d.cstride[iv + 1];
            packed |= (unsigned)ix[iv] << (10 * iv);
        }
        const int cls = base & (NCLASS - 1);
        int rank = 0;
        if (valid) {
            rank = atomicAdd(&scnt[cls], 1);
#pragma unroll
            for (int iv = 0; iv < NIND; ++iv) su[iv * TILE + tid] = u[iv];
            six[tid] = packed;
        }
        if (!(dbg & 4)) __syncthreads();

        // ---- B: slot assignment (each wave computes the prefix sums for itself)
        const int cnt_c = scnt[lane32];
        const int hole_c = cnt_c < ROWS ? ROWS - cnt_c : 0;
        const int over_c = cnt_c > ROWS ? cnt_c - ROWS : 0;
        int hs = hole_c, os = over_c;
#pragma unroll
        for (int off = 1; off < NCLASS; off <<= 1) {
            const int h2 = __shfl_up(hs, off, NCLASS);
            const int o2 = __shfl_up(os, off, NCLASS);
            if (lane32 >= off) { hs += h2; os += o2; }
        }
        const int hole_ex = hs - hole_c;                       // holes of the classes before mine
        const int over_ex = os - over_c;
        const int total_over = __shfl(os, NCLASS - 1, NCLASS);
        const int my_over_ex = __shfl(over_ex, cls, NCLASS);   // overflow offset of MY POINT's class
        if (valid) {
            if (rank < ROWS) sperm[rank * NCLASS + cls] = (unsigned short)tid;
            else sovf[my_over_ex + rank - ROWS] = (unsigned short)tid;
        }
        if (tid < NCLASS) s_cnt[(p ^ 1) * NCLASS + tid] = 0;   // counters of the next tile
        if (!(dbg & 4)) __syncthreads();

        // ---- C: the point assigned to this slot (row = half-wave, column = class)
        int src;
        bool have;
        if (row < cnt_c) {
            src = sperm[tid];
            have = true;
        } else {
            const int k = hole_ex + (row - cnt_c);
            have = k < total_over;
            src = have ? (int)sovf[k] : 0;
        }
        if (dbg & 2) { src = tid; have = valid; }
        if (have) {
            const unsigned pk = six[src];
#pragma unroll
            for (int iv = 0; iv < NIND; ++iv) {
                u[iv] = su[iv * TILE + src];
                ix[iv] = (int)((pk >> (10 * iv)) & 1023u);
            }
            T b[NIND][O];
            bases_all<T, NIND, O, DERIV>(tab_a, d, ix, u, wrt, b);
            unsigned caddr = coef_a;
#pragma unroll
            for (int iv = 0; iv < NIND; ++iv) caddr += (unsigned)((ix[iv] - O) * d.cstride[iv + 1]) * (unsigned)sizeof(T);
            T *o = out + n0 + ((dbg & 1) ? tid : src);
            if constexpr (NIND <= 2) {
                constexpr int R = NIND == 1 ? 1 : O;
                const unsigned rstride = NIND == 1 ? 0u : (unsigned)d.cstride[1] * (unsigned)sizeof(T);
                for (int dep = 0; dep < d.nDep; ++dep) {
                    T c[R][O];
                    block_issue<T, R, O>(caddr, rstride, c);
                    block_wait<0>(c);
                    T r;
                    if constexpr (NIND == 1) r = row_fma<T, O>(c, b[0]);
                    else r = slab_fma<T, O>(c, b[0], b[1]);
                    o[dep * ostride] = r;
                    caddr += dstride;
                }
            } else {
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
                    o[dep * ostride] = acc;
                    caddr += dstride;
                }
            }
        }
    }
}

}  // namespace bsk
