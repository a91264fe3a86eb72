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
                                                    const Wrt wrt, unsigned long long *bad, const int dbg_arg)
{
    // dbg: timing-only ablations of the surface path (results wrong), compiled in only with
    // -DBSK_ABLATE: 1 no coefficient loads, 2 no products / sums, 4 no span search, 8 no
    // recursion, 16 no rank atomic, 32 no stores
#ifdef BSK_ABLATE
    const int dbg = dbg_arg;
#else
    constexpr int dbg = 0;
    (void)dbg_arg;
#endif
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

    // Register ring of the parameters of the next PF iterations: PF x 16 KB of loads in flight
    // per CU.  With a single load in flight per lane the kernel is bound by HBM latency
    // (16 KB per CU outstanding ~ 2 TB/s chip-wide), not by anything it computes.
    constexpr int PF = 1;
    // d.lo in registers: `cond ? prm[i] : d.lo[iv]` lets hipcc select between the two ADDRESSES
    // and emit one flat_load (generic address space), which forces s_waitcnt vmcnt(0)
    // lgkmcnt(0) every iteration - that serialisation, not LDS or HBM bandwidth, dominated
    // earlier versions of this kernel.
    T lo_r[NIND];
#pragma unroll
    for (int iv = 0; iv < NIND; ++iv) lo_r[iv] = d.lo[iv];
    T ring[PF][NIND];
#pragma unroll
    for (int k = 0; k < PF; ++k)
#pragma unroll
        for (int iv = 0; iv < NIND; ++iv) {
            ring[k][iv] = lo_r[iv];
            if (n + k * stride < N) ring[k][iv] = prm.p[iv][n + k * stride];
        }

    for (; n < N; n += stride) {
        T u[NIND];
        bool outside = false;
#pragma unroll
        for (int iv = 0; iv < NIND; ++iv) {
            u[iv] = ring[0][iv];
            outside |= (u[iv] < lo_r[iv]) | (u[iv] > d.hi[iv]);
        }
#pragma unroll
        for (int k = 0; k + 1 < PF; ++k)
#pragma unroll
            for (int iv = 0; iv < NIND; ++iv) ring[k][iv] = ring[k + 1][iv];
        {
            const long long nn = n + PF * stride;
#pragma unroll
            for (int iv = 0; iv < NIND; ++iv) ring[PF - 1][iv] = lo_r[iv];
            if (nn < N) {
#pragma unroll
                for (int iv = 0; iv < NIND; ++iv) ring[PF - 1][iv] = prm.p[iv][nn];
            }
        }
        if (outside) record_bad(bad, n);

        int ix[NIND];
        if (dbg & 4) {
#pragma unroll
            for (int iv = 0; iv < NIND; ++iv) ix[iv] = O + (int)(u[iv] * T(d.ncoef[iv] - O));
        } else {
            find_spans<T, NIND>(tab_a, lut_a, d, td, steps, u, ix);
        }
        T b[NIND][O];
        if (dbg & 8) {
#pragma unroll
            for (int iv = 0; iv < NIND; ++iv)
#pragma unroll
                for (int k = 0; k < O; ++k) b[iv][k] = u[iv] + T(k);
        } else {
            bases_all<T, NIND, O, DERIV>(tab_a, d, ix, u, wrt, b);
        }
        unsigned caddr = coef_a;
#pragma unroll
        for (int iv = 0; iv < NIND; ++iv) caddr += (unsigned)((ix[iv] - O) * d.cstride[iv + 1]) * (unsigned)sizeof(T);

        if constexpr (ROT) {
            int base = 0;
#pragma unroll
            for (int iv = 0; iv < NIND; ++iv) base += (ix[iv] - O) * d.cstride[iv + 1];
            int rho = base & (O - 1);
            if (!(dbg & 16)) {
                s_rc[lane] = 0u;
                rho = (int)atomicAdd(&s_rc[(lane & 32) + (base & 31)], 1u) & (O - 1);
            }
            T br[O];
            unsigned co[O];
            rotate_basis<T, O>(b[NIND - 1], rho, br, co);
            if constexpr (NIND == 2) {
                // The rotated column offsets are run-time values, so hipcc cannot pair these
                // reads into ds_read2_b64: plain loads, scheduled and waited for by the compiler
                // (no asm destinations live across other code).
                const unsigned rstride = (unsigned)d.cstride[1] * (unsigned)sizeof(T);
                const char *cw = smem + (caddr - tab_a);
                if (dbg) {
                    for (int dep = 0; dep < d.nDep; ++dep) {
                        T acc = T(0);
#pragma unroll
                        for (int a = 0; a < O; ++a) {
#pragma clang fp contract(off)
                            const char *row = cw + (unsigned)a * rstride;
                            T v[O];
#pragma unroll
                            for (int j = 0; j < O; ++j) {
                                if (dbg & 1) v[j] = br[j] + T(a + j);
                                else v[j] = *reinterpret_cast<const T *>(row + co[j]);
                            }
                            if (dbg & 2) {
#pragma unroll
                                for (int j = 0; j < O; ++j) asm volatile("" :: "v"(v[j]));
                                acc = v[0];
                            } else {
                                T t = T(0);
                                if constexpr (O == 4)
                                    t = add_rn<T>(add_rn<T>(mul_rn<T>(v[0], br[0]), mul_rn<T>(v[2], br[2])),
                                                  add_rn<T>(mul_rn<T>(v[1], br[1]), mul_rn<T>(v[3], br[3])));
                                else
                                    t = add_rn<T>(mul_rn<T>(v[0], br[0]), mul_rn<T>(v[1], br[1]));
                                acc = add_rn<T>(acc, mul_rn<T>(t, b[0][a]));
                            }
                        }
                        if (!(dbg & 32) || acc == T(12345.678)) out[dep * ostride + n] = acc;
                        cw += dstride;
                    }
                }
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
                if (dbg) {
                } else if (d.nDep == 3) {   // the common case, unrolled so loads run ahead of the sums
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
        // Consume the prefetched parameters HERE, at the end of the iteration: the wait the
        // compiler inserts is then vmcnt(<stores issued above>) and leaves this iteration's
        // stores in flight.  Left to the loop header it becomes vmcnt(0) (the counter state is
        // merged over the back edge), which exposed every store's latency once per iteration.
#pragma unroll
        for (int iv = 0; iv < NIND; ++iv) asm volatile("" : "+v"(ring[0][iv]));
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
    for (int iv = 0; iv < NIND; ++iv) {
        un[iv] = d.lo[iv];
        if (tile < ntiles && tile * TILE + tid < N) un[iv] = prm.p[iv][tile * TILE + tid];
    }

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
            for (int iv = 0; iv < NIND; ++iv) un[iv] = u[iv];          // any in-domain value
            if (nn < N) {
#pragma unroll
                for (int iv = 0; iv < NIND; ++iv) un[iv] = prm.p[iv][nn];
            }
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

// -------------------------------------------------------------------------------------
// eval_surface2: surfaces of order 2 or 4, table image in LDS, TWO ADJACENT points per lane
// and a DEPTH-deep register prefetch of the parameter stream.
//
// Why (ablations on MI355X, profiles/): with one 8-byte load in flight per lane a CU keeps
// only 16 KB of parameter loads outstanding; at ~2 us loaded HBM latency that caps the chip at
// ~2 TB/s - removing the whole coefficient contraction from eval_stream changed its time by
// 20 %, removing the loads' latency exposure is what matters.  Here every lane loads and stores
// 16 bytes per instruction (points 2i, 2i+1: the coalescing sweet spot) and runs DEPTH
// iterations ahead, i.e. DEPTH x 32 KB of loads in flight per CU.
// Per point the arithmetic is eval_stream's (rank rotation included): bitwise identical results.
// Requires 16-byte aligned parameter / result rows (checked by the launcher).
// -------------------------------------------------------------------------------------
template <typename T> struct Vec2;
template <> struct Vec2<double> { typedef double2 type; };
template <> struct Vec2<float> { typedef float2 type; };

constexpr int SURF2_BLOCK = 512;   // 8 waves per CU (2 per SIMD): two points per lane need > 128 VGPRs

template <typename T, int O, bool DERIV, int DEPTH>
__global__ __launch_bounds__(SURF2_BLOCK) void eval_surface2(const Desc<T> d, const TileDesc<T> td,
                                                              const T *__restrict__ gtab,
                                                              const unsigned *__restrict__ glut,
                                                              const T *__restrict__ gcoef, const Params<T> prm,
                                                              const long long N, T *__restrict__ out,
                                                              const long long ostride, const Wrt wrt,
                                                              unsigned long long *bad)
{
    static_assert(O == 2 || O == 4, "rank rotation covers orders 2 and 4");
    typedef typename Vec2<T>::type V2;
    constexpr int P = 2;
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
    unsigned *s_rc = reinterpret_cast<unsigned *>(smem + td.tab_bytes + td.lut_bytes + td.coef_bytes) + (threadIdx.x & ~63);
    const int lane = threadIdx.x & 63;

    const int steps = td.lut_steps[0] > td.lut_steps[1] ? td.lut_steps[0] : td.lut_steps[1];
    const unsigned dstride = (unsigned)d.cstride[0] * (unsigned)sizeof(T);
    const unsigned rstride = (unsigned)d.cstride[1] * (unsigned)sizeof(T);
    const long long npairs = (N + 1) / 2;
    const long long stride = (long long)gridDim.x * blockDim.x;
    long long pn = (long long)blockIdx.x * blockDim.x + threadIdx.x;       // pair index: points 2 pn, 2 pn + 1

    // the four "variables" of the lock-step search: (point 0: u, v), (point 1: u, v)
    Desc<T> d4 = d;
    TileDesc<T> td4 = td;
#pragma unroll
    for (int k = 2; k < 4; ++k) {
        d4.off[k] = d.off[k - 2]; d4.ncoef[k] = d.ncoef[k - 2]; d4.lo[k] = d.lo[k - 2];
        td4.lut_scale[k] = td.lut_scale[k - 2]; td4.lut_m[k] = td.lut_m[k - 2]; td4.lut_off[k] = td.lut_off[k - 2];
    }

    const T lo0 = d.lo[0], lo1 = d.lo[1];      // register copies: see eval_stream (no flat loads)
    auto fetch = [&](long long q, V2 (&dst)[2]) {
        dst[0].x = lo0; dst[0].y = lo0; dst[1].x = lo1; dst[1].y = lo1;
        if (2 * q + 1 < N) {
            dst[0] = *reinterpret_cast<const V2 *>(prm.p[0] + 2 * q);
            dst[1] = *reinterpret_cast<const V2 *>(prm.p[1] + 2 * q);
        } else if (2 * q < N) {
            dst[0].x = prm.p[0][2 * q];
            dst[1].x = prm.p[1][2 * q];
        }
    };
    V2 ring[DEPTH][2];
#pragma unroll
    for (int k = 0; k < DEPTH; ++k) fetch(pn + k * stride, ring[k]);

    for (; pn < npairs; pn += stride) {
        T u[2 * P];
        u[0] = ring[0][0].x; u[1] = ring[0][1].x;      // point 0: (u, v)
        u[2] = ring[0][0].y; u[3] = ring[0][1].y;      // point 1: (u, v)
#pragma unroll
        for (int k = 0; k + 1 < DEPTH; ++k) { ring[k][0] = ring[k + 1][0]; ring[k][1] = ring[k + 1][1]; }
        fetch(pn + DEPTH * stride, ring[DEPTH - 1]);

        bool valid[P];
#pragma unroll
        for (int p = 0; p < P; ++p) {
            valid[p] = 2 * pn + p < N;
            const bool outside = (u[2 * p] < d.lo[0]) | (u[2 * p] > d.hi[0]) | (u[2 * p + 1] < d.lo[1]) | (u[2 * p + 1] > d.hi[1]);
            if (valid[p] && outside) record_bad(bad, 2 * pn + p);
        }

        int ix[2 * P];
        find_spans<T, 2 * P>(tab_a, lut_a, d4, td4, steps, u, ix);

        T b[P][2][O];
        int rho[P];
        const char *cw[P];
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const int ixp[2] = {ix[2 * p], ix[2 * p + 1]};
            const T up[2] = {u[2 * p], u[2 * p + 1]};
            bases_all<T, 2, O, DERIV>(tab_a, d, ixp, up, wrt, b[p]);
            const int base = (ixp[0] - O) * d.cstride[1] + (ixp[1] - O);
            cw[p] = smem + (coef_a - tab_a) + (unsigned)base * (unsigned)sizeof(T);
            s_rc[lane] = 0u;
            rho[p] = (int)atomicAdd(&s_rc[(lane & 32) + (base & 31)], 1u) & (O - 1);
        }
        T br[P][O];
        unsigned co[P][O];
#pragma unroll
        for (int p = 0; p < P; ++p) rotate_basis<T, O>(b[p][1], rho[p], br[p], co[p]);

        auto one_dep = [&](int dep) {
            T res[P];
#pragma unroll
            for (int p = 0; p < P; ++p) {
                T acc = T(0);
#pragma unroll
                for (int a = 0; a < O; ++a) {
#pragma clang fp contract(off)
                    const char *row = cw[p] + (unsigned)a * rstride;
                    T t;
                    if constexpr (O == 2) {
                        const T p0 = mul_rn<T>(*reinterpret_cast<const T *>(row + co[p][0]), br[p][0]);
                        const T p1 = mul_rn<T>(*reinterpret_cast<const T *>(row + co[p][1]), br[p][1]);
                        t = add_rn<T>(p0, p1);
                    } else {
                        const T p0 = mul_rn<T>(*reinterpret_cast<const T *>(row + co[p][0]), br[p][0]);
                        const T p1 = mul_rn<T>(*reinterpret_cast<const T *>(row + co[p][1]), br[p][1]);
                        const T p2 = mul_rn<T>(*reinterpret_cast<const T *>(row + co[p][2]), br[p][2]);
                        const T p3 = mul_rn<T>(*reinterpret_cast<const T *>(row + co[p][3]), br[p][3]);
                        t = add_rn<T>(add_rn<T>(p0, p2), add_rn<T>(p1, p3));
                    }
                    acc = add_rn<T>(acc, mul_rn<T>(t, b[p][0][a]));
                }
                res[p] = acc;
                cw[p] += dstride;
            }
            T *o = out + dep * ostride + 2 * pn;
            if (valid[1]) { V2 v; v.x = res[0]; v.y = res[1]; *reinterpret_cast<V2 *>(o) = v; }
            else if (valid[0]) o[0] = res[0];
        };
        if (d.nDep == 3) {
            one_dep(0); one_dep(1); one_dep(2);
        } else {
            for (int dep = 0; dep < d.nDep; ++dep) one_dep(dep);
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
        unsigned caddr = coef_a;
#pragma unroll
        for (int iv = 0; iv < NIND; ++iv) caddr += (unsigned)((ix[iv] - O) * d.cstride[iv + 1]) * (unsigned)sizeof(T);

        for (int dep = 0; dep < d.nDep; ++dep) {
            T *o = out + (long long)dep * NIND * N + n;
            if constexpr (NIND == 1) {
                T c[1][O];
                block_issue<T, 1, O>(caddr, 0u, c);
                block_wait<0>(c);
                o[0] = row_fma<T, O>(c, db[0]);
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
                o[0] = j0;
                o[N] = j1;
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
                o[0] = j0;
                o[N] = j1;
                o[2 * N] = j2;
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
