// Uniform-knot front end of the LDS-resident surface kernels (eval_uni / jac_uni / curv_uni).
//
// A knot vector whose domain knots are equally spaced - every BSpy constructor that is not handed
// explicit knots builds one (clamped ends, interior from linspace) - needs no tables for the
// Cox-de Boor recursion (reference bspy/_spline_evaluation.py:11-26): in the local coordinate
// z = (u - t_m) / h of span m every knot difference is an integer multiple of h, so
//     alpha = (u - knots[i]) / (knots[i + D] - knots[i]) = (z + (D - 1 - j)) / D          (value levels)
//     alpha = D / (knots[i + D] - knots[i])               = 1 / h                          (derivative levels)
// with compile-time constants.  The kernels of bsk_rowrot.hpp spend 22 of their 70 LDS reads per
// point on the span search and the knot / reciprocal records; here the span costs ONE LDS read per
// variable (the stored knot t[m0 + 1], which makes the span decision exactly the reference's
// searchsorted(knots, u, 'right') and gives z without cancellation) and the recursion none.
//
// Clamped ends: the first / last order-1 basis functions of a clamped knot vector differ from the
// uniform ones.  The host re-expresses the spline ONCE, at upload, in the uniform basis
// ("unclamping", Piegl & Tiller A12.1: only the first / last order-1 control points of each
// variable change, by a fixed (order-1) x (order-1) matrix computed in extended precision): the
// function is the same on the whole domain, every span then has the same basis polynomials, and
// the kernels never look at the boundary.  The caller's coefficients are untouched; the unclamped
// copy lives in the LDS image of this path only (bsk_api.hip: build_uniform_image).
//
// LDS image (built on the host in exactly this layout, staged by a linear copy):
//   [domain knots of variable 0: ns0 + 1][variable 1: ns1 + 1] pad16
//   [coefficients: control-point major [i0][i1][dep] when nDep <= 3 (every window read of every dependent
//    variable is then one of O row addresses plus a compile-time offset), else [dep][i0][i1];
//    row stride = 32 / O (mod 32) elements] pad1024 [rank counters: not in the global image]
// Row rotation (bsk_rowrot.hpp) on this stride: stepping to the next window row moves 32 / O banks, so
// the lanes whose windows start in classes equal mod 32 / O share O banks; each lane starts at the row
// that puts it (rank among those lanes) banks into that set - a conflict-free schedule whenever such a
// group holds at most O lanes of a half-wave (Monte Carlo multiplicity 2.1 against 2.4 for the odd stride).
// The rotation-invariant summation tree keeps results independent of the rank.
//
// Measured on 10 M random points, cfg2 (gpurun_out/r2b): general kernels 107-113 us,
// this front end 91-94, closed-form basis 88, control-point-major image 87.5, grouped rotation 86-87.
// Measured without gain and removed: domain knots recomputed in registers instead of read from LDS,
// window rows awaited one by one with the next dependent variable's row requested at once (12-16 reads
// in flight during the multiply-adds), image staged by LDS-DMA.
#pragma once
#include "bsk_rowrot.hpp"

namespace bsk {

// Linear global -> LDS copy of the prebuilt image, all loads of a lane in flight together.
__device__ __forceinline__ void stage_linear(char *smem, const void *__restrict__ gimg, unsigned bytes)
{
    const uint4 *__restrict__ g = static_cast<const uint4 *>(gimg);
    uint4 *s = reinterpret_cast<uint4 *>(smem);
    const unsigned n16 = bytes >> 4;
    constexpr int U = 8;
    const unsigned bd = blockDim.x;
    for (unsigned i0 = threadIdx.x; i0 < n16; i0 += bd * U) {
        uint4 v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) v[k] = g[min(i0 + k * bd, n16 - 1u)];
#pragma unroll
        for (int k = 0; k < U; ++k) { const unsigned i = i0 + k * bd; if (i < n16) s[i] = v[k]; }
    }
}

// Spans and local coordinates of both variables.
//   m0 = trunc((u - lo) / h - eps) clamped to [0, ns - 1] is the span or the one before it (eps
//   exceeds the rounding of the estimate and is far below 1); the stored knot t[m0 + 1] decides:
//   u >= t[m0 + 1] moves to the next span, exactly as searchsorted(..., 'right') would, except in the
//   last span, which is closed (the reference clamps the index to nCoef).
//   z = (u - t[m0 + 1]) / h  (+ 1 when the point stays in span m0): formed from an exact small
//   difference, so z carries ~1 ulp, not the ~ns ulp of (u - lo) / h - m.
// NANFIX: a NaN parameter takes the last span, as NumPy sorts NaN to the end (derivative levels
// do not multiply by u, so only they can tell).
template <typename T, bool NANFIX>
__device__ __forceinline__ void uni_spans2(const unsigned (&kn_a)[2], const UniDesc<T> &ud, const T (&u)[2], int (&m)[2],
                                           T (&z)[2])
{
    int m0[2];
    T t1[2];
#pragma unroll
    for (int iv = 0; iv < 2; ++iv) {
        const T x = (u[iv] - ud.lo[iv]) * ud.inv_h[iv] - ud.eps;
        m0[iv] = min(max((int)x, 0), ud.ns[iv] - 1);
        t1[iv] = LdsRead<T>::template at<(int)sizeof(T)>(kn_a[iv] + (unsigned)m0[iv] * (unsigned)sizeof(T));
    }
    lds_wait_n<0, 2>(t1);
#pragma unroll
    for (int iv = 0; iv < 2; ++iv) {
        const bool inc = (u[iv] >= t1[iv]) & (m0[iv] < ud.ns[iv] - 1);
        m[iv] = m0[iv] + (inc ? 1 : 0);
        z[iv] = (u[iv] - t1[iv]) * ud.inv_h[iv] + (inc ? T(0) : T(1));
        if (NANFIX && u[iv] != u[iv]) m[iv] = ud.ns[iv] - 1;
    }
}

// Cox-de Boor recursion on a uniform span (see the file header).  wrt is wave-uniform.
template <typename T, int O, bool DERIV>
__device__ __forceinline__ void uni_basis(T z, int wrt, T inv_h, T (&b)[O])
{
#pragma unroll
    for (int k = 0; k < O; ++k) b[k] = T(0);
    b[O - 1] = T(1);
#pragma unroll
    for (int D = 1; D < O; ++D) {
        if (!DERIV || D < O - wrt) {
            if (D == 1) {
                b[O - 2] = T(1) - z;
                b[O - 1] = z;
            } else {
                const T rD = T(1.0 / (double)D);
#pragma unroll
                for (int j = 0; j < D; ++j) {
                    const int bi = O - D + j;
                    const T alpha = (z + T(D - 1 - j)) * rD;
                    b[bi - 1] += (T(1) - alpha) * b[bi];
                    b[bi] *= alpha;
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < D; ++j) {
                const int bi = O - D + j;
                b[bi - 1] -= inv_h * b[bi];
                b[bi] *= inv_h;
            }
        }
    }
    if (DERIV && wrt >= O) {
#pragma unroll
        for (int k = 0; k < O; ++k) b[k] = T(0);
    }
}

// Value basis and first-derivative basis from one recursion (they differ in the last level only).
template <typename T, int O>
__device__ __forceinline__ void uni_basis_d1(T z, T inv_h, T (&b)[O], T (&db)[O])
{
#pragma unroll
    for (int k = 0; k < O; ++k) b[k] = T(0);
    b[O - 1] = T(1);
#pragma unroll
    for (int D = 1; D < O - 1; ++D) {
        if (D == 1) {
            b[O - 2] = T(1) - z;
            b[O - 1] = z;
        } else {
            const T rD = T(1.0 / (double)D);
#pragma unroll
            for (int j = 0; j < D; ++j) {
                const int bi = O - D + j;
                const T alpha = (z + T(D - 1 - j)) * rD;
                b[bi - 1] += (T(1) - alpha) * b[bi];
                b[bi] *= alpha;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < O; ++k) db[k] = b[k];
    if constexpr (O > 1) {
        constexpr int D = O - 1;
        const T rD = T(1.0 / (double)D);
#pragma unroll
        for (int j = 0; j < D; ++j) {
            const int bi = O - D + j;
            const T alpha = (z + T(D - 1 - j)) * rD;
            b[bi - 1] += (T(1) - alpha) * b[bi];
            b[bi] *= alpha;
            db[bi - 1] -= inv_h * db[bi];
            db[bi] *= inv_h;
        }
    } else {
        db[0] = T(0);
    }
}

// Window reads of the control-point-major image: element (a, k) of dependent variable dep sits
// (k * ND + dep) elements behind the row address (compile-time offsets; dep is unrolled by the caller).
template <typename T, int O, int ND, int DEP, int A = 0, int K = 0>
__device__ __forceinline__ void uni_issue_window_d(const unsigned (&ra)[O], T (&c)[O][O])
{
    if constexpr (A < O) {
        c[A][K] = LdsRead<T>::template at<(K * ND + DEP) * (int)sizeof(T)>(ra[A]);
        if constexpr (K + 1 < O) uni_issue_window_d<T, O, ND, DEP, A, K + 1>(ra, c);
        else uni_issue_window_d<T, O, ND, DEP, A + 1, 0>(ra, c);
    }
}
template <typename T, int O, int ND>
__device__ __forceinline__ void uni_issue_window(const unsigned (&ra)[O], int dep, T (&c)[O][O])
{
    // dep comes from a fully unrolled loop: the switch folds to one case
    if constexpr (ND >= 1) { if (dep == 0) { uni_issue_window_d<T, O, ND, 0>(ra, c); return; } }
    if constexpr (ND >= 2) { if (dep == 1) { uni_issue_window_d<T, O, ND, 1>(ra, c); return; } }
    if constexpr (ND >= 3) { if (dep == 2) { uni_issue_window_d<T, O, ND, 2>(ra, c); return; } }
}

// Closed forms of the uniform basis of orders 2 .. 5 on one span (plain evaluation): the same polynomials the
// recursion produces, in 12 (cubic) / 20 (quartic) instead of 26 / 50 operations per variable.
template <typename T, int O>
__device__ __forceinline__ void uni_basis_closed(T z, T (&b)[O])
{
    static_assert(O >= 2 && O <= 5, "closed forms for orders 2 .. 5");
    if constexpr (O == 2) {
        b[0] = T(1) - z;
        b[1] = z;
    } else if constexpr (O == 3) {
        const T w = T(1) - z;
        b[0] = (w * T(0.5)) * w;
        b[2] = (z * T(0.5)) * z;
        b[1] = (T(1) - z) * z + T(0.5);                     // (-2 z^2 + 2 z + 1) / 2
    } else if constexpr (O == 4) {
        const T sixth = T(1.0 / 6.0);
        const T w = T(1) - z;
        const T z2 = z * z, w2 = w * w;
        const T z3 = z2 * z;
        b[0] = (w2 * sixth) * w;
        b[3] = z3 * sixth;
        b[1] = z3 * T(0.5) + (T(2.0 / 3.0) - z2);
        b[2] = z3 * T(-0.5) + (z2 * T(0.5) + (z * T(0.5) + sixth));
    } else {
        // quartic: 24 b = (1 - z)^4 | -4 z^4 + 12 z^3 - 6 z^2 - 12 z + 11 | 6 z^4 - 12 z^3 - 6 z^2 + 12 z + 11 |
        //                 -4 z^4 + 4 z^3 + 6 z^2 + 4 z + 1 | z^4
        const T c = T(1.0 / 24.0);
        const T w = T(1) - z;
        const T w2 = w * w, z2 = z * z;
        b[0] = (w2 * c) * w2;
        b[4] = (z2 * c) * z2;
        b[1] = (((T(-4.0 / 24.0) * z + T(12.0 / 24.0)) * z + T(-6.0 / 24.0)) * z + T(-12.0 / 24.0)) * z + T(11.0 / 24.0);
        b[2] = (((T(6.0 / 24.0) * z + T(-12.0 / 24.0)) * z + T(-6.0 / 24.0)) * z + T(12.0 / 24.0)) * z + T(11.0 / 24.0);
        b[3] = (((T(-4.0 / 24.0) * z + T(4.0 / 24.0)) * z + T(6.0 / 24.0)) * z + T(4.0 / 24.0)) * z + c;
    }
}

// row stride of the coefficient image in elements: the smallest width = 32 / O (mod 32)
__host__ __device__ inline int uni_row_stride(int nc1, int nDep, int O)
{
    int r = nc1 * (nDep <= 3 ? nDep : 1);
    while ((r & 31) != 32 / O) ++r;
    return r;
}

// Rank of the lane among the lanes of its half-wave whose windows share banks (classes equal mod
// 32 / O), requested before the basis is computed and consumed after it (one LDS atomic on a per-wave
// counter row), and the O row addresses in rotated order.
// The counters are never reset: lanes that hit one counter in one instruction get CONSECUTIVE values whatever
// it held, and only the value mod O is used.
template <int O>
__device__ __forceinline__ int uni_rank_request(int base, unsigned *s_rc, int lane)
{
    return (int)atomicAdd(&s_rc[(lane & 32) + (base & (32 / O - 1))], 1u);
}
template <typename T, int O>
__device__ __forceinline__ int uni_rows(unsigned coef_a, int base, int rank, unsigned rstride, unsigned (&ra)[O])
{
    asm volatile("" : "+v"(rank));   // first use of the atomic's result: hipcc puts its wait here
    // bank inside the shared set at step a = (class / (32 / O) + a + rho) mod O: make it (rank + a) mod O
    const int rho = (rank - base / (32 / O)) & (O - 1);
    const unsigned a0 = coef_a + (unsigned)base * (unsigned)sizeof(T);
#pragma unroll
    for (int a = 0; a < O; ++a) ra[a] = a0 + __umul24((unsigned)((a + rho) & (O - 1)), rstride);
    return rho;
}

// eval_uni: the cfg2 kernel.  out[dep * ostride + n]
template <typename T, int O, bool DERIV, int ND>
__global__ __launch_bounds__(TILE) void eval_uni(const UniDesc<T> ud, const void *__restrict__ gimg, const Params<T> prm,
                                                 const unsigned N, const long long n0, T *__restrict__ out,
                                                 const long long ostride, const Wrt wrt, unsigned long long *bad)
{
    static_assert(O == 2 || O == 4, "row rotation covers orders 2 and 4");
    constexpr int CS = ND > 0 ? ND : 1;             // column step (elements): control-point-major image when ND > 0
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned kn_a[2] = {(unsigned)(size_t)smem + ud.kn_off[0], (unsigned)(size_t)smem + ud.kn_off[1]};
    const unsigned coef_a = (unsigned)(size_t)smem + ud.coef_off;
    const int rs = ud.rs;
    const unsigned rstride = (unsigned)rs * (unsigned)sizeof(T);
    const unsigned dstride = (unsigned)(ud.ncoef[0] * rs) * (unsigned)sizeof(T);
    const int lane = threadIdx.x & 63;
    const unsigned stride = gridDim.x * (unsigned)TILE;
    unsigned n = ((threadIdx.x >> 6) * gridDim.x + blockIdx.x) * 64u + (unsigned)lane;
    const T lo0 = ud.lo[0], lo1 = ud.lo[1], hi0 = ud.hi[0], hi1 = ud.hi[1];
    const T *p0 = prm.p[0], *p1 = prm.p[1];
    // the first parameters travel while the image is staged
    T un[2] = {lo0, lo1};
    if (n < N) { un[0] = rr_load(p0, n * (unsigned)sizeof(T)); un[1] = rr_load(p1, n * (unsigned)sizeof(T)); }
    stage_linear(smem, gimg, ud.img_bytes);
    __syncthreads();
    unsigned *s_rc = reinterpret_cast<unsigned *>(smem + ud.img_bytes) + (threadIdx.x & ~63);
    asm volatile("" : "+v"(un[0]), "+v"(un[1]));   // see eval_rowrot: no load pending at the loop header
    const int nDep = ND > 0 ? ND : ud.nDep;

    for (; n < N; n += stride) {
        const T u[2] = {un[0], un[1]};
        const bool outside = (u[0] < lo0) | (u[0] > hi0) | (u[1] < lo1) | (u[1] > hi1);
        {
            const unsigned nn = min(n + stride, N - 1u) * (unsigned)sizeof(T);
            un[0] = rr_load(p0, nn);
            un[1] = rr_load(p1, nn);
        }
        if (outside) record_bad(bad, n0 + (long long)n);

        int m[2];
        T z[2];
        uni_spans2<T, DERIV>(kn_a, ud, u, m, z);
        const int base = (int)__umul24((unsigned)m[0], (unsigned)rs) + m[1] * CS;
        const int rank = uni_rank_request<O>(base, s_rc, lane);
        T b[2][O];
        if constexpr (!DERIV) {
            uni_basis_closed<T, O>(z[0], b[0]);
            uni_basis_closed<T, O>(z[1], b[1]);
        } else {
            uni_basis<T, O, true>(z[0], wrt.w[0], ud.inv_h[0], b[0]);
            uni_basis<T, O, true>(z[1], wrt.w[1], ud.inv_h[1], b[1]);
        }
        unsigned ra[O];
        const int rho = uni_rows<T, O>(coef_a, base, rank, rstride, ra);
        T b0r[O];
        rotate_basis_values<T, O>(b[0], rho, b0r);

        const unsigned off = n * (unsigned)sizeof(T);
        T *o = out;
        __builtin_amdgcn_s_setprio(3);      // see eval_rowrot
#pragma unroll
        for (int dep = 0; dep < nDep; ++dep) {
            T c[O][O];
            if constexpr (ND > 0) {
                uni_issue_window<T, O, ND>(ra, dep, c);
            } else {
#pragma unroll
                for (int a = 0; a < O; ++a) lds_issue_row<T, O>(ra[a], c[a]);
            }
            block_wait<0>(c);
            T q[O];
#pragma unroll
            for (int a = 0; a < O; ++a) {
                T t = T(0);
#pragma unroll
                for (int k = 0; k < O; ++k) t += c[a][k] * b[1][k];
                q[a] = mul_rn<T>(t, b0r[a]);
            }
            T r;
            if constexpr (O == 2) r = add_rn<T>(q[0], q[1]);
            else r = add_rn<T>(add_rn<T>(q[0], q[2]), add_rn<T>(q[1], q[3]));
            rr_store(o, off, r);
            o += ostride;
            if constexpr (ND == 0) {
#pragma unroll
                for (int a = 0; a < O; ++a) ra[a] += dstride;
            }
        }
        __builtin_amdgcn_s_setprio(0);
    }
}

// jac_uni: the fused jacobian (see jac_rowrot), or the fused normal of a surface in 3-D, with the
// uniform front end.  out[(dep * 2 + j) * ostride + n]; NORMAL: out[i * ostride + n], i < 3.
template <typename T, int O, bool NORMAL, int ND>
__global__ __launch_bounds__(TILE) void jac_uni(const UniDesc<T> ud, const void *__restrict__ gimg, const Params<T> prm,
                                                const unsigned N, const long long n0, T *__restrict__ out,
                                                const long long ostride, unsigned long long *bad, const int normalize,
                                                const int negate)
{
    static_assert(O == 2 || O == 4, "row rotation covers orders 2 and 4");
    static_assert(!NORMAL || ND == 3, "the fused normal is a surface in 3-D");
    constexpr int CS = ND > 0 ? ND : 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned kn_a[2] = {(unsigned)(size_t)smem + ud.kn_off[0], (unsigned)(size_t)smem + ud.kn_off[1]};
    const unsigned coef_a = (unsigned)(size_t)smem + ud.coef_off;
    const int rs = ud.rs;
    const unsigned rstride = (unsigned)rs * (unsigned)sizeof(T);
    const unsigned dstride = (unsigned)(ud.ncoef[0] * rs) * (unsigned)sizeof(T);
    const int lane = threadIdx.x & 63;
    const unsigned stride = gridDim.x * (unsigned)TILE;
    unsigned n = ((threadIdx.x >> 6) * gridDim.x + blockIdx.x) * 64u + (unsigned)lane;
    const T lo0 = ud.lo[0], lo1 = ud.lo[1], hi0 = ud.hi[0], hi1 = ud.hi[1];
    const T *p0 = prm.p[0], *p1 = prm.p[1];
    T un[2] = {lo0, lo1};
    if (n < N) { un[0] = rr_load(p0, n * (unsigned)sizeof(T)); un[1] = rr_load(p1, n * (unsigned)sizeof(T)); }
    stage_linear(smem, gimg, ud.img_bytes);
    __syncthreads();
    unsigned *s_rc = reinterpret_cast<unsigned *>(smem + ud.img_bytes) + (threadIdx.x & ~63);
    asm volatile("" : "+v"(un[0]), "+v"(un[1]));
    const int ndep = ND > 0 ? ND : ud.nDep;

    for (; n < N; n += stride) {
        const T u[2] = {un[0], un[1]};
        const bool outside = (u[0] < lo0) | (u[0] > hi0) | (u[1] < lo1) | (u[1] > hi1);
        {
            const unsigned nn = min(n + stride, N - 1u) * (unsigned)sizeof(T);
            un[0] = rr_load(p0, nn);
            un[1] = rr_load(p1, nn);
        }
        if (outside) record_bad(bad, n0 + (long long)n);

        int m[2];
        T z[2];
        uni_spans2<T, true>(kn_a, ud, u, m, z);
        const int base = (int)__umul24((unsigned)m[0], (unsigned)rs) + m[1] * CS;
        const int rank = uni_rank_request<O>(base, s_rc, lane);
        T b[2][O], db[2][O];
        uni_basis_d1<T, O>(z[0], ud.inv_h[0], b[0], db[0]);
        uni_basis_d1<T, O>(z[1], ud.inv_h[1], b[1], db[1]);
        unsigned ra[O];
        const int rho = uni_rows<T, O>(coef_a, base, rank, rstride, ra);
        T b0r[O], db0r[O];
        rotate_basis_values<T, O>(b[0], rho, b0r);
        rotate_basis_values<T, O>(db[0], rho, db0r);

        const unsigned off = n * (unsigned)sizeof(T);
        T *o = out;
        T su[3], sv[3];
        __builtin_amdgcn_s_setprio(3);
#pragma unroll
        for (int dep = 0; dep < ndep; ++dep) {
            T c[O][O];
            if constexpr (ND > 0) {
                uni_issue_window<T, O, ND>(ra, dep, c);
            } else {
#pragma unroll
                for (int a = 0; a < O; ++a) lds_issue_row<T, O>(ra[a], c[a]);
            }
            block_wait<0>(c);
            T qu[O], qv[O];
#pragma unroll
            for (int a = 0; a < O; ++a) {
                T t = T(0), tdv = T(0);
#pragma unroll
                for (int k = 0; k < O; ++k) { t += c[a][k] * b[1][k]; tdv += c[a][k] * db[1][k]; }
                qu[a] = mul_rn<T>(t, db0r[a]);
                qv[a] = mul_rn<T>(tdv, b0r[a]);
            }
            T du, dv;
            if constexpr (O == 2) {
                du = add_rn<T>(qu[0], qu[1]);
                dv = add_rn<T>(qv[0], qv[1]);
            } else {
                du = add_rn<T>(add_rn<T>(qu[0], qu[2]), add_rn<T>(qu[1], qu[3]));
                dv = add_rn<T>(add_rn<T>(qv[0], qv[2]), add_rn<T>(qv[1], qv[3]));
            }
            if constexpr (NORMAL) {
                if (dep == 0) { su[0] = du; sv[0] = dv; }
                else if (dep == 1) { su[1] = du; sv[1] = dv; }
                else { su[2] = du; sv[2] = dv; }
            } else {
                rr_store(o, off, du);
                rr_store(o + ostride, off, dv);
                o += 2 * ostride;
            }
            if constexpr (ND == 0) {
#pragma unroll
                for (int a = 0; a < O; ++a) ra[a] += dstride;
            }
        }
        __builtin_amdgcn_s_setprio(0);
        if constexpr (NORMAL) {
            T nx = su[1] * sv[2] - sv[1] * su[2];
            T ny = -(su[0] * sv[2] - sv[0] * su[2]);
            T nz = su[0] * sv[1] - sv[0] * su[1];
            if (negate) { nx = -nx; ny = -ny; nz = -nz; }
            if (normalize) {
                // one division: 1 / |n| times the components (a second rounding, 1 ulp, against three fp64 divisions)
                const T inv = T(1) / sqrt(nx * nx + ny * ny + nz * nz);
                nx = nx * inv; ny = ny * inv; nz = nz * inv;
            }
            rr_store(out, off, nx);
            rr_store(out + ostride, off, ny);
            rr_store(out + 2 * ostride, off, nz);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The same table-free front end for the other LDS-resident shapes: curves, surfaces of order 3 / 5,
// volumes - eval_stream / jac_stream (bsk_stream.hpp) with `uni_spans` + `uni_basis` in place of the
// bucket-table search and the knot / reciprocal rows.  Image: [domain knots of every variable]
// [unclamped coefficients in the reference's (nDep, nCoef...) layout].  fp64 only (order <= 2: also
// fp32), and only while the unclamping factors of the variables multiply to <= 5000 (host:
// upload_uniform_nd), i.e. not for three variables of order 5.
// ---------------------------------------------------------------------------------------------
template <typename T, int NIND, bool NANFIX>
__device__ __forceinline__ void uni_spans(unsigned img_a, const UniDescN<T> &un, const T (&u)[NIND], int (&m)[NIND], T (&z)[NIND])
{
    int m0[NIND];
    T t1[NIND];
#pragma unroll
    for (int iv = 0; iv < NIND; ++iv) {
        const T x = (u[iv] - un.lo[iv]) * un.inv_h[iv] - un.eps;
        m0[iv] = min(max((int)x, 0), un.ns[iv] - 1);
        t1[iv] = LdsRead<T>::template at<(int)sizeof(T)>(img_a + un.kn_off[iv] + (unsigned)m0[iv] * (unsigned)sizeof(T));
    }
    lds_wait_n<0, NIND>(t1);
#pragma unroll
    for (int iv = 0; iv < NIND; ++iv) {
        const bool inc = (u[iv] >= t1[iv]) & (m0[iv] < un.ns[iv] - 1);
        m[iv] = m0[iv] + (inc ? 1 : 0);
        z[iv] = (u[iv] - t1[iv]) * un.inv_h[iv] + (inc ? T(0) : T(1));
        if (NANFIX && u[iv] != u[iv]) m[iv] = un.ns[iv] - 1;
    }
}

template <typename T, int NIND, int O, bool DERIV>
__global__ __launch_bounds__(STREAM_BLOCK) void eval_stream_uni(const Desc<T> d, const UniDescN<T> un, const void *__restrict__ gimg,
                                                                const Params<T> prm, const long long N, T *__restrict__ out,
                                                                const long long ostride, const Wrt wrt, unsigned long long *bad)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned img_a = (unsigned)(size_t)smem;
    const unsigned coef_a = img_a + un.coef_off;
    stage_linear(smem, gimg, un.img_bytes);
    __syncthreads();
    const unsigned dstride = (unsigned)d.cstride[0] * (unsigned)sizeof(T);
    const long long stride = (long long)gridDim.x * blockDim.x;
    long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    T lo_r[NIND];
#pragma unroll
    for (int iv = 0; iv < NIND; ++iv) lo_r[iv] = un.lo[iv];
    T unx[NIND];
#pragma unroll
    for (int iv = 0; iv < NIND; ++iv) {
        unx[iv] = lo_r[iv];
        if (n < N) unx[iv] = prm.p[iv][n];
    }
#pragma unroll
    for (int iv = 0; iv < NIND; ++iv) asm volatile("" : "+v"(unx[iv]));   // see eval_stream: no load pending at the loop header

    for (; n < N; n += stride) {
        T u[NIND];
        bool outside = false;
#pragma unroll
        for (int iv = 0; iv < NIND; ++iv) {
            u[iv] = unx[iv];
            outside |= (u[iv] < lo_r[iv]) | (u[iv] > un.hi[iv]);
            unx[iv] = lo_r[iv];
        }
        if (n + stride < N) {
#pragma unroll
            for (int iv = 0; iv < NIND; ++iv) unx[iv] = prm.p[iv][n + stride];
        }
        if (outside) record_bad(bad, n);

        int m[NIND];
        T z[NIND];
        uni_spans<T, NIND, DERIV || O == 1>(img_a, un, u, m, z);   // order 1: the basis is 1 whatever u is, so the span of a NaN shows
        T b[NIND][O];
#pragma unroll
        for (int iv = 0; iv < NIND; ++iv) {
            if constexpr (!DERIV && O >= 2 && O <= 5) uni_basis_closed<T, O>(z[iv], b[iv]);
            else uni_basis<T, O, DERIV>(z[iv], DERIV ? wrt.w[iv] : 0, un.inv_h[iv], b[iv]);
        }
#pragma unroll
        for (int iv = 0; iv < NIND; ++iv) asm volatile("" : "+v"(unx[iv]));   // prefetched parameters, before the stores
        unsigned caddr = coef_a;
#pragma unroll
        for (int iv = 0; iv < NIND; ++iv) caddr += (unsigned)(m[iv] * d.cstride[iv + 1]) * (unsigned)sizeof(T);

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
                nt_store(&out[dep * ostride + n], r);
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
                nt_store(&out[dep * ostride + n], acc);
                caddr += dstride;
            }
        }
    }
}

// fused jacobian of the same shapes.  out[(dep * NIND + j) * N + n]
template <typename T, int NIND, int O>
__global__ __launch_bounds__(STREAM_BLOCK) void jac_stream_uni(const Desc<T> d, const UniDescN<T> un, const void *__restrict__ gimg,
                                                               const Params<T> prm, const long long N, T *__restrict__ out,
                                                               unsigned long long *bad)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned img_a = (unsigned)(size_t)smem;
    const unsigned coef_a = img_a + un.coef_off;
    stage_linear(smem, gimg, un.img_bytes);
    __syncthreads();
    const unsigned dstride = (unsigned)d.cstride[0] * (unsigned)sizeof(T);
    const long long stride = (long long)gridDim.x * blockDim.x;
    long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    T lo_r[NIND];
#pragma unroll
    for (int iv = 0; iv < NIND; ++iv) lo_r[iv] = un.lo[iv];
    T unx[NIND];
#pragma unroll
    for (int iv = 0; iv < NIND; ++iv) { unx[iv] = lo_r[iv]; if (n < N) unx[iv] = prm.p[iv][n]; }
#pragma unroll
    for (int iv = 0; iv < NIND; ++iv) asm volatile("" : "+v"(unx[iv]));

    for (; n < N; n += stride) {
        T u[NIND];
        bool outside = false;
#pragma unroll
        for (int iv = 0; iv < NIND; ++iv) {
            u[iv] = unx[iv];
            outside |= (u[iv] < lo_r[iv]) | (u[iv] > un.hi[iv]);
            unx[iv] = lo_r[iv];
        }
        if (n + stride < N) {
#pragma unroll
            for (int iv = 0; iv < NIND; ++iv) unx[iv] = prm.p[iv][n + stride];
        }
        if (outside) record_bad(bad, n);

        int m[NIND];
        T z[NIND];
        uni_spans<T, NIND, true>(img_a, un, u, m, z);
        T b[NIND][O], db[NIND][O];
#pragma unroll
        for (int iv = 0; iv < NIND; ++iv) uni_basis_d1<T, O>(z[iv], un.inv_h[iv], b[iv], db[iv]);
#pragma unroll
        for (int iv = 0; iv < NIND; ++iv) asm volatile("" : "+v"(unx[iv]));
        unsigned caddr = coef_a;
#pragma unroll
        for (int iv = 0; iv < NIND; ++iv) caddr += (unsigned)(m[iv] * d.cstride[iv + 1]) * (unsigned)sizeof(T);

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
                        for (int mm = 0; mm < O; ++mm) { t += c[k][mm] * b[2][mm]; tdv += c[k][mm] * db[2][mm]; }
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

}  // namespace bsk
