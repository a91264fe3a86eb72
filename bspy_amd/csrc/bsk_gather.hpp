// eval_gather: coefficient tables too large for LDS (BASELINE cfg5: 40^3 x 4 fp32 = 1 MB) are
// gathered from L2.  The window of a point is prod(order) control points; with the reference's
// (nDep, nCoef...) layout every control point costs nDep separate 4/8-byte loads from nDep
// planes, i.e. nDep cache sectors.  Here the table is kept a second time in control-point-major
// order (..., nCoef_last, nDep): the nDep values of a control point are ONE vector load
// (16 bytes for nDep = 4 fp32) and the `order` control points of a window row are contiguous
// (80 bytes for order 5), so a row costs 1-2 sectors instead of 4-8.  Axis tables stay in LDS.
#pragma once
#include "bsk_kernels.hpp"

namespace bsk {

template <typename T, int ND> struct CP { T v[ND]; };   // one control point, naturally aligned below

template <typename T, int ND>
__device__ __forceinline__ void load_cp(const T *__restrict__ p, T (&v)[ND])
{
    if constexpr (ND == 4 && sizeof(T) == 4) {
        const float4 q = *reinterpret_cast<const float4 *>(p);
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    } else if constexpr (ND == 2 && sizeof(T) == 4) {
        const float2 q = *reinterpret_cast<const float2 *>(p);
        v[0] = q.x; v[1] = q.y;
    } else if constexpr (ND == 2 && sizeof(T) == 8) {
        const double2 q = *reinterpret_cast<const double2 *>(p);
        v[0] = q.x; v[1] = q.y;
    } else if constexpr (ND == 4 && sizeof(T) == 8) {
        const double2 q0 = *reinterpret_cast<const double2 *>(p), q1 = *reinterpret_cast<const double2 *>(p + 2);
        v[0] = q0.x; v[1] = q0.y; v[2] = q1.x; v[3] = q1.y;
    } else {
#pragma unroll
        for (int dd = 0; dd < ND; ++dd) v[dd] = p[dd];
    }
}

// aos: control-point-major coefficients, element (i0, .., i_last, dep) at
// ((i0 * nc1 + i1) * nc2 + i2) * ND + dep.   out[dep * ostride + n]
template <typename T, int NIND, int O, int ND, bool MIXED>
__global__ __launch_bounds__(256) void eval_gather(const Desc<T> d, const T *__restrict__ gtab,
                                                   const T *__restrict__ aos, const Params<T> prm,
                                                   const long long N, T *__restrict__ out, const long long ostride,
                                                   const Wrt wrt, unsigned long long *bad)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T *stab = reinterpret_cast<T *>(smem);
    for (int i = threadIdx.x; i < d.tab_len; i += blockDim.x) stab[i] = gtab[i];
    __syncthreads();

    // strides of the control-point-major table (in control points)
    int cs[NIND];
    cs[NIND - 1] = 1;
#pragma unroll
    for (int iv = NIND - 2; iv >= 0; --iv) cs[iv] = cs[iv + 1] * d.ncoef[iv + 1];
    // O is the LARGEST order of the spline; a variable of lower order has its basis right aligned
    // in the O slots (basis_bounded) and its first O - order window entries are neither weighted
    // nor loaded (wave-uniform tests: the orders are kernel arguments).  MIXED = false (one common
    // order) compiles all of that out.
    int pad[NIND];
#pragma unroll
    for (int iv = 0; iv < NIND; ++iv) pad[iv] = O - d.order[iv];

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
            const int ix = find_span<T>(tab, d.order[iv], d.ncoef[iv], d.steps[iv], u);
            if constexpr (MIXED) basis_bounded<T, O>(tab, d.nk[iv], d.order[iv], ix, u, wrt.w[iv], b[iv]);
            else basis_fixed<T, O>(tab, d.nk[iv], ix, u, wrt.w[iv], b[iv]);
            base += (ix - O) * cs[iv];                        // right-aligned window, see the kernel comment
        }
        if (outside) record_bad(bad, n);
        const T *__restrict__ w0 = aos + (long long)base * ND;
        T r[ND];
#pragma unroll
        for (int dd = 0; dd < ND; ++dd) r[dd] = T(0);
        if constexpr (NIND == 1) {
#pragma unroll
            for (int k = 0; k < O; ++k) {
                if (!MIXED || k >= pad[0]) {
                    T c[ND];
                    load_cp<T, ND>(w0 + k * ND, c);
#pragma unroll
                    for (int dd = 0; dd < ND; ++dd) r[dd] += c[dd] * b[0][k];
                }
            }
        } else if constexpr (NIND == 2) {
#pragma unroll
            for (int a = 0; a < O; ++a) {
                if (!MIXED || a >= pad[0]) {
                    T t[ND];
#pragma unroll
                    for (int dd = 0; dd < ND; ++dd) t[dd] = T(0);
#pragma unroll
                    for (int k = 0; k < O; ++k) {
                        if (!MIXED || k >= pad[1]) {
                            T c[ND];
                            load_cp<T, ND>(w0 + ((long long)a * cs[0] + k) * ND, c);
#pragma unroll
                            for (int dd = 0; dd < ND; ++dd) t[dd] += c[dd] * b[1][k];
                        }
                    }
#pragma unroll
                    for (int dd = 0; dd < ND; ++dd) r[dd] += t[dd] * b[0][a];
                }
            }
        } else {
#pragma unroll
            for (int a = 0; a < O; ++a) {
                if (!MIXED || a >= pad[0]) {
                    T s[ND];
#pragma unroll
                    for (int dd = 0; dd < ND; ++dd) s[dd] = T(0);
#pragma unroll
                    for (int k = 0; k < O; ++k) {
                        if (!MIXED || k >= pad[1]) {
                            T t[ND];
#pragma unroll
                            for (int dd = 0; dd < ND; ++dd) t[dd] = T(0);
                            const T *__restrict__ row = w0 + ((long long)a * cs[0] + (long long)k * cs[1]) * ND;
#pragma unroll
                            for (int m = 0; m < O; ++m) {
                                if (!MIXED || m >= pad[2]) {
                                    T c[ND];
                                    load_cp<T, ND>(row + m * ND, c);
#pragma unroll
                                    for (int dd = 0; dd < ND; ++dd) t[dd] += c[dd] * b[2][m];
                                }
                            }
#pragma unroll
                            for (int dd = 0; dd < ND; ++dd) s[dd] += t[dd] * b[1][k];
                        }
                    }
#pragma unroll
                    for (int dd = 0; dd < ND; ++dd) r[dd] += s[dd] * b[0][a];
                }
            }
        }
#pragma unroll
        for (int dd = 0; dd < ND; ++dd) nt_store(&out[dd * ostride + n], r[dd]);
    }
}

}  // namespace bsk
