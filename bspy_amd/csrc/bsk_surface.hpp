// eval_surface: surfaces (two variables, common order 1..5) with the whole table image in LDS,
// written entirely in plain HIP C++ - no asm loads.
//
// hipcc pairs LDS reads whose addresses differ by a compile-time constant into ds_read2_b64,
// which moves 8-byte elements at half the rate of ds_read_b64 (MI355X: 128 vs 256 B/clk).
// Instead of hand-issuing the reads (bsk_tile.hpp / bsk_stream.hpp: asm reads with counted
// waits - fast, but the compiler may neither reorder them nor interleave two points' reads),
// every element offset here is passed through an empty asm ("laundered"): hipcc then sees
// unrelated addresses, emits one ds_read_b64 per element, and keeps full control of
// scheduling and s_waitcnt.  With P = 2 points per lane it interleaves the two points' span
// searches, recursions and contractions freely.
//
// Per point the arithmetic is exactly eval_stream's (rank rotation with the
// rotation-invariant summation for orders 2 and 4, fused multiply-add chains otherwise), so
// results are bitwise identical across the kernel families.
#pragma once
#include "bsk_stream.hpp"

namespace bsk {

__device__ __forceinline__ unsigned launder(unsigned x)
{
    asm volatile("" : "+v"(x));
    return x;
}

template <typename T>
__device__ __forceinline__ T lds_at(const char *lds, unsigned byte_off)
{
    return *reinterpret_cast<const T *>(lds + byte_off);
}

// Cox-de Boor recursion of one variable from the LDS axis table (plain loads, laundered
// offsets).  tab_off: byte offset of the variable's table in the image.
template <typename T, int O, bool DERIV>
__device__ __forceinline__ void basis_plain(const char *lds, unsigned tab_off, int nk, int ix, T u, int wrt, T (&b)[O])
{
#pragma unroll
    for (int k = 0; k < O; ++k) b[k] = T(0);
    b[O - 1] = T(1);
    if constexpr (O > 1) {
        T kn[O - 1];
#pragma unroll
        for (int j = 0; j < O - 1; ++j)
            kn[j] = lds_at<T>(lds, launder(tab_off + (unsigned)(ix - (O - 1) + j) * (unsigned)sizeof(T)));
#pragma unroll
        for (int D = 1; D < O; ++D) {
            T rc[O - 1];
#pragma unroll
            for (int j = 0; j < D; ++j)
                rc[j] = lds_at<T>(lds, launder(tab_off + (unsigned)(D * nk + ix - D + j) * (unsigned)sizeof(T)));
            if (!DERIV || D < O - wrt) {
#pragma unroll
                for (int j = 0; j < D; ++j) {
                    const int bi = O - D + j;
                    const T alpha = (u - kn[(O - 1) - D + j]) * rc[j];
                    b[bi - 1] += (T(1) - alpha) * b[bi];
                    b[bi] *= alpha;
                }
            } else {
#pragma unroll
                for (int j = 0; j < D; ++j) {
                    const int bi = O - D + j;
                    const T alpha = T(D) * rc[j];
                    b[bi - 1] -= alpha * b[bi];
                    b[bi] *= alpha;
                }
            }
        }
    }
    if (DERIV && wrt >= O) {
#pragma unroll
        for (int k = 0; k < O; ++k) b[k] = T(0);
    }
}

constexpr int SURF_BLOCK = 1024;

template <typename T, int O, bool DERIV, int P>
__global__ __launch_bounds__(P == 1 ? 1024 : 512) void eval_surface(const Desc<T> d, const TileDesc<T> td,
                                                                  const T *__restrict__ gtab,
                                                                  const unsigned *__restrict__ glut,
                                                                  const T *__restrict__ gcoef, const Params<T> prm,
                                                                  const long long N, T *__restrict__ out,
                                                                  const long long ostride, const Wrt wrt,
                                                                  unsigned long long *bad)
{
    static_assert(P == 1 || P == 2, "points per lane");
    typedef typename Vec2<T>::type V2;
    constexpr bool ROT = (O == 2 || O == 4);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const char *lds = smem;
    const unsigned lut_o = td.tab_bytes;
    const unsigned coef_o = td.tab_bytes + td.lut_bytes;
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
    const T lo0 = d.lo[0], lo1 = d.lo[1], hi0 = d.hi[0], hi1 = d.hi[1];
    const long long nitems = (N + P - 1) / P;                     // one item = P adjacent points
    const long long stride = (long long)gridDim.x * blockDim.x;
    long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;

    auto fetch = [&](long long item, T (&dst)[P][2]) {
#pragma unroll
        for (int p = 0; p < P; ++p) { dst[p][0] = lo0; dst[p][1] = lo1; }
        if constexpr (P == 2) {
            if (2 * item + 1 < N) {
                const V2 a = *reinterpret_cast<const V2 *>(prm.p[0] + 2 * item);
                const V2 c = *reinterpret_cast<const V2 *>(prm.p[1] + 2 * item);
                dst[0][0] = a.x; dst[1][0] = a.y; dst[0][1] = c.x; dst[1][1] = c.y;
            } else if (2 * item < N) {
                dst[0][0] = prm.p[0][2 * item];
                dst[0][1] = prm.p[1][2 * item];
            }
        } else {
            if (item < N) { dst[0][0] = prm.p[0][item]; dst[0][1] = prm.p[1][item]; }
        }
    };
    T un[P][2];
    fetch(q, un);

    for (; q < nitems; q += stride) {
        T u[P][2];
#pragma unroll
        for (int p = 0; p < P; ++p) { u[p][0] = un[p][0]; u[p][1] = un[p][1]; }
        fetch(q + stride, un);

        bool valid[P];
#pragma unroll
        for (int p = 0; p < P; ++p) {
            valid[p] = P * q + p < N;
            const bool outside = (u[p][0] < lo0) | (u[p][0] > hi0) | (u[p][1] < lo1) | (u[p][1] > hi1);
            if (valid[p] && outside) record_bad(bad, P * q + p);
        }

        // span search (bucket table, then `steps` binary steps), all points and variables in lock step
        int l[P][2], h[P][2];
#pragma unroll
        for (int p = 0; p < P; ++p)
#pragma unroll
            for (int iv = 0; iv < 2; ++iv) {
                int bk = (int)((u[p][iv] - d.lo[iv]) * td.lut_scale[iv]);
                bk = bk < 0 ? 0 : (bk > td.lut_m[iv] - 1 ? td.lut_m[iv] - 1 : bk);
                const unsigned e = lds_at<unsigned>(lds, lut_o + 4u * (unsigned)(td.lut_off[iv] + bk));
                l[p][iv] = (int)(e & 0xffffu);
                h[p][iv] = (int)(e >> 16);
            }
        for (int s = 0; s < steps; ++s) {
#pragma unroll
            for (int p = 0; p < P; ++p)
#pragma unroll
                for (int iv = 0; iv < 2; ++iv) {
                    const int mid = (l[p][iv] + h[p][iv]) >> 1;
                    const T km = lds_at<T>(lds, (unsigned)(d.off[iv] + mid) * (unsigned)sizeof(T));
                    const bool open = l[p][iv] < h[p][iv];
                    const bool right = open && (km <= u[p][iv]);
                    const bool left = open && !right;
                    l[p][iv] = right ? mid + 1 : l[p][iv];
                    h[p][iv] = left ? mid : h[p][iv];
                }
        }

        T b[P][2][O];
        unsigned cw[P];
#pragma unroll
        for (int p = 0; p < P; ++p) {
            int ix[2];
#pragma unroll
            for (int iv = 0; iv < 2; ++iv) {
                ix[iv] = (u[p][iv] != u[p][iv]) ? d.ncoef[iv] : l[p][iv];
                basis_plain<T, O, DERIV>(lds, (unsigned)d.off[iv] * (unsigned)sizeof(T), d.nk[iv], ix[iv], u[p][iv],
                                         DERIV ? wrt.w[iv] : 0, b[p][iv]);
            }
            cw[p] = coef_o + (unsigned)((ix[0] - O) * d.cstride[1] + (ix[1] - O)) * (unsigned)sizeof(T);
        }

        T br[P][O];
        unsigned co[P][O];
#pragma unroll
        for (int p = 0; p < P; ++p) {
            if constexpr (ROT) {
                const int cls = (int)((cw[p] - coef_o) / (unsigned)sizeof(T)) & 31;
                s_rc[lane] = 0u;
                const int rho = (int)atomicAdd(&s_rc[(lane & 32) + cls], 1u) & (O - 1);
                rotate_basis<T, O>(b[p][1], rho, br[p], co[p]);
            } else {
#pragma unroll
                for (int j = 0; j < O; ++j) { br[p][j] = b[p][1][j]; co[p][j] = launder((unsigned)j * (unsigned)sizeof(T)); }
            }
        }

        auto one_dep = [&](int dep) {
            T res[P];
#pragma unroll
            for (int p = 0; p < P; ++p) {
                T acc = T(0);
#pragma unroll
                for (int a = 0; a < O; ++a) {
                    const unsigned row = cw[p] + (unsigned)a * rstride;
                    T c[O];
#pragma unroll
                    for (int j = 0; j < O; ++j) c[j] = lds_at<T>(lds, row + co[p][j]);
                    if constexpr (ROT) {
#pragma clang fp contract(off)
                        T t;
                        if constexpr (O == 2) {
                            t = add_rn<T>(mul_rn<T>(c[0], br[p][0]), mul_rn<T>(c[1], br[p][1]));
                        } else {
                            t = add_rn<T>(add_rn<T>(mul_rn<T>(c[0], br[p][0]), mul_rn<T>(c[2], br[p][2])),
                                          add_rn<T>(mul_rn<T>(c[1], br[p][1]), mul_rn<T>(c[3], br[p][3])));
                        }
                        acc = add_rn<T>(acc, mul_rn<T>(t, b[p][0][a]));
                    } else {
                        T t = T(0);
#pragma unroll
                        for (int j = 0; j < O; ++j) t += c[j] * br[p][j];
                        acc += t * b[p][0][a];
                    }
                }
                res[p] = acc;
                cw[p] += dstride;
            }
            if constexpr (P == 2) {
                T *o = out + dep * ostride + 2 * q;
                if (valid[1]) { V2 v; v.x = res[0]; v.y = res[1]; *reinterpret_cast<V2 *>(o) = v; }
                else if (valid[0]) o[0] = res[0];
            } else {
                if (valid[0]) out[dep * ostride + q] = res[0];
            }
        };
        if (d.nDep == 3) {
            one_dep(0); one_dep(1); one_dep(2);
        } else {
            for (int dep = 0; dep < d.nDep; ++dep) one_dep(dep);
        }
    }
}

}  // namespace bsk
