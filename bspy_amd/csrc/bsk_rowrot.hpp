// eval_rowrot: surfaces of order 2 or 4 with the table image in LDS - the cfg2 kernel.
//
// Same structure as eval_stream (one point per lane, persistent 1024-lane workgroup per CU,
// explicit LDS reads, bucket-table span search in lock step, prefetched parameters), with a
// cheaper way to take the bank conflicts out of the coefficient reads:
//
//  * the coefficient rows are staged into LDS with an ODD row stride (nCoef1 | 1 elements), so
//    the LDS bank of a window element is (i0 + i1 + const) mod 32: stepping to the next ROW of
//    the window moves one bank, exactly like stepping to the next column;
//  * lanes of a half-wave whose windows start in the same bank class would collide at every
//    read; each lane therefore walks the O rows of its window starting at row (rank mod O),
//    rank = its index among the equal-class lanes of its half-wave (one LDS atomic on a per-wave
//    counter row).  Monte Carlo of the bank model: conflict multiplicity 3.57 -> 2.41;
//  * because only the ROW order is rotated, every row is still read with compile-time column
//    offsets from one address register (4 address registers per window instead of 16, no
//    per-element address arithmetic) and summed by a fused multiply-add chain in natural column
//    order; only the O row sums are combined with a rotation-invariant tree
//    ((q0 + q2) + (q1 + q3), separately rounded), so the result does not depend on the rank:
//    runs stay bitwise reproducible and independent of a point's position in the batch.
#pragma once
#include "bsk_stream.hpp"

namespace bsk {

// Copy the table image into LDS with several global loads in flight per lane (the plain
// `s[i] = g[i]` loop compiles to load - wait - write, one L2 round trip per element: ~13 round
// trips for the 104 KB image).  Coefficient rows get the odd stride rs.
template <typename T>
__device__ __forceinline__ void stage_image_rowrot(char *smem, const Desc<T> &d, const TileDesc<T> &td,
                                                   const T *__restrict__ gtab, const unsigned *__restrict__ glut,
                                                   const T *__restrict__ gcoef, int nc1, int rs)
{
    T *stab = reinterpret_cast<T *>(smem);
    unsigned *slut = reinterpret_cast<unsigned *>(smem + td.tab_bytes);
    T *scoef = reinterpret_cast<T *>(smem + td.tab_bytes + td.lut_bytes);
    constexpr int U = 8;
    const int bd = blockDim.x;
    for (int i0 = threadIdx.x; i0 < d.coef_len; i0 += bd * U) {
        T v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) { const int i = i0 + k * bd; v[k] = i < d.coef_len ? gcoef[i] : T(0); }
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const int i = i0 + k * bd;
            if (i < d.coef_len) { const int row = i / nc1, col = i - row * nc1; scoef[row * rs + col] = v[k]; }
        }
    }
    for (int i = threadIdx.x; i < d.tab_len; i += bd) stab[i] = gtab[i];
    for (int i = threadIdx.x; i < td.lut_len; i += bd) slut[i] = glut[i];
}

template <typename T, int O, bool DERIV, int ND>
__global__ __launch_bounds__(TILE) void eval_rowrot(const Desc<T> d, const TileDesc<T> td,
                                                    const T *__restrict__ gtab, const unsigned *__restrict__ glut,
                                                    const T *__restrict__ gcoef, const Params<T> prm,
                                                    const long long N, T *__restrict__ out, const long long ostride,
                                                    const Wrt wrt, unsigned long long *bad)
{
    static_assert(O == 2 || O == 4, "row rotation covers orders 2 and 4");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned tab_a = (unsigned)(size_t)smem;
    const unsigned lut_a = tab_a + td.tab_bytes;
    const unsigned coef_a = lut_a + td.lut_bytes;
    const int nc0 = d.ncoef[0], nc1 = d.ncoef[1];
    const int rs = nc1 | 1;                                   // odd LDS row stride (elements)
    const unsigned rstride = (unsigned)rs * (unsigned)sizeof(T);
    const unsigned dstride = (unsigned)(nc0 * rs) * (unsigned)sizeof(T);
    stage_image_rowrot<T>(smem, d, td, gtab, glut, gcoef, nc1, rs);
    __syncthreads();
    // per-wave class counters of the rank rotation: [wave][half-wave][class]
    unsigned *s_rc = reinterpret_cast<unsigned *>(smem + td.tab_bytes + td.lut_bytes +
                                                 (((unsigned)d.nDep * dstride + 15u) & ~15u)) + (threadIdx.x & ~63);
    const int lane = threadIdx.x & 63;

    const int steps = td.lut_steps[0] > td.lut_steps[1] ? td.lut_steps[0] : td.lut_steps[1];
    // Wave-granular round robin: consecutive 64-point wave tiles go to different workgroups
    // (global wave = wave-in-block * gridDim + block), so the ragged last round is spread over
    // all CUs instead of keeping a few workgroups busy for one more full iteration.
    const long long stride = (long long)gridDim.x * blockDim.x;
    long long n = ((long long)(threadIdx.x >> 6) * gridDim.x + blockIdx.x) * 64 + lane;
    const T lo0 = d.lo[0], lo1 = d.lo[1], hi0 = d.hi[0], hi1 = d.hi[1];
    T un[2] = {lo0, lo1};
    if (n < N) { un[0] = prm.p[0][n]; un[1] = prm.p[1][n]; }
    // The first parameters must have landed before the loop: hipcc merges wait-count state at the
    // loop header, and a load still pending here makes it wait for the prefetch below right after
    // issuing it, in every iteration.
    asm volatile("" : "+v"(un[0]), "+v"(un[1]));
    const int nDep = ND > 0 ? ND : d.nDep;   // ND > 0: the store count per iteration is a constant,
                                             // so the prefetch is awaited with a counted vmcnt

    for (; n < N; n += stride) {
        const T u[2] = {un[0], un[1]};
        const bool outside = (u[0] < lo0) | (u[0] > hi0) | (u[1] < lo1) | (u[1] > hi1);
        un[0] = lo0; un[1] = lo1;
        if (n + stride < N) { un[0] = prm.p[0][n + stride]; un[1] = prm.p[1][n + stride]; }
        if (outside) record_bad(bad, n);

        int ix[2];
        find_spans<T, 2>(tab_a, lut_a, d, td, steps, u, ix);
        T b[2][O];
        bases_all<T, 2, O, DERIV>(tab_a, d, ix, u, wrt, b);

        const int base = (ix[0] - O) * rs + (ix[1] - O);
        s_rc[lane] = 0u;
        int rho = (int)atomicAdd(&s_rc[(lane & 32) + (base & 31)], 1u) & (O - 1);
        // rotated row order: step a reads window row (a + rho) mod O, weighted by b0 of that row
        T b0r[O];
        rotate_basis_values<T, O>(b[0], rho, b0r);
        unsigned ra[O];
#pragma unroll
        for (int a = 0; a < O; ++a)
            ra[a] = coef_a + (unsigned)(base + ((a + rho) & (O - 1)) * rs) * (unsigned)sizeof(T);

#pragma unroll
        for (int dep = 0; dep < nDep; ++dep) {
            T c[O][O];
#pragma unroll
            for (int a = 0; a < O; ++a) lds_issue_row<T, O>(ra[a], c[a]);
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
            out[dep * ostride + n] = r;
#pragma unroll
            for (int a = 0; a < O; ++a) ra[a] += dstride;
        }
    }
}

// jac_rowrot: the fused jacobian (see jac_stream) on the odd-stride image with row rotation.
// Per dependent variable the window is read once; every row gives t_a = sum_k c[a][k] b1[k] and
// td_a = sum_k c[a][k] db1[k]; du = sum_a t_a db0[a] and dv = sum_a td_a b0[a] are combined with
// the rotation-invariant tree.  out[(dep * 2 + j) * N + n]
// NORMAL = true (surfaces in 3-D, nDep == 3): the two tangents stay in registers and the kernel
// writes the normal (cross product, reference bspy/_spline_evaluation.py:215-246: optional unit
// length, optional negation) instead of the six partials: out[i * N + n], i < 3.
template <typename T, int O, bool NORMAL, int ND>
__global__ __launch_bounds__(TILE) void jac_rowrot(const Desc<T> d, const TileDesc<T> td,
                                                   const T *__restrict__ gtab, const unsigned *__restrict__ glut,
                                                   const T *__restrict__ gcoef, const Params<T> prm,
                                                   const long long N, T *__restrict__ out, unsigned long long *bad,
                                                   const int normalize, const int negate)
{
    static_assert(O == 2 || O == 4, "row rotation covers orders 2 and 4");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned tab_a = (unsigned)(size_t)smem;
    const unsigned lut_a = tab_a + td.tab_bytes;
    const unsigned coef_a = lut_a + td.lut_bytes;
    const int nc0 = d.ncoef[0], nc1 = d.ncoef[1];
    const int rs = nc1 | 1;
    const unsigned dstride = (unsigned)(nc0 * rs) * (unsigned)sizeof(T);
    stage_image_rowrot<T>(smem, d, td, gtab, glut, gcoef, nc1, rs);
    __syncthreads();
    unsigned *s_rc = reinterpret_cast<unsigned *>(smem + td.tab_bytes + td.lut_bytes +
                                                 (((unsigned)d.nDep * dstride + 15u) & ~15u)) + (threadIdx.x & ~63);
    const int lane = threadIdx.x & 63;
    const int steps = td.lut_steps[0] > td.lut_steps[1] ? td.lut_steps[0] : td.lut_steps[1];
    const long long stride = (long long)gridDim.x * blockDim.x;
    long long n = ((long long)(threadIdx.x >> 6) * gridDim.x + blockIdx.x) * 64 + lane;
    const T lo0 = d.lo[0], lo1 = d.lo[1], hi0 = d.hi[0], hi1 = d.hi[1];
    T un[2] = {lo0, lo1};
    if (n < N) { un[0] = prm.p[0][n]; un[1] = prm.p[1][n]; }
    asm volatile("" : "+v"(un[0]), "+v"(un[1]));   // see eval_rowrot: no load pending at the loop header

    for (; n < N; n += stride) {
        const T u[2] = {un[0], un[1]};
        const bool outside = (u[0] < lo0) | (u[0] > hi0) | (u[1] < lo1) | (u[1] > hi1);
        un[0] = lo0; un[1] = lo1;
        if (n + stride < N) { un[0] = prm.p[0][n + stride]; un[1] = prm.p[1][n + stride]; }
        if (outside) record_bad(bad, n);

        int ix[2];
        find_spans<T, 2>(tab_a, lut_a, d, td, steps, u, ix);
        T b[2][O], db[2][O];
        {
            T kn[2][O];
            T rc[2][O][O];
            if constexpr (O > 1) {
#pragma unroll
                for (int iv = 0; iv < 2; ++iv) {
                    const unsigned ta = tab_a + (unsigned)d.off[iv] * (unsigned)sizeof(T);
                    lds_issue_n<T, O - 1, O>(ta + (unsigned)(ix[iv] - (O - 1)) * (unsigned)sizeof(T), kn[iv]);
                    basis_issue<T, O, 1>(ta, d.nk[iv], ix[iv], rc[iv]);
                }
            }
            bases_d1_compute<T, 2, O, 0>(u, kn, rc, b, db);
        }
        const int base = (ix[0] - O) * rs + (ix[1] - O);
        s_rc[lane] = 0u;
        const int rho = (int)atomicAdd(&s_rc[(lane & 32) + (base & 31)], 1u) & (O - 1);
        T b0r[O], db0r[O];
        rotate_basis_values<T, O>(b[0], rho, b0r);
        rotate_basis_values<T, O>(db[0], rho, db0r);
        unsigned ra[O];
#pragma unroll
        for (int a = 0; a < O; ++a)
            ra[a] = coef_a + (unsigned)(base + ((a + rho) & (O - 1)) * rs) * (unsigned)sizeof(T);

        T su[3], sv[3];                    // NORMAL: the two tangent vectors
        const int ndep = NORMAL ? 3 : (ND > 0 ? ND : d.nDep);
#pragma unroll
        for (int dep = 0; dep < ndep; ++dep) {
            T c[O][O];
#pragma unroll
            for (int a = 0; a < O; ++a) lds_issue_row<T, O>(ra[a], c[a]);
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
                T *o = out + (long long)dep * 2 * N + n;
                o[0] = du;
                o[N] = dv;
            }
#pragma unroll
            for (int a = 0; a < O; ++a) ra[a] += dstride;
        }
        if constexpr (NORMAL) {
            // cofactors of the 3 x 2 tangent space: normal[i] = (-1)^i det(rows != i)
            T nx = su[1] * sv[2] - sv[1] * su[2];
            T ny = -(su[0] * sv[2] - sv[0] * su[2]);
            T nz = su[0] * sv[1] - sv[0] * su[1];
            if (negate) { nx = -nx; ny = -ny; nz = -nz; }
            if (normalize) {
                const T len = sqrt(nx * nx + ny * ny + nz * nz);
                nx = nx / len; ny = ny / len; nz = nz / len;
            }
            out[n] = nx;
            out[N + n] = ny;
            out[2 * N + n] = nz;
        }
    }
}

}  // namespace bsk
