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
//
// The loop is VALU- and latency-bound rather than conflict-bound (tools/conflict_probe.py: a
// conflict-free batch runs only 12 % faster than a random one), so the instruction count per
// point is trimmed as well:
//  * knots and reciprocal rows are staged as per-knot RECORDS {k[i], r1[i], .., r(O-1)[i]} of odd
//    stride (O | 1 elements): every table read of a variable is one address register plus a
//    compile-time offset, and the banks still spread like the flat rows;
//  * a bucket whose bracket holds at most two spans (lut_steps == 1: every well-conditioned knot
//    vector) takes one compare instead of the bisection loop;
//  * point indices are 32 bit (the launcher cuts batches into chunks of at most 2^28 points) and
//    parameters / results are addressed as scalar base + 32-bit byte offset;
//  * the prefetch of the next parameters is unconditional (clamped index), so its wait is a
//    counted vmcnt at the end of the iteration;
//  * integer multiplies are 24 bit (v_mad_u32_u24 is full rate, v_mul_lo_u32 quarter rate).
// Tried on top of this and measured no faster (kept out): two window buffers per lane (next
// variable's reads in flight during the contraction), two consecutive points per lane with
// 16-byte parameter loads / result stores, knot and reciprocal rows through the vector L1.
#pragma once
#include "bsk_stream.hpp"

namespace bsk {

constexpr unsigned RR_MAX_CHUNK = 1u << 28;   // points per launch (32-bit byte offsets: 8 B * 2^28 = 2 GiB)

// LDS image of the rowrot kernels (bytes, 16-byte aligned parts):
//   [axis records: (nk0 + nk1) x (O | 1) x T] [bucket tables: lut_len x u32] [coefficients, odd row
//   stride] [rank counters]
template <typename T, int O>
__host__ __device__ constexpr unsigned rr_rec_bytes() { return (unsigned)((O | 1) * sizeof(T)); }

template <typename T, int O>
__host__ __device__ inline unsigned rr_records_bytes(int nk0, int nk1)
{
    return ((unsigned)(nk0 + nk1) * rr_rec_bytes<T, O>() + 15u) & ~15u;
}

template <typename T>
__host__ __device__ inline unsigned rr_lut_bytes(int lut_len)
{
    return ((unsigned)lut_len * (unsigned)sizeof(unsigned) + 15u) & ~15u;
}

// Copy the table image into LDS with several global loads in flight per lane (the plain
// `s[i] = g[i]` loop compiles to load - wait - write, one L2 round trip per element: ~13 round
// trips for the 104 KB image).  Coefficient rows get the odd stride rs; the axis tables are
// transposed into per-knot records.
template <typename T, int O>
__device__ __forceinline__ void stage_image_rowrot(char *smem, const Desc<T> &d, const TileDesc<T> &td,
                                                   const T *__restrict__ gtab, const unsigned *__restrict__ glut,
                                                   const T *__restrict__ gcoef, int nc1, int rs)
{
    constexpr int REC = O | 1;
    const unsigned rec_bytes = rr_records_bytes<T, O>(d.nk[0], d.nk[1]);
    T *srec = reinterpret_cast<T *>(smem);
    unsigned *slut = reinterpret_cast<unsigned *>(smem + rec_bytes);
    T *scoef = reinterpret_cast<T *>(smem + rec_bytes + rr_lut_bytes<T>(td.lut_len));
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
#pragma unroll
    for (int iv = 0; iv < 2; ++iv) {
        const T *t = gtab + d.off[iv];
        const int nk = d.nk[iv];
        T *r = srec + (iv ? d.nk[0] * REC : 0);
        for (int i = threadIdx.x; i < nk * O; i += bd) {
            const int D = i / nk, k = i - D * nk;
            r[k * REC + D] = t[i];
        }
    }
    for (int i = threadIdx.x; i < td.lut_len; i += bd) slut[i] = glut[i];
}

// Reads of one variable's span from its records, base rb = address of record ix - (O - 1):
// the O - 1 knots, then per level D the D reciprocals r_D[ix - D .. ix - 1] (same order as
// bases_all, so bases_compute's counted waits apply unchanged).
template <typename T, int O, int J = 0>
__device__ __forceinline__ void rr_issue_knots(unsigned rb, T (&kn)[O])
{
    if constexpr (J < O - 1) {
        kn[J] = LdsRead<T>::template at<J * (int)rr_rec_bytes<T, O>()>(rb);
        rr_issue_knots<T, O, J + 1>(rb, kn);
    }
}
template <typename T, int O, int D, int J = 0>
__device__ __forceinline__ void rr_issue_level(unsigned rb, T (&rc)[O])
{
    if constexpr (J < D) {
        rc[J] = LdsRead<T>::template at<(O - 1 - D + J) * (int)rr_rec_bytes<T, O>() + D * (int)sizeof(T)>(rb);
        rr_issue_level<T, O, D, J + 1>(rb, rc);
    }
}
template <typename T, int O, int D = 1>
__device__ __forceinline__ void rr_issue_levels(unsigned rb, T (&rc)[O][O])
{
    if constexpr (D < O) {
        rr_issue_level<T, O, D>(rb, rc[D]);
        rr_issue_levels<T, O, D + 1>(rb, rc);
    }
}
template <typename T, int O>
__device__ __forceinline__ void rr_issue_tables(const unsigned (&rec_a)[2], const int (&ix)[2], T (&kn)[2][O], T (&rc)[2][O][O])
{
#pragma unroll
    for (int iv = 0; iv < 2; ++iv) {
        const unsigned rb = rec_a[iv] + __umul24((unsigned)(ix[iv] - (O - 1)), rr_rec_bytes<T, O>());
        rr_issue_knots<T, O>(rb, kn[iv]);
        rr_issue_levels<T, O>(rb, rc[iv]);
    }
}

// Span search of both variables in lock step through the bucket tables (see find_spans), knots
// read from the records.  NANFIX: a NaN parameter takes the last span like the reference's
// searchsorted (only derivative levels can tell: value levels turn every basis value into NaN).
template <typename T, int O, bool NANFIX>
__device__ __forceinline__ void rr_find_spans(const unsigned (&rec_a)[2], unsigned lut_a, const Desc<T> &d,
                                              const TileDesc<T> &td, int steps, const T (&u)[2], int (&ix)[2])
{
    constexpr unsigned RB = rr_rec_bytes<T, O>();
    unsigned e[2];
#pragma unroll
    for (int iv = 0; iv < 2; ++iv) {
        int b = (int)((u[iv] - d.lo[iv]) * td.lut_scale[iv]);
        b = min(max(b, 0), td.lut_m[iv] - 1);
        asm volatile("ds_read_b32 %0, %1" : "=v"(e[iv]) : "v"(lut_a + 4u * (unsigned)td.lut_off[iv] + 4u * (unsigned)b) : "memory");
    }
    lds_wait_n<0, 2>(e);
    int l[2], h[2];
#pragma unroll
    for (int iv = 0; iv < 2; ++iv) { l[iv] = (int)(e[iv] & 0xffffu); h[iv] = (int)(e[iv] >> 16); }
    if (steps == 1) {
        // every bracket holds at most two spans: one compare against the knot between them
        T km[2];
#pragma unroll
        for (int iv = 0; iv < 2; ++iv) km[iv] = LdsRead<T>::template at<0>(rec_a[iv] + __umul24((unsigned)l[iv], RB));
        lds_wait_n<0, 2>(km);
#pragma unroll
        for (int iv = 0; iv < 2; ++iv) l[iv] += (int)((l[iv] < h[iv]) & (km[iv] <= u[iv]));
    } else {
        for (int s = 0; s < steps; ++s) {
            T km[2];
#pragma unroll
            for (int iv = 0; iv < 2; ++iv)
                km[iv] = LdsRead<T>::template at<0>(rec_a[iv] + __umul24((unsigned)((l[iv] + h[iv]) >> 1), RB));
            lds_wait_n<0, 2>(km);
#pragma unroll
            for (int iv = 0; iv < 2; ++iv) {
                const int mid = (l[iv] + h[iv]) >> 1;
                const bool open = l[iv] < h[iv];
                const bool right = open && (km[iv] <= u[iv]);
                const bool left = open && !right;
                l[iv] = right ? mid + 1 : l[iv];
                h[iv] = left ? mid : h[iv];
            }
        }
    }
#pragma unroll
    for (int iv = 0; iv < 2; ++iv) ix[iv] = (NANFIX && u[iv] != u[iv]) ? d.ncoef[iv] : l[iv];
}

// Addresses of the O window rows in rank-rotated order and the rotation rank itself.
// (A row stride of 32 / O mod 32 with lanes ranked inside groups of 32 / O classes makes the
// rotation a pure bank shift - Monte Carlo multiplicity 2.11 against 2.37 - but measured the same
// time on MI355X and needs up to 31 pad elements per row: not kept.)
// The rank is requested (counter reset + LDS atomic) BEFORE the table reads of the recursion are
// issued and consumed after the recursion, so its round trip overlaps theirs.
__device__ __forceinline__ int rr_rank_request(int base, unsigned *s_rc, int lane)
{
    s_rc[lane] = 0u;
    return (int)atomicAdd(&s_rc[(lane & 32) + (base & 31)], 1u);
}
template <typename T, int O>
__device__ __forceinline__ int rr_rows(unsigned coef_a, int base, int rank, unsigned rstride, unsigned (&ra)[O])
{
    asm volatile("" : "+v"(rank));   // first use of the atomic's result: hipcc puts its wait here
    const int rho = rank & (O - 1);
    const unsigned a0 = coef_a + (unsigned)base * (unsigned)sizeof(T);
#pragma unroll
    for (int a = 0; a < O; ++a) ra[a] = a0 + __umul24((unsigned)((a + rho) & (O - 1)), rstride);
    return rho;
}

template <typename T>
__device__ __forceinline__ T rr_load(const T *base, unsigned byte_off)
{
    return *reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + byte_off);
}
// Results are written once and never read by the kernel: non-temporal stores (measured 6 % on
// the whole cfg2 kernel; non-temporal parameter LOADS measured slightly slower and are not used).
template <typename T>
__device__ __forceinline__ void rr_store(T *base, unsigned byte_off, T v)
{
    __builtin_nontemporal_store(v, reinterpret_cast<T *>(reinterpret_cast<char *>(base) + byte_off));
}

// N <= RR_MAX_CHUNK points of one launch; n0 = index of its first point in the caller's batch
// (for the out-of-domain record); out[dep * ostride + n].
template <typename T, int O, bool DERIV, int ND>
__global__ __launch_bounds__(TILE) void eval_rowrot(const Desc<T> d, const TileDesc<T> td,
                                                    const T *__restrict__ gtab, const unsigned *__restrict__ glut,
                                                    const T *__restrict__ gcoef, const Params<T> prm,
                                                    const unsigned N, const long long n0, T *__restrict__ out,
                                                    const long long ostride, const Wrt wrt, unsigned long long *bad)
{
    static_assert(O == 2 || O == 4, "row rotation covers orders 2 and 4");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned rec_bytes = rr_records_bytes<T, O>(d.nk[0], d.nk[1]);
    const unsigned rec_a[2] = {(unsigned)(size_t)smem, (unsigned)(size_t)smem + (unsigned)d.nk[0] * rr_rec_bytes<T, O>()};
    const unsigned lut_a = rec_a[0] + rec_bytes;
    const unsigned coef_a = lut_a + rr_lut_bytes<T>(td.lut_len);
    const int nc0 = d.ncoef[0], nc1 = d.ncoef[1];
    const int rs = nc1 | 1;                                   // odd LDS row stride (elements)
    const unsigned rstride = (unsigned)rs * (unsigned)sizeof(T);
    const unsigned dstride = (unsigned)(nc0 * rs) * (unsigned)sizeof(T);
    stage_image_rowrot<T, O>(smem, d, td, gtab, glut, gcoef, nc1, rs);
    __syncthreads();
    // per-wave class counters of the rank rotation: [wave][half-wave][class]
    unsigned *s_rc = reinterpret_cast<unsigned *>(smem + rec_bytes + rr_lut_bytes<T>(td.lut_len) +
                                                 (((unsigned)d.nDep * dstride + 15u) & ~15u)) + (threadIdx.x & ~63);
    const int lane = threadIdx.x & 63;

    const int steps = td.lut_steps[0] > td.lut_steps[1] ? td.lut_steps[0] : td.lut_steps[1];
    // Wave-granular round robin: consecutive 64-point wave tiles go to different workgroups
    // (global wave = wave-in-block * gridDim + block), so the ragged last round is spread over
    // all CUs instead of keeping a few workgroups busy for one more full iteration.
    const unsigned stride = gridDim.x * (unsigned)TILE;
    unsigned n = ((threadIdx.x >> 6) * gridDim.x + blockIdx.x) * 64u + (unsigned)lane;
    const T lo0 = d.lo[0], lo1 = d.lo[1], hi0 = d.hi[0], hi1 = d.hi[1];
    const T *p0 = prm.p[0], *p1 = prm.p[1];
    T un[2] = {lo0, lo1};
    if (n < N) { un[0] = rr_load(p0, n * (unsigned)sizeof(T)); un[1] = rr_load(p1, n * (unsigned)sizeof(T)); }
    // The first parameters must have landed before the loop: hipcc merges wait-count state at the
    // loop header, and a load still pending here makes it wait for the prefetch below right after
    // issuing it, in every iteration.
    asm volatile("" : "+v"(un[0]), "+v"(un[1]));
    const int nDep = ND > 0 ? ND : d.nDep;   // ND > 0: the store count per iteration is a constant,
                                             // so the prefetch is awaited with a counted vmcnt

    for (; n < N; n += stride) {
        const T u[2] = {un[0], un[1]};
        const bool outside = (u[0] < lo0) | (u[0] > hi0) | (u[1] < lo1) | (u[1] > hi1);
        {
            const unsigned nn = min(n + stride, N - 1u) * (unsigned)sizeof(T);
            un[0] = rr_load(p0, nn);
            un[1] = rr_load(p1, nn);
        }
        if (outside) record_bad(bad, n0 + (long long)n);

        int ix[2];
        rr_find_spans<T, O, DERIV>(rec_a, lut_a, d, td, steps, u, ix);
        const int base = (int)__umul24((unsigned)(ix[0] - O), (unsigned)rs) + (ix[1] - O);
        const int rank = rr_rank_request(base, s_rc, lane);
        T b[2][O];
        {
            T kn[2][O], rc[2][O][O];
            rr_issue_tables<T, O>(rec_a, ix, kn, rc);
            bases_compute<T, 2, O, DERIV, 0>(u, wrt, kn, rc, b);
        }

        unsigned ra[O];
        const int rho = rr_rows<T, O>(coef_a, base, rank, rstride, ra);
        // rotated row order: step a reads window row (a + rho) mod O, weighted by b0 of that row
        T b0r[O];
        rotate_basis_values<T, O>(b[0], rho, b0r);

        const unsigned off = n * (unsigned)sizeof(T);
        T *o = out;
        // the window phase is the LDS-bound part: its waves issue ahead of the ones still in the
        // (VALU-bound) recursion (measured 1 %)
        __builtin_amdgcn_s_setprio(3);
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
            rr_store(o, off, r);
            o += ostride;
#pragma unroll
            for (int a = 0; a < O; ++a) ra[a] += dstride;
        }
        __builtin_amdgcn_s_setprio(0);
    }
}

// jac_rowrot: the fused jacobian (see jac_stream) on the odd-stride image with row rotation.
// Per dependent variable the window is read once; every row gives t_a = sum_k c[a][k] b1[k] and
// td_a = sum_k c[a][k] db1[k]; du = sum_a t_a db0[a] and dv = sum_a td_a b0[a] are combined with
// the rotation-invariant tree.  out[(dep * 2 + j) * ostride + n]
// NORMAL = true (surfaces in 3-D, nDep == 3): the two tangents stay in registers and the kernel
// writes the normal (cross product, reference bspy/_spline_evaluation.py:215-246: optional unit
// length, optional negation) instead of the six partials: out[i * ostride + n], i < 3.
template <typename T, int O, bool NORMAL, int ND>
__global__ __launch_bounds__(TILE) void jac_rowrot(const Desc<T> d, const TileDesc<T> td,
                                                   const T *__restrict__ gtab, const unsigned *__restrict__ glut,
                                                   const T *__restrict__ gcoef, const Params<T> prm,
                                                   const unsigned N, const long long n0, T *__restrict__ out,
                                                   const long long ostride, unsigned long long *bad,
                                                   const int normalize, const int negate)
{
    static_assert(O == 2 || O == 4, "row rotation covers orders 2 and 4");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned rec_bytes = rr_records_bytes<T, O>(d.nk[0], d.nk[1]);
    const unsigned rec_a[2] = {(unsigned)(size_t)smem, (unsigned)(size_t)smem + (unsigned)d.nk[0] * rr_rec_bytes<T, O>()};
    const unsigned lut_a = rec_a[0] + rec_bytes;
    const unsigned coef_a = lut_a + rr_lut_bytes<T>(td.lut_len);
    const int nc0 = d.ncoef[0], nc1 = d.ncoef[1];
    const int rs = nc1 | 1;
    const unsigned rstride = (unsigned)rs * (unsigned)sizeof(T);
    const unsigned dstride = (unsigned)(nc0 * rs) * (unsigned)sizeof(T);
    stage_image_rowrot<T, O>(smem, d, td, gtab, glut, gcoef, nc1, rs);
    __syncthreads();
    unsigned *s_rc = reinterpret_cast<unsigned *>(smem + rec_bytes + rr_lut_bytes<T>(td.lut_len) +
                                                 (((unsigned)d.nDep * dstride + 15u) & ~15u)) + (threadIdx.x & ~63);
    const int lane = threadIdx.x & 63;
    const int steps = td.lut_steps[0] > td.lut_steps[1] ? td.lut_steps[0] : td.lut_steps[1];
    const unsigned stride = gridDim.x * (unsigned)TILE;
    unsigned n = ((threadIdx.x >> 6) * gridDim.x + blockIdx.x) * 64u + (unsigned)lane;
    const T lo0 = d.lo[0], lo1 = d.lo[1], hi0 = d.hi[0], hi1 = d.hi[1];
    const T *p0 = prm.p[0], *p1 = prm.p[1];
    T un[2] = {lo0, lo1};
    if (n < N) { un[0] = rr_load(p0, n * (unsigned)sizeof(T)); un[1] = rr_load(p1, n * (unsigned)sizeof(T)); }
    asm volatile("" : "+v"(un[0]), "+v"(un[1]));   // see eval_rowrot: no load pending at the loop header

    for (; n < N; n += stride) {
        const T u[2] = {un[0], un[1]};
        const bool outside = (u[0] < lo0) | (u[0] > hi0) | (u[1] < lo1) | (u[1] > hi1);
        {
            const unsigned nn = min(n + stride, N - 1u) * (unsigned)sizeof(T);
            un[0] = rr_load(p0, nn);
            un[1] = rr_load(p1, nn);
        }
        if (outside) record_bad(bad, n0 + (long long)n);

        int ix[2];
        rr_find_spans<T, O, true>(rec_a, lut_a, d, td, steps, u, ix);
        const int base = (int)__umul24((unsigned)(ix[0] - O), (unsigned)rs) + (ix[1] - O);
        const int rank = rr_rank_request(base, s_rc, lane);
        T b[2][O], db[2][O];
        {
            T kn[2][O], rc[2][O][O];
            rr_issue_tables<T, O>(rec_a, ix, kn, rc);
            bases_d1_compute<T, 2, O, 0>(u, kn, rc, b, db);
        }
        unsigned ra[O];
        const int rho = rr_rows<T, O>(coef_a, base, rank, rstride, ra);
        T b0r[O], db0r[O];
        rotate_basis_values<T, O>(b[0], rho, b0r);
        rotate_basis_values<T, O>(db[0], rho, db0r);

        const unsigned off = n * (unsigned)sizeof(T);
        T *o = out;
        T su[3], sv[3];                    // NORMAL: the two tangent vectors
        const int ndep = NORMAL ? 3 : (ND > 0 ? ND : d.nDep);
        __builtin_amdgcn_s_setprio(3);     // see eval_rowrot
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
                rr_store(o, off, du);
                rr_store(o + ostride, off, dv);
                o += 2 * ostride;
            }
#pragma unroll
            for (int a = 0; a < O; ++a) ra[a] += dstride;
        }
        __builtin_amdgcn_s_setprio(0);
        if constexpr (NORMAL) {
            // cofactors of the 3 x 2 tangent space: normal[i] = (-1)^i det(rows != i)
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

// curv_rowrot: Gaussian curvature of a surface in 3-D, fused (reference
// bspy/_spline_evaluation.py:94-107): K = (L N - M^2) / (E G - F^2) with E, F, G from the tangents
// S_u, S_v and L, M, N = S_uu . n, S_uv . n, S_vv . n, n the unit normal.  One kernel instead of
// five derivative passes + normal + epilogue (8 B per point out instead of 144 B of intermediates):
//   - one span search and one set of table reads per variable; the recursion is run for the
//     value, first- and second-derivative bases on the same registers;
//   - pass 1 over the three dependent variables reads each window once and forms S_u, S_v
//     (as jac_rowrot), then E, F, G and the unit normal;
//   - pass 2 reads the windows again, half a window (two rows) at a time to keep the register
//     count down, and accumulates L, M, N.
// Row rotation and the rotation-invariant tree as in eval_rowrot.  out[n]
template <typename T, int O>
__global__ __launch_bounds__(TILE) void curv_rowrot(const Desc<T> d, const TileDesc<T> td,
                                                    const T *__restrict__ gtab, const unsigned *__restrict__ glut,
                                                    const T *__restrict__ gcoef, const Params<T> prm,
                                                    const unsigned N, const long long n0, T *__restrict__ out,
                                                    unsigned long long *bad)
{
    static_assert(O == 2 || O == 4, "row rotation covers orders 2 and 4");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned rec_bytes = rr_records_bytes<T, O>(d.nk[0], d.nk[1]);
    const unsigned rec_a[2] = {(unsigned)(size_t)smem, (unsigned)(size_t)smem + (unsigned)d.nk[0] * rr_rec_bytes<T, O>()};
    const unsigned lut_a = rec_a[0] + rec_bytes;
    const unsigned coef_a = lut_a + rr_lut_bytes<T>(td.lut_len);
    const int nc0 = d.ncoef[0], nc1 = d.ncoef[1];
    const int rs = nc1 | 1;
    const unsigned rstride = (unsigned)rs * (unsigned)sizeof(T);
    const unsigned dstride = (unsigned)(nc0 * rs) * (unsigned)sizeof(T);
    stage_image_rowrot<T, O>(smem, d, td, gtab, glut, gcoef, nc1, rs);
    __syncthreads();
    unsigned *s_rc = reinterpret_cast<unsigned *>(smem + rec_bytes + rr_lut_bytes<T>(td.lut_len) +
                                                 (((unsigned)d.nDep * dstride + 15u) & ~15u)) + (threadIdx.x & ~63);
    const int lane = threadIdx.x & 63;
    const int steps = td.lut_steps[0] > td.lut_steps[1] ? td.lut_steps[0] : td.lut_steps[1];
    const unsigned stride = gridDim.x * (unsigned)TILE;
    unsigned n = ((threadIdx.x >> 6) * gridDim.x + blockIdx.x) * 64u + (unsigned)lane;
    const T lo0 = d.lo[0], lo1 = d.lo[1], hi0 = d.hi[0], hi1 = d.hi[1];
    const T *p0 = prm.p[0], *p1 = prm.p[1];
    Wrt w0, w1, w2;
    for (int iv = 0; iv < MAXI; ++iv) { w0.w[iv] = 0; w1.w[iv] = 1; w2.w[iv] = 2; }
    T un[2] = {lo0, lo1};
    if (n < N) { un[0] = rr_load(p0, n * (unsigned)sizeof(T)); un[1] = rr_load(p1, n * (unsigned)sizeof(T)); }
    asm volatile("" : "+v"(un[0]), "+v"(un[1]));   // see eval_rowrot: no load pending at the loop header

    for (; n < N; n += stride) {
        const T u[2] = {un[0], un[1]};
        const bool outside = (u[0] < lo0) | (u[0] > hi0) | (u[1] < lo1) | (u[1] > hi1);
        {
            const unsigned nn = min(n + stride, N - 1u) * (unsigned)sizeof(T);
            un[0] = rr_load(p0, nn);
            un[1] = rr_load(p1, nn);
        }
        if (outside) record_bad(bad, n0 + (long long)n);

        int ix[2];
        rr_find_spans<T, O, true>(rec_a, lut_a, d, td, steps, u, ix);
        const int base = (int)__umul24((unsigned)(ix[0] - O), (unsigned)rs) + (ix[1] - O);
        const int rank = rr_rank_request(base, s_rc, lane);
        T b[2][O], db[2][O], ddb[2][O];
        {
            T kn[2][O], rc[2][O][O];
            rr_issue_tables<T, O>(rec_a, ix, kn, rc);
            // the same table values give the value, first- and second-derivative bases (the waits
            // inside the second and third run find their reads complete)
            bases_compute<T, 2, O, true, 0>(u, w0, kn, rc, b);
            bases_compute<T, 2, O, true, 0>(u, w1, kn, rc, db);
            bases_compute<T, 2, O, true, 0>(u, w2, kn, rc, ddb);
        }
        unsigned ra[O];
        const int rho = rr_rows<T, O>(coef_a, base, rank, rstride, ra);
        T b0r[O], db0r[O], ddb0r[O];
        rotate_basis_values<T, O>(b[0], rho, b0r);
        rotate_basis_values<T, O>(db[0], rho, db0r);
        rotate_basis_values<T, O>(ddb[0], rho, ddb0r);

        __builtin_amdgcn_s_setprio(3);
        // ---- pass 1: tangents
        T su[3], sv[3];
#pragma unroll
        for (int dep = 0; dep < 3; ++dep) {
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
            asm volatile("" : "+v"(du), "+v"(dv));
            su[dep] = du;
            sv[dep] = dv;
#pragma unroll
            for (int a = 0; a < O; ++a) ra[a] += dstride;
        }
        // first fundamental form and unit normal (cofactors of the tangent space, as in jac_rowrot)
        T E = T(0), F = T(0), G = T(0);
#pragma unroll
        for (int dd = 0; dd < 3; ++dd) { E += su[dd] * su[dd]; F += su[dd] * sv[dd]; G += sv[dd] * sv[dd]; }
        T nrm[3];
        {
            T nx = su[1] * sv[2] - sv[1] * su[2];
            T ny = -(su[0] * sv[2] - sv[0] * su[2]);
            T nz = su[0] * sv[1] - sv[0] * su[1];
            const T len = sqrt(nx * nx + ny * ny + nz * nz);
            nrm[0] = nx / len; nrm[1] = ny / len; nrm[2] = nz / len;
        }
        asm volatile("" : "+v"(E), "+v"(F), "+v"(G), "+v"(nrm[0]), "+v"(nrm[1]), "+v"(nrm[2]));
        // ---- pass 2: second fundamental form, two window rows at a time
        T L = T(0), M = T(0), Nn = T(0);
#pragma unroll
        for (int a = 0; a < O; ++a) ra[a] -= 3u * dstride;
#pragma unroll
        for (int dep = 0; dep < 3; ++dep) {
            T part[O / 2][3];                  // per half: (q_a + q_a') of S_uu, S_uv, S_vv
#pragma unroll
            for (int h = 0; h < O / 2; ++h) {
                // rows h and h + O/2: the pairs the rotation-invariant tree adds first
                T c[2][O];
                lds_issue_row<T, O>(ra[h], c[0]);
                lds_issue_row<T, O>(ra[h + O / 2], c[1]);
                block_wait<0>(c);
                T q[2][3];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int a = h + j * (O / 2);
                    T t = T(0), tdv = T(0), tdd = T(0);
#pragma unroll
                    for (int k = 0; k < O; ++k) {
                        t += c[j][k] * b[1][k];
                        tdv += c[j][k] * db[1][k];
                        tdd += c[j][k] * ddb[1][k];
                    }
                    q[j][0] = mul_rn<T>(t, ddb0r[a]);
                    q[j][1] = mul_rn<T>(tdv, db0r[a]);
                    q[j][2] = mul_rn<T>(tdd, b0r[a]);
                }
#pragma unroll
                for (int m = 0; m < 3; ++m) part[h][m] = add_rn<T>(q[0][m], q[1][m]);
                asm volatile("" : "+v"(part[h][0]), "+v"(part[h][1]), "+v"(part[h][2]));
            }
            T suu, suv, svv;
            if constexpr (O == 2) { suu = part[0][0]; suv = part[0][1]; svv = part[0][2]; }
            else {
                suu = add_rn<T>(part[0][0], part[1][0]);
                suv = add_rn<T>(part[0][1], part[1][1]);
                svv = add_rn<T>(part[0][2], part[1][2]);
            }
            L += suu * nrm[dep];
            M += suv * nrm[dep];
            Nn += svv * nrm[dep];
#pragma unroll
            for (int a = 0; a < O; ++a) ra[a] += dstride;
        }
        __builtin_amdgcn_s_setprio(0);
        rr_store(out, n * (unsigned)sizeof(T), (L * Nn - M * M) / (E * G - F * F));
    }
}

}  // namespace bsk
