// eval_rec32: fp32 bicubic surfaces (order 4 x 4, any knots) whose table image fits LDS - the dtype of the Utah
// teapot patches (reference examples/teapot.py:349-358) and of every all-fp32 spline.
//
// eval_rowrot<float> runs the fp64 kernel's instruction stream on 4-byte data: 22 table reads and 48 coefficient
// reads per point, each a ds_read_b32 - which moves HALF the bytes per LDS cycle of a 64- or 128-bit read
// (128 against 256 B/clk/CU) - and so takes as long as the fp64 kernel for half the HBM bytes (26 % of the HBM
// roofline for evaluate).  In fp32 everything a point needs packs into 16-byte units:
//   * per-SPAN records: the 3 knots and 6 reciprocals of the span's Cox-de Boor recursion (reference
//     bspy/_spline_evaluation.py:11-26) are 9 floats = three aligned ds_read_b128 per variable (not 9 reads), and
//     the knot a bucket-table search compares against is the third float of a record;
//   * control-point-major coefficients padded to FOUR dependent variables: a control point is one aligned
//     ds_read_b128, a point's 4 x 4 window 16 reads for all dependent variables (not 16 per dependent variable).
// 26 LDS instructions per point instead of 74.  All reads are explicit (asm, counted waits: two window rows are in
// flight while the previous ones are consumed; check_lds_hazards.py replays the assembly).
// Bank conflicts: a ds_read_b128 is served in four groups of 16 lanes (MI355X_MICROARCH.md: {0-3,12-15,20-27},
// {4-11,16-19,28-31} and the same + 32), each lane taking a bank quad; 16 random windows on 16 quads collide ~2.9-fold.
// As in eval_rowrot the lanes of a group whose windows start in the same quad class walk the four window rows from
// different starting rows (rank among them from one LDS atomic on a per-wave counter; a row step moves one quad when
// the row stride is 1 mod 16 quads), and the row sums are combined by a rotation-invariant tree of separately
// rounded products: the result does not depend on the rank.
// LDS image (staged by the kernel from the handle's tables):
//   [span records of variable 0: ns0 x 12 floats][variable 1: ns1 x 12][bucket tables: lut_len x u32]
//   [coefficients [i0][i1][4], row stride rs floats]
#pragma once
#include "bsk_rowrot.hpp"
#include "bsk_binned.hpp"      // cs_f4, cs_lds_b128, SpanTab, basis_regs

namespace bsk {

constexpr int R32_O = 4;
constexpr int R32_REC = 12;                 // floats per span record (9 used)

__host__ __device__ inline int r32_row_stride(int nc1)
{
    // in bank quads (16 bytes): the smallest row length >= nc1 that is 1 mod 16 - a row step then moves ONE quad, like a
    // column step, so the quad of window element (a, k) is (class + a + k) mod 16 and the rank rotation below is a pure shift
    int q = nc1;
    while ((q & 15) != 1) ++q;
    return 4 * q;
}

__host__ __device__ inline size_t r32_lds_bytes(int ns0, int ns1, int lut_len, int nc0, int nc1)
{
    const size_t rec = (size_t)(ns0 + ns1) * R32_REC * 4;
    const size_t lut = ((size_t)lut_len * 4 + 15) & ~(size_t)15;
    return rec + lut + (size_t)nc0 * r32_row_stride(nc1) * 4 + TILE * 4;      // + rank counters: 64 per wave
}

// wait until at most CNT younger LDS reads are outstanding; the four values become available here
template <int CNT>
__device__ __forceinline__ void r32_wait4(cs_f4 &a, cs_f4 &b, cs_f4 &c, cs_f4 &d)
{
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(CNT) : "memory");
}
template <int CNT>
__device__ __forceinline__ void r32_wait3(cs_f4 &a, cs_f4 &b, cs_f4 &c)
{
    asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(a), "+v"(b), "+v"(c) : "n"(CNT) : "memory");
}

__device__ __forceinline__ void r32_tab(const cs_f4 &q0, const cs_f4 &q1, const cs_f4 &q2, SpanTab<float, 4> &t)
{
    t.kn[0] = q0[0]; t.kn[1] = q0[1]; t.kn[2] = q0[2];
    t.rc[1][0] = q0[3];
    t.rc[2][0] = q1[0]; t.rc[2][1] = q1[1];
    t.rc[3][0] = q1[2]; t.rc[3][1] = q1[3]; t.rc[3][2] = q2[0];
}

// N <= RR_MAX_CHUNK points of one launch; n0 = index of its first point in the caller's batch; out[dep * ostride + n]
template <bool DERIV, int ND>
__global__ __launch_bounds__(TILE) void eval_rec32(const Desc<float> d, const TileDesc<float> td, const float *__restrict__ gtab,
                                                   const unsigned *__restrict__ glut, const float *__restrict__ gcoef,
                                                   const Params<float> prm, const unsigned N, const long long n0,
                                                   float *__restrict__ out, const long long ostride, const Wrt wrt,
                                                   unsigned long long *bad)
{
    static_assert(ND >= 1 && ND <= 4, "control points are padded to four dependent variables");
    constexpr int O = R32_O;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nc0 = d.ncoef[0], nc1 = d.ncoef[1];
    const int ns0 = nc0 - O + 1, ns1 = nc1 - O + 1;
    const int rs = r32_row_stride(nc1);
    float *srec = reinterpret_cast<float *>(smem);
    unsigned *slut = reinterpret_cast<unsigned *>(smem + (size_t)(ns0 + ns1) * R32_REC * 4);
    float *scoef = reinterpret_cast<float *>(reinterpret_cast<char *>(slut) + (((size_t)td.lut_len * 4 + 15) & ~(size_t)15));
    // ---- stage the image
    for (int i = threadIdx.x; i < (ns0 + ns1) * R32_REC; i += blockDim.x) {
        const int s = i / R32_REC, e = i - s * R32_REC;
        const int iv = s >= ns0, sp = iv ? s - ns0 : s;
        const int ix = sp + O, nk = d.nk[iv];
        const float *t = gtab + d.off[iv];
        float v = 0.f;
        if (e < 3) v = t[ix - 3 + e];                                 // knots ix-3 .. ix-1
        else if (e == 3) v = t[1 * nk + ix - 1];                      // r1[ix-1]
        else if (e < 6) v = t[2 * nk + ix - 2 + (e - 4)];             // r2[ix-2 .. ix-1]
        else if (e < 9) v = t[3 * nk + ix - 3 + (e - 6)];             // r3[ix-3 .. ix-1]
        srec[i] = v;
    }
    for (int i = threadIdx.x; i < td.lut_len; i += blockDim.x) slut[i] = glut[i];
    for (int i = threadIdx.x; i < nc0 * nc1 * 4; i += blockDim.x) {
        const int dep = i & 3, cp = i >> 2;
        const int i0 = cp / nc1, i1 = cp - i0 * nc1;
        scoef[i0 * rs + i1 * 4 + dep] = dep < ND ? gcoef[(size_t)dep * d.cstride[0] + (size_t)i0 * d.cstride[1] + i1] : 0.f;
    }
    // rank counters of this wave: [hardware lane group of the 16-byte reads: 4][quad class: 16]; never reset (lanes that
    // hit one counter in one instruction get consecutive values whatever it held; only the value mod 4 is used)
    unsigned *s_rc = reinterpret_cast<unsigned *>(scoef + (size_t)nc0 * rs) + (threadIdx.x & ~63);
    s_rc[threadIdx.x & 63] = 0u;
    const int gid = (int)((0xF00F0FF0u >> (threadIdx.x & 31)) & 1u) + 2 * (int)((threadIdx.x >> 5) & 1);
    const unsigned rec_a[2] = {(unsigned)(size_t)srec, (unsigned)(size_t)srec + (unsigned)ns0 * (R32_REC * 4)};
    const unsigned lut_a = (unsigned)(size_t)slut, coef_a = (unsigned)(size_t)scoef;
    const int steps = td.lut_steps[0] > td.lut_steps[1] ? td.lut_steps[0] : td.lut_steps[1];
    const int lane = threadIdx.x & 63;
    const unsigned stride = gridDim.x * (unsigned)TILE;
    unsigned n = ((threadIdx.x >> 6) * gridDim.x + blockIdx.x) * 64u + (unsigned)lane;
    const float lo0 = d.lo[0], lo1 = d.lo[1], hi0 = d.hi[0], hi1 = d.hi[1];
    const float *p0 = prm.p[0], *p1 = prm.p[1];
    float un[2] = {lo0, lo1};
    if (n < N) { un[0] = rr_load(p0, n * 4u); un[1] = rr_load(p1, n * 4u); }
    __syncthreads();
    asm volatile("" : "+v"(un[0]), "+v"(un[1]));   // see eval_rowrot: no load pending at the loop header

    for (; n < N; n += stride) {
        const float u[2] = {un[0], un[1]};
        const bool outside = (u[0] < lo0) | (u[0] > hi0) | (u[1] < lo1) | (u[1] > hi1);
        {
            const unsigned nn = min(n + stride, N - 1u) * 4u;
            un[0] = rr_load(p0, nn);
            un[1] = rr_load(p1, nn);
        }
        if (outside) record_bad(bad, n0 + (long long)n);

        // ---- spans: bucket table, then the knot between the bracket's spans = third float of a record
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
        for (int s = 0; s < steps; ++s) {
            float km[2];
#pragma unroll
            for (int iv = 0; iv < 2; ++iv) {
                const int mid = (l[iv] + h[iv]) >> 1;                  // knots[mid] = left knot of span mid + 1 - O = record mid - 3, float 2
                km[iv] = LdsRead<float>::template at<8>(rec_a[iv] + __umul24((unsigned)max(mid - 3, 0), (unsigned)(R32_REC * 4)));
            }
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
        int ix[2];
#pragma unroll
        for (int iv = 0; iv < 2; ++iv) ix[iv] = (u[iv] != u[iv]) ? d.ncoef[iv] : l[iv];

        // ---- all reads of the point: 2 x 3 record reads, then the 16 control points of the window
        const unsigned ra0 = rec_a[0] + __umul24((unsigned)(ix[0] - O), (unsigned)(R32_REC * 4));
        const unsigned ra1 = rec_a[1] + __umul24((unsigned)(ix[1] - O), (unsigned)(R32_REC * 4));
        cs_f4 q0a = cs_lds_b128<0>(ra0), q0b = cs_lds_b128<16>(ra0), q0c = cs_lds_b128<32>(ra0);
        cs_f4 q1a = cs_lds_b128<0>(ra1), q1b = cs_lds_b128<16>(ra1), q1c = cs_lds_b128<32>(ra1);
        const unsigned w_a = coef_a + (__umul24((unsigned)(ix[0] - O), (unsigned)rs) + (unsigned)(ix[1] - O) * 4u) * 4u;
        const unsigned rsb = (unsigned)rs * 4u;
        // rank among the lanes of this lane's read group whose windows start in the same quad class
        const int cls = ((ix[0] - O) + (ix[1] - O)) & 15;
        int rank = (int)atomicAdd(&s_rc[gid * 16 + cls], 1u);
        // (lgkmcnt counts to 15: records + two rows = 14 reads first; a further row is requested whenever one was consumed)
        cs_f4 c[2][O];
        asm volatile("" : "+v"(rank));           // first use of the atomic's result: hipcc puts its wait here
        const int rho = rank & 3;
        auto issue_row = [&](const int a) __attribute__((always_inline)) {       // step a reads window row (a + rho) mod 4
            const unsigned row_a = w_a + __umul24((unsigned)((a + rho) & 3), rsb);
            c[a & 1][0] = cs_lds_b128<0>(row_a);
            c[a & 1][1] = cs_lds_b128<16>(row_a);
            c[a & 1][2] = cs_lds_b128<32>(row_a);
            c[a & 1][3] = cs_lds_b128<48>(row_a);
        };
        issue_row(0);
        issue_row(1);
        r32_wait3<11>(q0a, q0b, q0c);
        r32_wait3<8>(q1a, q1b, q1c);
        float b[2][O];
        {
            SpanTab<float, 4> t0, t1;
            r32_tab(q0a, q0b, q0c, t0);
            r32_tab(q1a, q1b, q1c, t1);
            basis_regs<float, 4, DERIV>(t0, u[0], wrt.w[0], b[0]);
            basis_regs<float, 4, DERIV>(t1, u[1], wrt.w[1], b[1]);
        }
        float b0r[O];
        rotate_basis_values<float, O>(b[0], rho, b0r);
        float q[O][ND];
        auto row = [&](const int a) __attribute__((always_inline)) {
            float t[ND];
#pragma unroll
            for (int dd = 0; dd < ND; ++dd) t[dd] = 0.f;
#pragma unroll
            for (int k = 0; k < O; ++k)
#pragma unroll
                for (int dd = 0; dd < ND; ++dd) t[dd] += c[a & 1][k][dd] * b[1][k];
#pragma unroll
            for (int dd = 0; dd < ND; ++dd) q[a][dd] = mul_rn<float>(t[dd], b0r[a]);
        };
        r32_wait4<4>(c[0][0], c[0][1], c[0][2], c[0][3]);
        row(0);
        issue_row(2);
        r32_wait4<4>(c[1][0], c[1][1], c[1][2], c[1][3]);
        row(1);
        issue_row(3);
        r32_wait4<4>(c[0][0], c[0][1], c[0][2], c[0][3]);
        row(2);
        r32_wait4<0>(c[1][0], c[1][1], c[1][2], c[1][3]);
        row(3);
        const unsigned off = n * 4u;
        float *o = out;
#pragma unroll
        for (int dd = 0; dd < ND; ++dd) {
            // rotation-invariant combination of the four row sums (pairs {0, 2} and {1, 3} whatever the starting row)
            rr_store(o, off, add_rn<float>(add_rn<float>(q[0][dd], q[2][dd]), add_rn<float>(q[1][dd], q[3][dd])));
            o += ostride;
        }
    }
}

}  // namespace bsk
