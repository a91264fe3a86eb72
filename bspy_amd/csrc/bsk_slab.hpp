// eval_slab2: surfaces (two variables) whose coefficient table does not fit LDS but is small enough to be STREAMED
// through it once per chunk of points - e.g. the reference's examples/TomsNasty.json shape, order (4, 5),
// nCoef (900, 11), nDep 3, fp64 = 238 KB.
//
// The cell-order pipeline moves every point's record to bin order and its result back (four kernels, ~1 GB of
// traffic per 10 M points; round 2: 888 us on the TomsNasty shape, where the runs of its 6279 bins were too short to
// combine).  For a surface the windows of the points of one span slab of the FIRST variable lie in a contiguous block
// of rows of the control-point-major table, and the whole table is only a few hundred KB.  So nothing is sorted
// globally: one persistent 1024-lane workgroup per CU takes chunks of 8192 points and
//   1. loads the chunk, finds the spans of the first variable, and orders the chunk BY PASS (pass = block of `spp`
//      consecutive spans whose rows fit the LDS slab) - wave ballots, no atomics; the order (batch position | span of
//      every point, 4 bytes) goes to a chunk-private scratch in global memory, which the same workgroup reads back
//      pass by pass; the points themselves are read again from the caller's arrays (L2 / Infinity Cache).  Inside
//      a pass the points keep their neighbours (wave, lane order), so the reads and the result stores of a wave fall
//      into a few lines;
//   2. for every pass: stages the slab (rows [g spp, g spp + spp + order0 - 1) of the table: one contiguous copy) and
//      the slice of the first variable's axis table the pass needs, then evaluates the pass's points with
//      eval_gather's arithmetic (same basis functions, same window_contract: bitwise the same results);
//   3. stores each result at its point's batch position.
// L2 -> LDS traffic: table bytes per chunk (238 KB per 8192 points = 29 B per point).  Used when the table is
// <= SLAB_MAX_TABLE bytes and at most SLAB_MAX_PASS passes cover it.
// LDS: [axis table of variable 1][bucket tables][wave counts: SLAB_WAVES x SLAB_MAX_PASS][pass starts]
//      [slice of variable 0's axis table: order0 rows x snk][slab]
#pragma once
#include "bsk_gather.hpp"
#include "bsk_binned.hpp"

namespace bsk {

constexpr int SLAB_BLOCK = 1024;            // lanes per workgroup, one per CU (two of 512 lanes with half the slab each: 409 -> 560 us on the TomsNasty shape)
constexpr int SLAB_PPT = 8;                 // points per lane and chunk (the ordering phase keeps them in registers)
constexpr int SLAB_CHUNK = SLAB_PPT * SLAB_BLOCK;
constexpr int SLAB_WAVES = SLAB_BLOCK / 64;
constexpr int SLAB_MAX_PASS = 16;
constexpr int SLAB_ROUND = 32;              // chunks of a workgroup ordered before their slabs are staged (their pass starts live in LDS)
constexpr size_t SLAB_MAX_TABLE = 2u << 20;

struct SlabPlan {
    int spp;          // spans of the first variable per pass
    int npass;
    int chunk;      // points per chunk: SLAB_CHUNK, or less for a one-slab table on a batch that would not fill the CUs
    int rows;         // rows of a slab (spp + order0 - 1)
    int snk;          // entries per row of the axis-table slice (spp + order0)
    unsigned off_lut, off_wcnt, off_pstart, off_tab0, off_slab, total;   // byte offsets in LDS
};

template <typename T>
struct alignas(2 * sizeof(T)) SlabPt { T u, v; };

// One window row (O control points x ND dependent variables, contiguous in the slab) by explicit LDS reads: all of them
// in flight together, ONE wait (left to itself hipcc reads right before every use, pairs 8-byte reads into the
// half-rate ds_read2_b64, and waits for each).  Values are tied to the wait (hipcc does not track asm operands;
// check_lds_hazards.py replays the assembly).  MIXED: the first pad1 control points of a row are READ too (not
// weighted): issuing them conditionally makes hipcc copy the in-flight registers where the paths merge - the build's
// hazard check caught exactly that; the addresses stay inside the workgroup's LDS or read as zero.
template <typename T, int O, int ND, int K = 0>
__device__ __forceinline__ void slab_row_issue(unsigned addr, T (&c)[O][ND])
{
    if constexpr (K < O) {
        lds_issue_n<T, ND, ND>(addr + (unsigned)(K * ND * (int)sizeof(T)), c[K]);
        slab_row_issue<T, O, ND, K + 1>(addr, c);
    }
}
template <typename T, int O, int ND>
__device__ __forceinline__ void slab_row_wait(T (&c)[O][ND])
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < O; ++k)
#pragma unroll
        for (int dd = 0; dd < ND; ++dd) asm volatile("" : "+v"(c[k][dd]) :: "memory");
}

// b shifted left by P places, zero filled (P = the second variable's pad: a wave-uniform run-time value, dispatched once)
template <typename T, int O, int P>
__device__ __forceinline__ void slab_shift(const T (&b)[O], T (&bs)[O])
{
#pragma unroll
    for (int k = 0; k < O; ++k) bs[k] = k + P < O ? b[k + P < O ? k + P : 0] : T(0);
}

// window_contract's arithmetic (same operations in the same order: same bits) on rows read by slab_row_issue.
// MIXED: a second variable of lower order uses the LAST order1 columns of the O-wide window.  Testing `k >= pad1` per
// term made hipcc select every accumulator update (120 of the 527 vector instructions per point of the fp64 order-5
// loop were v_cndmask); instead the row reads START at column pad1 and the weights are shifted left by pad1 once: the
// trailing columns carry weight zero and read finite values (the next control points of the slab; zero past its end).
template <typename T, int O, int ND, bool MIXED>
__device__ __forceinline__ void slab_window(unsigned w_addr, unsigned rstride, const int (&pad)[2], const T (&b)[2][O], T (&r)[ND])
{
    T b1[O];
    if constexpr (MIXED) {
        switch (pad[1]) {
            case 0: slab_shift<T, O, 0>(b[1], b1); break;
            case 1: slab_shift<T, O, 1>(b[1], b1); break;
            case 2: slab_shift<T, O, (O > 2 ? 2 : 0)>(b[1], b1); break;
            case 3: slab_shift<T, O, (O > 3 ? 3 : 0)>(b[1], b1); break;
            case 4: slab_shift<T, O, (O > 4 ? 4 : 0)>(b[1], b1); break;
            default: slab_shift<T, O, (O > 5 ? 5 : 0)>(b[1], b1); break;
        }
        w_addr += (unsigned)(pad[1] * ND * (int)sizeof(T));
    } else {
        slab_shift<T, O, 0>(b[1], b1);
    }
#pragma unroll
    for (int dd = 0; dd < ND; ++dd) r[dd] = T(0);
#pragma unroll
    for (int a = 0; a < O; ++a) {
        if (!MIXED || a >= pad[0]) {
            T c[O][ND];
            slab_row_issue<T, O, ND>(w_addr + (unsigned)a * rstride, c);
            slab_row_wait<T, O, ND>(c);
            T t[ND];
#pragma unroll
            for (int dd = 0; dd < ND; ++dd) t[dd] = T(0);
#pragma unroll
            for (int k = 0; k < O; ++k)
#pragma unroll
                for (int dd = 0; dd < ND; ++dd) t[dd] += c[k][dd] * b1[k];
#pragma unroll
            for (int dd = 0; dd < ND; ++dd) r[dd] += t[dd] * b[0][a];
        }
    }
}

// basis_bounded's recursion (same operations in the same order: same bits) on a SpanTab whose reads were issued for
// the LARGEST order: a variable of lower order uses the entries of its own levels only (the others were read from
// whatever lies there and are never touched).
// DERIV = false: plain evaluation (wrt is 0): no value / derivative branches (their merges cost copies and selects:
// 129 v_cndmask and 139 moves of the 597 vector instructions per point of the fp64 order-5 loop).
template <typename T, int OMAX, bool DERIV>
__device__ __forceinline__ void basis_regs_bounded(const SpanTab<T, OMAX> &t, int order, T u, int wrt, T (&r)[OMAX])
{
#pragma unroll
    for (int k = 0; k < OMAX; ++k) r[k] = T(0);
    if (DERIV && wrt >= order) return;
    r[OMAX - 1] = T(1);
#pragma unroll
    for (int degree = 1; degree < OMAX; ++degree) {
        if (degree < order) {
            if (!DERIV || degree < order - wrt) {
#pragma unroll
                for (int j = 0; j < degree; ++j) {
                    const int bi = OMAX - degree + j;
                    const T alpha = (u - t.kn[(OMAX - 1) - degree + j]) * t.rc[degree][j];
                    r[bi - 1] += (T(1) - alpha) * r[bi];
                    r[bi] *= alpha;
                }
            } else {
#pragma unroll
                for (int j = 0; j < degree; ++j) {
                    const int bi = OMAX - degree + j;
                    const T alpha = T(degree) * t.rc[degree][j];
                    r[bi - 1] -= alpha * r[bi];
                    r[bi] *= alpha;
                }
            }
        }
    }
}

template <typename T, int O, int ND, bool MIXED, bool DERIV>
__global__ __launch_bounds__(SLAB_BLOCK) void eval_slab2(const Desc<T> d, const TileDesc<T> td, const SlabPlan sp,
                                                   const T *__restrict__ gtab, const unsigned *__restrict__ glut,
                                                   const T *__restrict__ aos, const Params<T> prm, const long long N,
                                                   const long long base, SlabPt<T> *__restrict__ spts, unsigned *__restrict__ sidx,
                                                   T *__restrict__ out, const long long ostride,
                                                   const Wrt wrt, unsigned long long *bad, const int dbg)
{
    // dbg: timing-only switches of tools/ (BSK_SLAB_DBG; results are wrong with any of them): 1 = no evaluation,
    // 2 = no slab staging, 4 = result stores collapsed onto a few lines
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T *stab1 = reinterpret_cast<T *>(smem);                                     // axis table of variable 1 (order1 rows x nk1)
    unsigned *slut = reinterpret_cast<unsigned *>(smem + sp.off_lut);
    unsigned *wcnt = reinterpret_cast<unsigned *>(smem + sp.off_wcnt);          // [wave][pass]
    unsigned *pstart = reinterpret_cast<unsigned *>(smem + sp.off_pstart);      // [pass + 1]
    T *stab0 = reinterpret_cast<T *>(smem + sp.off_tab0);                       // slice of variable 0's axis table
    T *slab = reinterpret_cast<T *>(smem + sp.off_slab);
    const int nk0 = d.nk[0], nk1 = d.nk[1], O0 = d.order[0];
    for (int i = threadIdx.x; i < nk1 * d.order[1]; i += blockDim.x) stab1[i] = gtab[d.off[1] + i];
    for (int i = threadIdx.x; i < td.lut_len; i += blockDim.x) slut[i] = glut[i];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nc1 = d.ncoef[1];
    const int rowlen = nc1 * ND;                             // elements per table row (one control-point index of variable 0)
    int pad[2];
    pad[0] = O - d.order[0];
    pad[1] = O - d.order[1];
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const T *__restrict__ kn0 = gtab + d.off[0];              // knots of variable 0 (classification: through the vector L1)
    const int chunk = sp.chunk;                              // SLAB_CHUNK; shorter for one-slab tables on small batches (host)
    const long long nchunks = (N + chunk - 1) / chunk;
    // A ROUND = up to SLAB_ROUND of this workgroup's chunks (c, c + grid, ...): first every chunk of the round is
    // ordered (phase 1), then every slab is staged ONCE per round and the points of all the round's chunks that fall in
    // it are evaluated with no barrier between them (phase 2: the slab is static, the waves run free).  Staging a slab
    // per chunk cost 21 us and three barriers per pass and chunk (10 M points, TomsNasty shape).
    unsigned *rstart = pstart + (SLAB_MAX_PASS + 1);          // [chunk of the round][pass + 1]
    // The whole table in ONE slab (LDS-resident tables of mixed orders, which have no explicit-read kernel of their own):
    // no ordering phase and no order scratch - a lane's points are its batch positions, its results coalesced stores.
    // (MIXED instantiations only: same-order LDS-resident surfaces have their own kernels and never come here, and the
    // order-6 fp32 instantiation has no register left for the extra branch)
    const bool single = MIXED && sp.npass == 1;
    for (long long cr = blockIdx.x; cr < nchunks; cr += (long long)gridDim.x * SLAB_ROUND) {
    int nround = 0;
    for (long long c = cr; c < nchunks && nround < SLAB_ROUND; c += gridDim.x, ++nround) {
        const long long lo = c * chunk;
        const int cnt = (int)((N - lo) < chunk ? (N - lo) : chunk);
        __syncthreads();                                     // tables staged (first chunk); the previous chunk's counters / the previous round's last pass are done with LDS
        if (single) {                                        // one pass: nothing to order - phase 2 takes the chunk in batch order
            if (threadIdx.x == 0) {
                rstart[nround * (SLAB_MAX_PASS + 1)] = 0u;
                rstart[nround * (SLAB_MAX_PASS + 1) + 1] = (unsigned)cnt;
            }
        } else
        {   // ---- phase 1: order the chunk by pass
            T pu[SLAB_PPT];
#pragma unroll
            for (int k = 0; k < SLAB_PPT; ++k) {
                const int i = k * SLAB_BLOCK + (int)threadIdx.x;
                pu[k] = prm.p[0][lo + (i < cnt ? i : cnt - 1)];
            }
            int ix0[SLAB_PPT], ps[SLAB_PPT];
#pragma unroll
            for (int k = 0; k < SLAB_PPT; ++k) ix0[k] = find_span_lut<T>(kn0, slut, td, 0, d.lo[0], d.ncoef[0], pu[k]);
#pragma unroll
            for (int k = 0; k < SLAB_PPT; ++k) {
                const bool valid = k * SLAB_BLOCK + (int)threadIdx.x < cnt;
                ps[k] = valid ? (ix0[k] - O0) / sp.spp : -1;
                if (valid && ((pu[k] < d.lo[0]) | (pu[k] > d.hi[0]))) record_bad(bad, base + lo + k * SLAB_BLOCK + (long long)threadIdx.x);
            }
            // rank of every point among the wave's points of its pass (k major, then lane: neighbours stay neighbours)
            unsigned rank[SLAB_PPT];
            for (int g = 0; g < sp.npass; ++g) {
                unsigned run = 0;
#pragma unroll
                for (int k = 0; k < SLAB_PPT; ++k) {
                    const unsigned long long m = __ballot(ps[k] == g);
                    if (ps[k] == g) rank[k] = run + (unsigned)__popcll(m & lt_mask);
                    run += (unsigned)__popcll(m);
                }
                if (lane == 0) wcnt[wave * SLAB_MAX_PASS + g] = run;
            }
            __syncthreads();
            if ((int)threadIdx.x <= sp.npass) {              // pass starts: thread g sums the passes before it
                unsigned s = 0;
                for (int g = 0; g < (int)threadIdx.x; ++g)
                    for (int w = 0; w < SLAB_WAVES; ++w) s += wcnt[w * SLAB_MAX_PASS + g];
                pstart[threadIdx.x] = s;
                rstart[nround * (SLAB_MAX_PASS + 1) + (int)threadIdx.x] = s;
            }
            __syncthreads();
            // first position of (wave, pass): one thread per pair turns the wave counts into exclusive prefixes over the waves
            if ((int)threadIdx.x < sp.npass) {
                unsigned run = pstart[threadIdx.x];
                for (int w = 0; w < SLAB_WAVES; ++w) {
                    const unsigned cw = wcnt[w * SLAB_MAX_PASS + (int)threadIdx.x];
                    wcnt[w * SLAB_MAX_PASS + (int)threadIdx.x] = run;
                    run += cw;
                }
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < SLAB_PPT; ++k) {
                if (ps[k] >= 0) {
                    const unsigned p = wcnt[wave * SLAB_MAX_PASS + ps[k]] + rank[k];
                    const int i = k * SLAB_BLOCK + (int)threadIdx.x;
                    sidx[lo + p] = (unsigned)i | ((unsigned)ix0[k] << 16);
                }
            }
        }
    }
        // ---- phase 2: slab by slab
        for (int g = 0; g < sp.npass; ++g) {
            __syncthreads();                                 // scratch written (g = 0: the barrier drains this wave's stores) / previous slab's readers done
            const int r0 = g * sp.spp;
            const int r1 = min(r0 + sp.rows, d.ncoef[0]);
            {   // the slab: rows r0 .. r1 of the control-point-major table, one contiguous copy; eight loads in flight per lane
                const T *__restrict__ src = aos + (size_t)r0 * rowlen;
                const int len = (dbg & 2) ? 0 : (r1 - r0) * rowlen;
                constexpr int U = 8;
                for (int i0 = threadIdx.x; i0 < len; i0 += SLAB_BLOCK * U) {
                    T v[U];
#pragma unroll
                    for (int q = 0; q < U; ++q) v[q] = i0 + q * SLAB_BLOCK < len ? src[i0 + q * SLAB_BLOCK] : T(0);
#pragma unroll
                    for (int q = 0; q < U; ++q) if (i0 + q * SLAB_BLOCK < len) slab[i0 + q * SLAB_BLOCK] = v[q];
                }
                // axis-table slice: entries r0 .. r0 + snk of every row (knots, reciprocal rows) of variable 0
                for (int i = threadIdx.x; i < sp.snk * O0; i += SLAB_BLOCK) {
                    const int D = i / sp.snk, j = i - D * sp.snk;
                    stab0[i] = r0 + j < nk0 ? gtab[d.off[0] + D * nk0 + r0 + j] : T(0);
                }
            }
            __syncthreads();
            // axis-table addresses: entry i of row D of variable 0 sits at tab0_a + (D * snk + i) * sizeof(T)
            const unsigned tab0_a = (unsigned)(size_t)stab0 - (unsigned)(r0 * (int)sizeof(T)), tab1_a = (unsigned)(size_t)stab1;
            // (register-tight instantiations - fp64 from order 5 on - read each point when they get to it)
            constexpr bool BOTH = 2 * ((O - 1) + O * (O - 1) / 2) * ((int)sizeof(T) / 4) <= 40;
            for (int jr = 0; jr < nround; ++jr) {
            const long long lo = (cr + (long long)jr * gridDim.x) * chunk;
            const int p0 = (int)rstart[jr * (SLAB_MAX_PASS + 1) + g], p1 = (int)rstart[jr * (SLAB_MAX_PASS + 1) + g + 1];
            // The point itself is read again from the caller's arrays at its batch position: the chunk's lines were
            // loaded by this CU a moment ago (L2 / Infinity Cache), and neighbours in a pass are near neighbours in
            // the batch - a scratch copy of {u, v} in pass order cost 160 MB of writes and reads per 10 M points more.
            // Two dependent loads per point (order entry, then the point): the entry is requested two iterations ahead,
            // the point one iteration ahead.
            int p = p0 + (int)threadIdx.x;
            SlabPt<T> qn;
            unsigned en = 0, en2 = 0;
            auto fetch_pt = [&]() {
                qn.u = prm.p[0][lo + (en & 0xffffu)];
                qn.v = prm.p[1][lo + (en & 0xffffu)];
            };
            if (p < p1) { en = single ? (unsigned)p : sidx[lo + p]; if (BOTH) fetch_pt(); }
            if (p + SLAB_BLOCK < p1) en2 = single ? (unsigned)(p + SLAB_BLOCK) : sidx[lo + p + SLAB_BLOCK];
            if (dbg & 1) p = p1;
            for (; p < p1; p += SLAB_BLOCK) {
                if (!BOTH) fetch_pt();
                const SlabPt<T> q = qn;
                const unsigned e = en;
                en = en2;
                if (p + 2 * SLAB_BLOCK < p1) en2 = single ? (unsigned)(p + 2 * SLAB_BLOCK) : sidx[lo + p + 2 * SLAB_BLOCK];
                if (BOTH && p + SLAB_BLOCK < p1) fetch_pt();                // the next point streams in meanwhile
                if ((q.v < d.lo[1]) | (q.v > d.hi[1])) record_bad(bad, base + lo + (long long)(e & 0xffffu));
                int i0 = (int)(e >> 16);
                if (single) {                                // (phase 1 did not run: the span and the domain test of variable 0 here;
                    if ((q.u < d.lo[0]) | (q.u > d.hi[0])) record_bad(bad, base + lo + (long long)(e & 0xffffu));   //  row 0 of the staged axis table = all knots)
                    i0 = find_span_lut<T>(stab0, slut, td, 0, d.lo[0], d.ncoef[0], q.u);
                }
                const int i1 = find_span_lut<T>(stab1, slut, td, 1, d.lo[1], d.ncoef[1], q.v);
                T b[2][O];
                // fp64 at order 6: neither the span tables nor a window row fit the registers beside the rest - that
                // instantiation is plain C++ (compiler-managed LDS reads; it may spill, which asm reads must not:
                // check_spills.py exempts exactly it)
                constexpr bool ASM = !(sizeof(T) == 8 && O >= 6);
                if constexpr (!ASM) {
                    const T *tab0 = stab0 - r0;
                    if constexpr (MIXED) {
                        basis_bounded<T, O>(tab0, sp.snk, d.order[0], i0, q.u, wrt.w[0], b[0]);
                        basis_bounded<T, O>(stab1, nk1, d.order[1], i1, q.v, wrt.w[1], b[1]);
                    } else {
                        basis_fixed<T, O>(tab0, sp.snk, i0, q.u, wrt.w[0], b[0]);
                        basis_fixed<T, O>(stab1, nk1, i1, q.v, wrt.w[1], b[1]);
                    }
                } else
                // all table reads of a recursion in flight together, one wait (see SpanTab in bsk_binned.hpp); both
                // variables' at once while their tables fit 40 registers (fp64 from order 5 on: one after the other)
                if constexpr (BOTH) {
                    SpanTab<T, O> t0, t1;
                    span_tab_issue<T, O>(tab0_a, sp.snk, i0, t0);
                    span_tab_issue<T, O>(tab1_a, nk1, i1, t1);
                    span_tab_wait<T, O>(t0);
                    span_tab_wait<T, O>(t1);
                    if constexpr (MIXED) {
                        basis_regs_bounded<T, O, DERIV>(t0, d.order[0], q.u, wrt.w[0], b[0]);
                        basis_regs_bounded<T, O, DERIV>(t1, d.order[1], q.v, wrt.w[1], b[1]);
                    } else {
                        basis_regs<T, O, DERIV>(t0, q.u, wrt.w[0], b[0]);
                        basis_regs<T, O, DERIV>(t1, q.v, wrt.w[1], b[1]);
                    }
                } else {
                    {
                        SpanTab<T, O> t0;
                        span_tab_issue<T, O>(tab0_a, sp.snk, i0, t0);
                        span_tab_wait<T, O>(t0);
                        if constexpr (MIXED) basis_regs_bounded<T, O, DERIV>(t0, d.order[0], q.u, wrt.w[0], b[0]);
                        else basis_regs<T, O, DERIV>(t0, q.u, wrt.w[0], b[0]);
                    }
                    {
                        SpanTab<T, O> t1;
                        span_tab_issue<T, O>(tab1_a, nk1, i1, t1);
                        span_tab_wait<T, O>(t1);
                        if constexpr (MIXED) basis_regs_bounded<T, O, DERIV>(t1, d.order[1], q.v, wrt.w[1], b[1]);
                        else basis_regs<T, O, DERIV>(t1, q.v, wrt.w[1], b[1]);
                    }
                }
                // right-aligned window: rows i0 - O .. i0 - 1 (the first pad[0] of them are neither weighted nor read)
                const int wbase = (i0 - O - r0) * nc1 + (i1 - O);
                T r[ND];
                // explicit row reads while a row fits 32 registers; the widest rows (fp64, order >= 5 with four dependent
                // variables) keep the compiler's reads - their row buffer would spill, which asm reads must not
                if constexpr (ASM && O * ND * (int)sizeof(T) <= 128)
                    slab_window<T, O, ND, MIXED>((unsigned)(size_t)slab + (unsigned)(wbase * ND * (int)sizeof(T)), (unsigned)(rowlen * (int)sizeof(T)), pad, b, r);
                else
                    window_contract<T, 2, O, ND, MIXED>(slab + (long long)wbase * ND, nc1, 1, pad, b, r);
                const long long n = lo + (long long)(e & ((dbg & 4) ? 0x3fu : 0xffffu));     // (dbg 4: every store into the chunk's first lines)
#pragma unroll
                for (int dd = 0; dd < ND; ++dd)
                    nt_store(&out[dd * ostride + n], r[dd]);
            }
            }
        }
    }
}

}  // namespace bsk
