// eval_binned: large batches on coefficient tables that live in L2 (BASELINE cfg5: 40^3 x 4 fp32 =
// 1 MB) are evaluated in CELL ORDER.
//
// eval_gather is bound by the vector L1's tag rate: the lanes of a wave hold unrelated points, so
// every load instruction of the window walk touches 64 different cache lines (one tag lookup per
// line per clock).  Points that share the spans of the first two variables read the same
// order x order bundle of coefficient rows, and those rows are contiguous along the last variable
// in the control-point-major table - a wave of such points touches a handful of lines per load.
// The batch is therefore counting-sorted by the cell (span0, span1) before it is evaluated:
//
//   bin_count    one workgroup per contiguous chunk of the batch: spans of variables 0 and 1,
//                domain test, LDS histogram over the cells; writes cell[n] and the chunk's
//                histogram column M[cell][chunk]
//   bin_rowscan  one workgroup per cell: exclusive scan over the chunks, cell total
//   bin_topscan  one workgroup: exclusive scan of the cell totals
//   bin_scatter  same chunks: slot = start[cell] + M[cell][chunk] + rank inside the chunk (LDS
//                atomic); writes the record {u..} to rec[slot] and slot[n]
//   eval_binned_lds  eval_gather's arithmetic on the records in slot order with the cell's
//                coefficient rows staged in LDS; control point results to tmp[slot] (one
//                16/32-byte store per point)
//   bin_unpermute out[dep][n] = tmp[slot[n]][dep]: gathered 16/32-byte reads, coalesced SoA stores
//
// No global atomics (64 lanes adding to 64 different addresses run at 0.08 TB/s on MI355X); a
// point's result does not depend on its slot, so the non-deterministic order inside a cell (LDS
// atomics of different waves) cannot change the output.
#pragma once
#include "bsk_gather.hpp"

namespace bsk {

constexpr int BIN_BLOCK = 512;          // threads per workgroup of the binning kernels
constexpr int BIN_MAX_CELLS = 8192;     // LDS histogram capacity (32 KB of counters)
constexpr int BIN_MAX_CHUNKS = 1024;    // one row scan handles this many chunks

// Cells of the first two variables, coarsened by shifts until they fit BIN_MAX_CELLS.
struct BinPlan {
    int sh0, sh1;       // span index >> shift
    int n1;             // cells along variable 1
    int cells;          // total
    int chunks;         // workgroups of bin_count / bin_scatter
    long long chunk;    // points per chunk
};

// record of one point in slot order: its parameters, padded to 16-byte multiples (one vector access)
template <typename T, int NIND>
struct __attribute__((aligned(16))) BinRec {
    static constexpr int WORDS = (NIND * (int)sizeof(T) + 15) / 16 * 16 / (int)sizeof(T);
    T v[WORDS];
};
template <typename T, int ND>
struct __attribute__((aligned(16))) BinOut {
    static constexpr int WORDS = (ND * (int)sizeof(T) + 15) / 16 * 16 / (int)sizeof(T);
    T v[WORDS];
};

template <typename T, int O>
__device__ __forceinline__ int bin_cell(const T *stab, const Desc<T> &d, const BinPlan &bp, T u0, T u1)
{
    const int i0 = find_span<T>(stab + d.off[0], d.order[0], d.ncoef[0], d.steps[0], u0) - d.order[0];
    const int i1 = find_span<T>(stab + d.off[1], d.order[1], d.ncoef[1], d.steps[1], u1) - d.order[1];
    return (i0 >> bp.sh0) * bp.n1 + (i1 >> bp.sh1);
}

// LDS: [axis tables][cells x u32]
template <typename T, int NIND, int O>
__global__ __launch_bounds__(BIN_BLOCK) void bin_count(const Desc<T> d, const BinPlan bp, const T *__restrict__ gtab,
                                                       const Params<T> prm, const long long N,
                                                       unsigned short *__restrict__ cell, unsigned *__restrict__ M,
                                                       unsigned long long *bad)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T *stab = reinterpret_cast<T *>(smem);
    unsigned *hist = reinterpret_cast<unsigned *>(smem + ((sizeof(T) * (size_t)d.tab_len + 15) & ~(size_t)15));
    for (int i = threadIdx.x; i < d.tab_len; i += blockDim.x) stab[i] = gtab[i];
    for (int i = threadIdx.x; i < bp.cells; i += blockDim.x) hist[i] = 0u;
    __syncthreads();
    const long long lo = (long long)blockIdx.x * bp.chunk, hi = lo + bp.chunk < N ? lo + bp.chunk : N;
    for (long long n = lo + threadIdx.x; n < hi; n += blockDim.x) {
        bool outside = false;
        T u[NIND];
#pragma unroll
        for (int iv = 0; iv < NIND; ++iv) {
            u[iv] = prm.p[iv][n];
            outside |= (u[iv] < d.lo[iv]) | (u[iv] > d.hi[iv]);
        }
        if (outside) record_bad(bad, n);
        const int c = bin_cell<T, O>(stab, d, bp, u[0], u[1]);
        cell[n] = (unsigned short)c;
        atomicAdd(&hist[c], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < bp.cells; i += blockDim.x) M[(size_t)i * bp.chunks + blockIdx.x] = hist[i];
}

// one workgroup per cell: M[cell][0..chunks) -> exclusive prefix, total[cell]
__global__ __launch_bounds__(BIN_MAX_CHUNKS) void bin_rowscan(const int chunks, unsigned *__restrict__ M,
                                                              unsigned *__restrict__ total)
{
    __shared__ unsigned s[BIN_MAX_CHUNKS];
    unsigned *row = M + (size_t)blockIdx.x * chunks;
    const int t = threadIdx.x;
    const unsigned v = t < chunks ? row[t] : 0u;
    s[t] = v;
    __syncthreads();
    for (int off = 1; off < BIN_MAX_CHUNKS; off <<= 1) {
        const unsigned add = t >= off ? s[t - off] : 0u;
        __syncthreads();
        s[t] += add;
        __syncthreads();
    }
    if (t < chunks) row[t] = s[t] - v;
    if (t == BIN_MAX_CHUNKS - 1) total[blockIdx.x] = s[t];
}

// one workgroup: total[0..cells) -> exclusive prefix in start[]
__global__ __launch_bounds__(1024) void bin_topscan(const int cells, const unsigned *__restrict__ total,
                                                    unsigned *__restrict__ start)
{
    __shared__ unsigned s[1024];
    const int t = threadIdx.x;
    const int per = (cells + 1023) / 1024;
    unsigned sum = 0;
    for (int i = 0; i < per; ++i) { const int k = t * per + i; if (k < cells) sum += total[k]; }
    s[t] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const unsigned add = t >= off ? s[t - off] : 0u;
        __syncthreads();
        s[t] += add;
        __syncthreads();
    }
    unsigned run = s[t] - sum;
    for (int i = 0; i < per; ++i) {
        const int k = t * per + i;
        if (k < cells) { start[k] = run; run += total[k]; }
    }
}

template <typename T, int NIND>
__global__ __launch_bounds__(BIN_BLOCK) void bin_scatter(const BinPlan bp, const Params<T> prm, const long long N,
                                                         const unsigned short *__restrict__ cell,
                                                         const unsigned *__restrict__ M, const unsigned *__restrict__ start,
                                                         BinRec<T, NIND> *__restrict__ rec, unsigned *__restrict__ slot)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned *next = reinterpret_cast<unsigned *>(smem);
    for (int i = threadIdx.x; i < bp.cells; i += blockDim.x) next[i] = start[i] + M[(size_t)i * bp.chunks + blockIdx.x];
    __syncthreads();
    const long long lo = (long long)blockIdx.x * bp.chunk, hi = lo + bp.chunk < N ? lo + bp.chunk : N;
    for (long long n = lo + threadIdx.x; n < hi; n += blockDim.x) {
        const unsigned p = atomicAdd(&next[cell[n]], 1u);
        BinRec<T, NIND> r;
#pragma unroll
        for (int k = 0; k < BinRec<T, NIND>::WORDS; ++k) r.v[k] = T(0);
#pragma unroll
        for (int iv = 0; iv < NIND; ++iv) r.v[iv] = prm.p[iv][n];
        rec[p] = r;
        slot[n] = p;
    }
}

// Window contraction on a control-point-major table with strides s0 / s1 (control points) of the
// first / second variable; the last variable is contiguous.  Same operation order as eval_gather.
// O is the spline's LARGEST order: a variable of lower order has its basis right aligned
// (basis_bounded) and its first pad = O - order window entries are neither weighted nor loaded
// (MIXED; splines of one common order compile the tests out).
template <typename T, int NIND, int O, int ND, bool MIXED, typename CP>
__device__ __forceinline__ void window_contract(CP w0, int s0, int s1, const int (&pad)[NIND], const T (&b)[NIND][O],
                                                T (&r)[ND])
{
#pragma unroll
    for (int dd = 0; dd < ND; ++dd) r[dd] = T(0);
    if constexpr (NIND == 2) {
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
                        load_cp<T, ND>(w0 + (a * s0 + k) * ND, c);
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
                        CP row = w0 + (a * s0 + k * s1) * ND;
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
}

// eval_binned_lds: eval_gather's arithmetic on the records in slot order, with the coefficient rows
// of the current cell staged in LDS; results control-point-major in tmp[slot].
// All points of a cell read the same bundle of rows: R0 (x R1 for three variables) rows of the
// first (two) variable(s), each a full line of the last variable (cfg5: 25 rows x 640 B = 16 KB).
// Walking the windows through the vector L1 costs 2 KB per point at 64 B/clk/CU (measured: 843 us
// per 10 M cfg5 points in cell order); the bundle is read from L2 once per cell and workgroup
// instead and the windows come from LDS (256 B/clk/CU; 414 us).  A workgroup owns a contiguous slot range and goes through the
// cells that overlap it.  LDS: [axis tables][bundle]
template <typename T, int NIND, int O, int ND, bool MIXED>
__global__ __launch_bounds__(256) void eval_binned_lds(const Desc<T> d, const BinPlan bp, const T *__restrict__ gtab,
                                                       const T *__restrict__ aos, const unsigned *__restrict__ start,
                                                       const BinRec<T, NIND> *__restrict__ rec, const long long N,
                                                       BinOut<T, ND> *__restrict__ tmp, const Wrt wrt)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T *stab = reinterpret_cast<T *>(smem);
    T *bun = reinterpret_cast<T *>(smem + ((sizeof(T) * (size_t)d.tab_len + 15) & ~(size_t)15));
    __shared__ int s_first;
    for (int i = threadIdx.x; i < d.tab_len; i += blockDim.x) stab[i] = gtab[i];
    const long long per = (N + gridDim.x - 1) / gridDim.x;
    const long long lo = (long long)blockIdx.x * per, hi = lo + per < N ? lo + per : N;
    if (threadIdx.x == 0) {
        // last cell whose start is <= lo
        int a = 0, b = bp.cells - 1;
        while (a < b) {
            const int m = (a + b + 1) >> 1;
            if ((long long)start[m] <= lo) a = m; else b = m - 1;
        }
        s_first = a;
    }
    __syncthreads();
    const int ncl = d.ncoef[NIND - 1];                       // control points per row (last variable)
    const int S0 = d.ncoef[0] - d.order[0] + 1, S1 = d.ncoef[1] - d.order[1] + 1;
    int pad[NIND];
#pragma unroll
    for (int iv = 0; iv < NIND; ++iv) pad[iv] = O - d.order[iv];
    const int cs0 = NIND == 3 ? d.ncoef[1] * d.ncoef[2] : d.ncoef[1];   // table strides (control points)
    const int cs1 = NIND == 3 ? d.ncoef[2] : 1;
    for (int c = s_first; c < bp.cells && lo < hi; ++c) {
        const long long cb = start[c], ce = c + 1 < bp.cells ? (long long)start[c + 1] : N;
        if (cb >= hi) break;
        const long long sl = cb > lo ? cb : lo, sh = ce < hi ? ce : hi;
        if (sl >= sh) continue;
        // rows of this cell
        const int q0 = (c / bp.n1) << bp.sh0, q1 = NIND == 3 ? (c % bp.n1) << bp.sh1 : 0;
        const int n0 = min(1 << bp.sh0, S0 - q0) + d.order[0] - 1;
        const int n1 = NIND == 3 ? min(1 << bp.sh1, S1 - q1) + d.order[1] - 1 : 1;
        const int rowlen = ncl * ND;                         // elements per row
        __syncthreads();                                     // previous cell's readers are done
        for (int r = threadIdx.x / 64; r < n0 * n1; r += blockDim.x / 64) {
            const int r0 = r / n1, r1 = r - r0 * n1;
            const T *__restrict__ src = aos + ((long long)(q0 + r0) * cs0 + (long long)(q1 + r1) * cs1) * ND;
            T *dst = bun + (size_t)r * rowlen;
            for (int i = threadIdx.x & 63; i < rowlen; i += 64) dst[i] = src[i];
        }
        __syncthreads();
        const int ls0 = NIND == 3 ? n1 * ncl : ncl, ls1 = ncl;     // bundle strides (control points)
        for (long long p = sl + threadIdx.x; p < sh; p += blockDim.x) {
            const BinRec<T, NIND> rc = rec[p];
            T b[NIND][O];
            int ix[NIND];
#pragma unroll
            for (int iv = 0; iv < NIND; ++iv) {
                const T u = rc.v[iv];
                const T *tab = stab + d.off[iv];
                const int sp = find_span<T>(tab, d.order[iv], d.ncoef[iv], d.steps[iv], u);
                if constexpr (MIXED) basis_bounded<T, O>(tab, d.nk[iv], d.order[iv], sp, u, wrt.w[iv], b[iv]);
                else basis_fixed<T, O>(tab, d.nk[iv], sp, u, wrt.w[iv], b[iv]);
                ix[iv] = sp - O;                               // right-aligned window start (may precede the bundle: padded)
            }
            int base;
            if constexpr (NIND == 3) base = (ix[0] - q0) * ls0 + (ix[1] - q1) * ls1 + ix[2];
            else base = (ix[0] - q0) * ls0 + ix[1];
            T r[ND];
            window_contract<T, NIND, O, ND, MIXED>(bun + (long long)base * ND, ls0, ls1, pad, b, r);
            BinOut<T, ND> o;
#pragma unroll
            for (int k = 0; k < BinOut<T, ND>::WORDS; ++k) o.v[k] = k < ND ? r[k < ND ? k : 0] : T(0);
            tmp[p] = o;
        }
    }
}

template <typename T, int ND>
__global__ __launch_bounds__(256) void bin_unpermute(const long long N, const unsigned *__restrict__ slot,
                                                     const BinOut<T, ND> *__restrict__ tmp, T *__restrict__ out,
                                                     const long long ostride)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += stride) {
        const BinOut<T, ND> r = tmp[slot[n]];
#pragma unroll
        for (int dd = 0; dd < ND; ++dd) nt_store(&out[dd * ostride + n], r.v[dd]);
    }
}

}  // namespace bsk
