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
//                histogram row M[chunk][cell]
//   bin_scan_ranges / bin_scan_top  exclusive scan of every cell's counts over the chunks (ranges of chunks, then
//                the ranges), cell totals and their exclusive scan
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
//
// The kernels above are the ROUND-2 sort: they serve eval_binned_lds (two-variable tables too large to be
// streamed through LDS, mixed orders of three variables, tables beyond the span-key range).  Three variables of
// one order - the cfg5 shape - take the round-3 sort further down (bin_totals / bin_starts / bin_scatter_tag /
// bin_unpermute_stream) in front of eval_cellsort.
#pragma once
#include "bsk_gather.hpp"
#include <type_traits>

namespace bsk {

constexpr int BIN_BLOCK = 512;          // threads per workgroup of the binning kernels
constexpr int BIN_MAX_CELLS = 8192;     // LDS histogram capacity (32 KB of counters)
constexpr int BIN_MAX_CHUNKS = 1024;    // chunks of the direct (not write-combining) scatter
constexpr int BIN_MAX_RANGES = 16;      // ranges of chunks of the histogram scan (bin_scan_ranges / bin_scan_top)

// Cells of the first two variables, coarsened by shifts until they fit BIN_MAX_CELLS.
struct BinPlan {
    int sh0, sh1;       // span index >> shift
    int n1;             // cells along variable 1
    int cells;          // total
    int chunks;         // workgroups of bin_count / bin_scatter
    long long chunk;    // points per chunk
    int rlen, ranges;   // the chunks are scanned in `ranges` ranges of `rlen` chunks (bin_scan_ranges)
};

// First slot of the run of bin i in chunk c: the bin's start + the runs of the bin in earlier ranges + in earlier
// chunks of the range.  M is kept [chunk][bin]: a chunk's histogram row is written, scanned and read coalesced
// (as [bin][chunk] every workgroup of bin_count / bin_scatter / bin_unpermute touched `bins` different lines).
__device__ __forceinline__ unsigned bin_run_start(const BinPlan &bp, const unsigned *__restrict__ start, const unsigned *__restrict__ Tr,
                                                  const unsigned *__restrict__ M, int c, int i)
{
    return start[i] + Tr[(size_t)(c / bp.rlen) * bp.cells + i] + M[(size_t)c * bp.cells + i];
}

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
__device__ __forceinline__ int bin_cell(const T *stab, const unsigned *slut, const Desc<T> &d, const TileDesc<T> &td,
                                        const BinPlan &bp, T u0, T u1)
{
    const int i0 = find_span_lut<T>(stab + d.off[0], slut, td, 0, d.lo[0], d.ncoef[0], u0) - d.order[0];
    const int i1 = find_span_lut<T>(stab + d.off[1], slut, td, 1, d.lo[1], d.ncoef[1], u1) - d.order[1];
    return (i0 >> bp.sh0) * bp.n1 + (i1 >> bp.sh1);
}

// LDS: [axis tables][cells x u32]
// Four independent points per lane and iteration (loads, span searches and histogram updates of the four
// interleave: a single chain per lane is bound by its dependent LDS / HBM round trips).
// (Carrying the span of the third variable from here to eval_cellsort in the records' spare word was
// measured: the search costs this kernel 23 us and saves the evaluation 16 - not kept.)
constexpr int BIN_ILP = 4;

template <typename T, int NIND, int O>
__global__ __launch_bounds__(1024) void bin_count(const Desc<T> d, const TileDesc<T> td, const BinPlan bp,
                                                       const T *__restrict__ gtab, const unsigned *__restrict__ glut,
                                                       const Params<T> prm, const long long N,
                                                       unsigned short *__restrict__ cell, unsigned *__restrict__ M,
                                                       unsigned long long *bad)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T *stab = reinterpret_cast<T *>(smem);
    unsigned *slut = reinterpret_cast<unsigned *>(smem + ((sizeof(T) * (size_t)d.tab_len + 15) & ~(size_t)15));
    unsigned *hist = slut + ((td.lut_len + 3) & ~3);
    for (int i = threadIdx.x; i < d.tab_len; i += blockDim.x) stab[i] = gtab[i];
    for (int i = threadIdx.x; i < td.lut_len; i += blockDim.x) slut[i] = glut[i];
    for (int i = threadIdx.x; i < bp.cells; i += blockDim.x) hist[i] = 0u;
    __syncthreads();
    const long long lo = (long long)blockIdx.x * bp.chunk, hi = lo + bp.chunk < N ? lo + bp.chunk : N;
    // three variables: this kernel reads (and tests against the domain) the two that make the bin; the third is
    // tested by the scatter kernel, which reads it anyway (8 instead of 12 bytes per point here)
    constexpr int NV = NIND == 3 ? 2 : NIND;
    for (long long n0 = lo + threadIdx.x; n0 < hi; n0 += (long long)blockDim.x * BIN_ILP) {
        T u[BIN_ILP][NV];
#pragma unroll
        for (int k = 0; k < BIN_ILP; ++k) {
            const long long n = n0 + (long long)k * blockDim.x;
            const long long nn = n < hi ? n : hi - 1;
#pragma unroll
            for (int iv = 0; iv < NV; ++iv) u[k][iv] = prm.p[iv][nn];
        }
#pragma unroll
        for (int k = 0; k < BIN_ILP; ++k) {
            const long long n = n0 + (long long)k * blockDim.x;
            if (n < hi) {
                bool outside = false;
#pragma unroll
                for (int iv = 0; iv < NV; ++iv) outside |= (u[k][iv] < d.lo[iv]) | (u[k][iv] > d.hi[iv]);
                if (outside) record_bad(bad, n);
                const int c = bin_cell<T, O>(stab, slut, d, td, bp, u[k][0], u[k][1]);
                cell[n] = (unsigned short)c;
                atomicAdd(&hist[c], 1u);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < bp.cells; i += blockDim.x) M[(size_t)blockIdx.x * bp.cells + i] = hist[i];
}

// Scan of the chunk histograms along the chunks, in two steps.
//   bin_scan_ranges  thread = bin, workgroup = 256 bins x one range of chunks: exclusive prefix inside the range
//                    (in place), range total -> Tr[range][bin]; rows are read and written coalesced, sixteen in flight
//   bin_scan_top     the ranges, then the bins (one workgroup)
static __global__ __launch_bounds__(256) void bin_scan_ranges(const BinPlan bp, unsigned *__restrict__ M, unsigned *__restrict__ Tr)
{
    const int i = blockIdx.x * 256 + threadIdx.x, r = blockIdx.y;
    if (i >= bp.cells) return;
    const int c0 = r * bp.rlen, c1 = c0 + bp.rlen < bp.chunks ? c0 + bp.rlen : bp.chunks;
    unsigned run = 0;
    int c = c0;
    for (; c + 16 <= c1; c += 16) {
        unsigned v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = M[(size_t)(c + k) * bp.cells + i];
#pragma unroll
        for (int k = 0; k < 16; ++k) { M[(size_t)(c + k) * bp.cells + i] = run; run += v[k]; }
    }
    for (; c < c1; ++c) { const unsigned v = M[(size_t)c * bp.cells + i]; M[(size_t)c * bp.cells + i] = run; run += v; }
    Tr[(size_t)r * bp.cells + i] = run;
}

// one workgroup, thread = BIN_MAX_CELLS / 1024 consecutive bins: exclusive prefix over the ranges (in place), bin
// totals, and their exclusive prefix over the bins -> start[]
static __global__ __launch_bounds__(1024) void bin_scan_top(const BinPlan bp, unsigned *__restrict__ Tr, unsigned *__restrict__ start)
{
    __shared__ unsigned s[16];
    constexpr int PER = BIN_MAX_CELLS / 1024;
    const int t = threadIdx.x, cells = bp.cells;
    const int per = (cells + 1023) / 1024;                  // <= PER
    unsigned tot[PER];
    unsigned sum = 0;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        tot[i] = 0;
        const int k = t * per + i;
        if (i < per && k < cells) {
            unsigned v[BIN_MAX_RANGES];                       // all loads first: the stores below may alias them
#pragma unroll
            for (int r = 0; r < BIN_MAX_RANGES; ++r) v[r] = r < bp.ranges ? Tr[(size_t)r * cells + k] : 0u;
            unsigned run = 0;
#pragma unroll
            for (int r = 0; r < BIN_MAX_RANGES; ++r) {
                if (r < bp.ranges) Tr[(size_t)r * cells + k] = run;
                run += v[r];
            }
            tot[i] = run;
            sum += run;
        }
    }
    // exclusive prefix of `sum` over the 1024 threads: wave scans, then the 16 wave totals
    const int lane = t & 63, wv = t >> 6;
    unsigned inc = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned o = __shfl_up(inc, off);
        if (lane >= off) inc += o;
    }
    if (lane == 63) s[wv] = inc;
    __syncthreads();
    unsigned base = 0;
    for (int w2 = 0; w2 < wv; ++w2) base += s[w2];
    unsigned run = base + inc - sum;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int k = t * per + i;
        if (i < per && k < cells) { start[k] = run; run += tot[i]; }
    }
}

template <typename T, int NIND>
__global__ __launch_bounds__(1024) void bin_scatter(const BinPlan bp, const Params<T> prm, const long long N,
                                                         const unsigned short *__restrict__ cell,
                                                         const unsigned *__restrict__ M, const unsigned *__restrict__ Tr,
                                                         const unsigned *__restrict__ start,
                                                         BinRec<T, NIND> *__restrict__ rec, unsigned *__restrict__ slot,
                                                         const Desc<T> d, const TileDesc<T> td, const T *__restrict__ gtab,
                                                         const unsigned *__restrict__ glut,
                                                         unsigned long long *bad)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned *next = reinterpret_cast<unsigned *>(smem);
    for (int i = threadIdx.x; i < bp.cells; i += blockDim.x) next[i] = bin_run_start(bp, start, Tr, M, blockIdx.x, i);
    __syncthreads();
    const long long lo = (long long)blockIdx.x * bp.chunk, hi = lo + bp.chunk < N ? lo + bp.chunk : N;
    for (long long n0 = lo + threadIdx.x; n0 < hi; n0 += (long long)blockDim.x * BIN_ILP) {
        unsigned w[BIN_ILP];
        BinRec<T, NIND> r[BIN_ILP];
#pragma unroll
        for (int k = 0; k < BIN_ILP; ++k) {
            const long long n = n0 + (long long)k * blockDim.x;
            const long long nn = n < hi ? n : hi - 1;
            w[k] = cell[nn];
#pragma unroll
            for (int q = 0; q < BinRec<T, NIND>::WORDS; ++q) r[k].v[q] = T(0);
#pragma unroll
            for (int iv = 0; iv < NIND; ++iv) r[k].v[iv] = prm.p[iv][nn];
        }
#pragma unroll
        for (int k = 0; k < BIN_ILP; ++k) {
            const long long n = n0 + (long long)k * blockDim.x;
            if (n < hi) {
                if constexpr (NIND == 3) { if ((r[k].v[2] < d.lo[2]) | (r[k].v[2] > d.hi[2])) record_bad(bad, n); }
                const unsigned p = atomicAdd(&next[w[k]], 1u);
                rec[p] = r[k];
                slot[n] = p;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Write-combining forms of the scatter and the un-permute.
//
// A random 16-byte access costs the memory system a whole 128-byte line (tools/gather_probe.hip: 47 - 60 G
// random granules per second whatever the granule size up to a line, from a 20 MB table as from a 160 MB one),
// and bin_scatter / bin_unpermute issue one per point.  But the records of one (chunk, bin) RUN are
// contiguous in `rec` / `tmp` by construction (slot = start[bin] + M[bin][chunk] + rank): a workgroup that
// first orders its chunk by bin in LDS can move every run with neighbouring lanes - a run of 6 records is
// one or two lines instead of six.  The chunk (<= 8192 points) is sized so that its records fit in LDS.
//   bin_scatter_wc    counts per bin (LDS atomics, rank kept), scan, records into bin order in LDS,
//                     lpos[n] = local position of point n (coalesced), then rec[...] run by run; also writes the
//                     chunk's bin-order tables for the way back: Lb[chunk][bin] (first local position of the bin)
//                     and pbin[lo + p] (bin of local position p)
//   bin_unpermute_wc  loads the chunk's runs of `tmp` into LDS the same way, then every point reads its
//                     result from LDS at lpos[n] and the SoA rows are stored coalesced
// ---------------------------------------------------------------------------------------------
constexpr int WC_PPT = 8;             // points per lane of a 1024-lane workgroup: chunk <= 8192
// the round-3 sort kernels (bin_totals, bin_scatter_tag, bin_unpermute_stream): an fp64 chunk is at most 4096 records
// (32-byte records in 160 KB of LDS), so the lane holds half as many
template <typename T>
__host__ __device__ constexpr int wc_ppt() { return sizeof(T) == 8 ? WC_PPT / 2 : WC_PPT; }
constexpr int BIN_MAX_WC_CELLS = 2048; // bins the write-combining kernels hold tables for

template <typename T, int NIND>
__global__ __launch_bounds__(1024) void bin_scatter_wc(const BinPlan bp, const Params<T> prm, const long long N,
                                                       const unsigned short *__restrict__ cell,
                                                       const unsigned *__restrict__ M, const unsigned *__restrict__ Tr,
                                                       const unsigned *__restrict__ start,
                                                       BinRec<T, NIND> *__restrict__ rec, unsigned short *__restrict__ lpos,
                                                       unsigned short *__restrict__ pbin, unsigned *__restrict__ Lb,
                                                       const Desc<T> d, const TileDesc<T> td, const T *__restrict__ gtab,
                                                       const unsigned *__restrict__ glut,
                                                       unsigned long long *bad)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int cells = bp.cells;
    unsigned *lcnt = reinterpret_cast<unsigned *>(smem);
    unsigned *locb = lcnt + cells;
    unsigned *next0 = locb + cells;
    BinRec<T, NIND> *srec = reinterpret_cast<BinRec<T, NIND> *>(smem + ((12 * (size_t)cells + 15) & ~(size_t)15));
    unsigned short *sbin = reinterpret_cast<unsigned short *>(srec + bp.chunk);
    __shared__ unsigned s_wave[16];
    // PERSISTENT: one workgroup per CU (the chunk fills LDS) walks the chunks blockIdx, blockIdx + grid, ...  The next
    // chunk's points, bins and run starts are fetched into registers BEFORE the store phase of the current one, so that
    // loads and stores of a CU overlap (one chunk per workgroup: load, order, store one after the other, 125 us).
    constexpr int NEXT_PPT = (BIN_MAX_WC_CELLS + 1023) / 1024;
    unsigned ck[WC_PPT], rk[WC_PPT], nx[NEXT_PPT];
    BinRec<T, NIND> r[WC_PPT];
    auto fetch = [&](int c) {
        const long long lo = (long long)c * bp.chunk, hi = lo + bp.chunk < N ? lo + bp.chunk : N;
        const int cnt = (int)(hi - lo);
#pragma unroll
        for (int k = 0; k < WC_PPT; ++k) {
            const int i = k * 1024 + (int)threadIdx.x;
            const long long nn = lo + (i < cnt ? i : cnt - 1);
            ck[k] = cell[nn];
#pragma unroll
            for (int q = 0; q < BinRec<T, NIND>::WORDS; ++q) r[k].v[q] = T(0);
#pragma unroll
            for (int iv = 0; iv < NIND; ++iv) r[k].v[iv] = prm.p[iv][nn];
        }
#pragma unroll
        for (int k = 0; k < NEXT_PPT; ++k) {
            const int i = k * 1024 + (int)threadIdx.x;
            nx[k] = i < cells ? bin_run_start(bp, start, Tr, M, c, i) : 0u;
        }
    };
    int c = blockIdx.x;
    if (c < bp.chunks) fetch(c);
    for (; c < bp.chunks; c += gridDim.x) {
        const long long lo = (long long)c * bp.chunk, hi = lo + bp.chunk < N ? lo + bp.chunk : N;
        const int cnt = (int)(hi - lo);
        __syncthreads();                                          // the previous chunk's store phase is done with LDS
#pragma unroll
        for (int k = 0; k < NEXT_PPT; ++k) {
            const int i = k * 1024 + (int)threadIdx.x;
            if (i < cells) { lcnt[i] = 0u; next0[i] = nx[k]; }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < WC_PPT; ++k)
            if (k * 1024 + (int)threadIdx.x < cnt) {
                if constexpr (NIND == 3) { if ((r[k].v[2] < d.lo[2]) | (r[k].v[2] > d.hi[2])) record_bad(bad, lo + k * 1024 + (long long)threadIdx.x); }
                rk[k] = atomicAdd(&lcnt[ck[k]], 1u);
            }
        __syncthreads();
        {   // exclusive scan of the bin counts: every thread NEXT_PPT consecutive bins, wave scans, 16 wave totals
            const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
            const int per = (cells + 1023) / 1024;
            unsigned v[NEXT_PPT], sum = 0;
#pragma unroll
            for (int i = 0; i < NEXT_PPT; ++i) {
                const int b = (int)threadIdx.x * per + i;
                v[i] = i < per && b < cells ? lcnt[b] : 0u;
                sum += v[i];
            }
            unsigned inc = sum;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const unsigned o = __shfl_up(inc, off);
                if (lane >= off) inc += o;
            }
            if (lane == 63) s_wave[wv] = inc;
            __syncthreads();
            unsigned run = inc - sum;
            for (int w2 = 0; w2 < wv; ++w2) run += s_wave[w2];
#pragma unroll
            for (int i = 0; i < NEXT_PPT; ++i) {
                const int b = (int)threadIdx.x * per + i;
                if (i < per && b < cells) { locb[b] = run; run += v[i]; }
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < WC_PPT; ++k) {
            const int i = k * 1024 + (int)threadIdx.x;
            if (i < cnt) {
                const unsigned p = locb[ck[k]] + rk[k];
                srec[p] = r[k];
                sbin[p] = (unsigned short)ck[k];
                lpos[lo + i] = (unsigned short)p;             // where bin_unpermute_wc finds the point's result in the chunk
            }
        }
        for (int i = threadIdx.x; i < cells; i += blockDim.x) Lb[(size_t)c * cells + i] = locb[i];
        if (c + (int)gridDim.x < bp.chunks) fetch(c + (int)gridDim.x);     // in flight during the store phase
        __syncthreads();
        for (int p = threadIdx.x; p < cnt; p += blockDim.x) {
            const unsigned b = sbin[p];
            rec[next0[b] + ((unsigned)p - locb[b])] = srec[p];
            pbin[lo + p] = (unsigned short)b;
        }
    }
}

template <typename T, int ND>
__global__ __launch_bounds__(1024) void bin_unpermute_wc(const BinPlan bp, const long long N,
                                                         const unsigned short *__restrict__ lpos, const unsigned *__restrict__ M,
                                                         const unsigned *__restrict__ Tr,
                                                         const unsigned *__restrict__ start, const unsigned *__restrict__ Lb,
                                                         const unsigned short *__restrict__ pbin,
                                                         const BinOut<T, ND> *__restrict__ tmp, T *__restrict__ out,
                                                         const long long ostride)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int cells = bp.cells;
    unsigned *locb = reinterpret_cast<unsigned *>(smem);
    unsigned *next0 = locb + cells;
    BinOut<T, ND> *sout = reinterpret_cast<BinOut<T, ND> *>(smem + ((8 * (size_t)cells + 15) & ~(size_t)15));
    // PERSISTENT, as bin_scatter_wc: while the rows of chunk c are stored, the runs of chunk c + grid are already being
    // gathered into registers and the run tables of chunk c + 2 grid are on their way.
    constexpr int NEXT_PPT = (BIN_MAX_WC_CELLS + 1023) / 1024;
    const int G = (int)gridDim.x;
    unsigned tl[NEXT_PPT], tn[NEXT_PPT];
    typedef T gvec __attribute__((ext_vector_type(BinOut<T, ND>::WORDS)));      // (an array of structs ends up in scratch)
    gvec g[WC_PPT];
    unsigned lpn[WC_PPT];                                    // local positions of the chunk being gathered
    auto fetch_tables = [&](int c) {
#pragma unroll
        for (int k = 0; k < NEXT_PPT; ++k) {
            const int i = k * 1024 + (int)threadIdx.x;
            tl[k] = i < cells ? Lb[(size_t)c * cells + i] : 0u;
            tn[k] = i < cells ? bin_run_start(bp, start, Tr, M, c, i) : 0u;
        }
    };
    auto put_tables = [&]() {
#pragma unroll
        for (int k = 0; k < NEXT_PPT; ++k) {
            const int i = k * 1024 + (int)threadIdx.x;
            if (i < cells) { locb[i] = tl[k]; next0[i] = tn[k]; }
        }
    };
    auto gather = [&](int c) {
        const long long lo = (long long)c * bp.chunk, hi = lo + bp.chunk < N ? lo + bp.chunk : N;
        const int cnt = (int)(hi - lo);
#pragma unroll
        for (int k = 0; k < WC_PPT; ++k) {
            const int p0 = k * 1024 + (int)threadIdx.x, p = p0 < cnt ? p0 : cnt - 1;
            const unsigned b = pbin[lo + p];
            // (non-temporal loads here measured slower, 99 -> 111 us: part of `tmp` is still in the caches)
            g[k] = *reinterpret_cast<const gvec *>(&tmp[next0[b] + ((unsigned)p - locb[b])]);
            lpn[k] = lpos[lo + p];
        }
    };
    int c = blockIdx.x;
    if (c < bp.chunks) {
        fetch_tables(c);
        put_tables();
        __syncthreads();
        gather(c);
        if (c + G < bp.chunks) fetch_tables(c + G);
        __syncthreads();        // every wave has formed its addresses from the tables of chunk c before iteration 0 replaces them
    }
    for (; c < bp.chunks; c += G) {
        const long long lo = (long long)c * bp.chunk, hi = lo + bp.chunk < N ? lo + bp.chunk : N;
        const int cnt = (int)(hi - lo);
#pragma unroll
        for (int k = 0; k < WC_PPT; ++k) {
            const int p = k * 1024 + (int)threadIdx.x;
            if (p < cnt) *reinterpret_cast<gvec *>(&sout[p]) = g[k];
        }
        if (c + G < bp.chunks) put_tables();                 // the gather of chunk c has its addresses: its tables may go
        __syncthreads();
        unsigned lp[WC_PPT];
#pragma unroll
        for (int k = 0; k < WC_PPT; ++k) lp[k] = lpn[k];      // fetched with the gather, one chunk ago
        if (c + G < bp.chunks) {
            gather(c + G);
            if (c + 2 * G < bp.chunks) fetch_tables(c + 2 * G);
        }
#pragma unroll
        for (int k = 0; k < WC_PPT; ++k) {
            const int i = k * 1024 + (int)threadIdx.x;
            if (i < cnt) {
                const gvec r = *reinterpret_cast<const gvec *>(&sout[lp[k]]);
#pragma unroll
                for (int dd = 0; dd < ND; ++dd) nt_store(&out[dd * ostride + lo + i], (T)r[dd]);
            }
        }
        __syncthreads();                                     // sout is free for the next chunk
    }
}

// ---------------------------------------------------------------------------------------------
// Round 3: the sort side of the three-variable pipeline (eval_cellsort) without chunk histograms in memory.
//
// Round 2 kept a [chunk][bin] matrix of counts (bin_count), scanned it along the chunks (two kernels) and read
// it back in the scatter AND in the un-permute, which also needed two more per-point arrays (pbin, and the bin of
// every point from bin_count) to find the runs of `tmp` again: 49 us of counting / scanning and 91 us of gathering
// runs per 10 M cfg5 points.  Here
//   bin_totals            persistent, 16-byte loads: bins of the whole batch counted in one LDS histogram per
//                         workgroup and added to tot[] once per workgroup (contiguous no-return atomics); the LAST
//                         workgroup to finish (ticket) scans tot[] -> start[], fill[] = start[]
//   bin_scatter_tag       persistent over the chunks: bin and span key computed here (the batch is read once, no
//                         cell[] array); a chunk's run of bin b starts at atomicAdd(fill[b], count) - runs of
//                         different chunks land in ARRIVAL order: the slot of a record is not deterministic, its
//                         result is; the record's spare word carries  dest | key << dest_bits,  dest = the
//                         record's position in CHUNK order (lo + local position in the chunk ordered by bin)
//   eval_cellsort         stores a record's result at tmp[dest]: the evaluation is issue bound, its scattered
//                         16-byte stores (runs of ~6 neighbours, as the scatter's) are hidden
//   bin_unpermute_stream  chunk c: tmp[lo .. hi) is CONTIGUOUS - streamed into LDS, then
//                         out[dep][lo + i] = sout[lpos[lo + i]][dep]: no run tables, no gather
// ---------------------------------------------------------------------------------------------
template <typename T>
struct SpanLds {                        // knots and bucket tables of up to three variables in LDS
    const T *kn[3];
    const unsigned *lut;
};

template <typename T, int NV>
__host__ __device__ __forceinline__ size_t span_lds_bytes(const Desc<T> &d, const TileDesc<T> &td)
{
    size_t b = 0;
    for (int iv = 0; iv < NV; ++iv) b += sizeof(T) * (size_t)((d.nk[iv] + 3) & ~3);
    return b + 4 * (size_t)((td.lut_len + 3) & ~3);
}

template <typename T, int NV>
__device__ __forceinline__ SpanLds<T> span_lds_stage(char *base, const Desc<T> &d, const TileDesc<T> &td, const T *__restrict__ gtab,
                                                     const unsigned *__restrict__ glut)
{
    SpanLds<T> sl;
    T *k = reinterpret_cast<T *>(base);
    for (int iv = 0; iv < 3; ++iv) sl.kn[iv] = k;
    for (int iv = 0; iv < NV; ++iv) {
        sl.kn[iv] = k;
        for (int i = threadIdx.x; i < d.nk[iv]; i += blockDim.x) k[i] = gtab[d.off[iv] + i];
        k += (d.nk[iv] + 3) & ~3;
    }
    unsigned *l = reinterpret_cast<unsigned *>(k);
    for (int i = threadIdx.x; i < td.lut_len; i += blockDim.x) l[i] = glut[i];
    sl.lut = l;
    return sl;
}

template <typename T, int STEPS = 0>
__device__ __forceinline__ int bin_of(const SpanLds<T> &sl, const Desc<T> &d, const TileDesc<T> &td, const BinPlan &bp, T u0, T u1)
{
    const int i0 = find_span_lut_n<T, STEPS>(sl.kn[0], sl.lut, td, 0, d.lo[0], d.ncoef[0], u0) - d.order[0];
    const int i1 = find_span_lut_n<T, STEPS>(sl.kn[1], sl.lut, td, 1, d.lo[1], d.ncoef[1], u1) - d.order[1];
    return (i0 >> bp.sh0) * bp.n1 + (i1 >> bp.sh1);
}

// bisection steps the straight-line searches of a kernel need (0: more than two - the run-time loop)
template <typename T>
__device__ __forceinline__ int lut_steps_class(const TileDesc<T> &td, int nv)
{
    int m = 0;
    for (int iv = 0; iv < nv; ++iv) m = max(m, td.lut_steps[iv]);
    return m <= 1 ? 1 : m <= 2 ? 2 : 0;
}
template <int N> using cs_int = std::integral_constant<int, N>;

// exclusive prefix over `cells` values held `per` consecutive ones per thread of a 1024-lane workgroup (v[i] of bin
// t * per + i); returns the prefix of the thread's first bin.  s_wave: 16 words of LDS.  One barrier inside.
template <int PER>
__device__ __forceinline__ unsigned block_excl_scan(const unsigned (&v)[PER], unsigned *s_wave)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned sum = 0;
#pragma unroll
    for (int i = 0; i < PER; ++i) sum += v[i];
    unsigned inc = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned o = __shfl_up(inc, off);
        if (lane >= off) inc += o;
    }
    if (lane == 63) s_wave[wv] = inc;
    __syncthreads();
    unsigned run = inc - sum;
    for (int w2 = 0; w2 < wv; ++w2) run += s_wave[w2];
    return run;
}

// bin_totals: workgroup w counts the bins of the chunks w, w + G, w + 2 G, ... - exactly the chunks workgroup w of
// bin_scatter_tag (same grid) will move - and writes ONE histogram row rows[w][cells] (plain coalesced stores: adding
// the rows into shared totals with atomics measured 150 us per 10 M points - hundreds of workgroups adding to the same
// 40 lines serialise at the memory side).  16-byte loads; no barrier between chunks.
// LDS: [knots + bucket tables of variables 0, 1][cells x u32]
template <typename T>
__global__ __launch_bounds__(1024) void bin_totals(const Desc<T> d, const TileDesc<T> td, const BinPlan bp,
                                                   const T *__restrict__ gtab, const unsigned *__restrict__ glut,
                                                   const Params<T> prm, const long long N, unsigned *__restrict__ rows)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const SpanLds<T> sl = span_lds_stage<T, 2>(smem, d, td, gtab, glut);
    unsigned *hist = reinterpret_cast<unsigned *>(smem + ((span_lds_bytes<T, 2>(d, td) + 15) & ~(size_t)15));
    const int cells = bp.cells;
    for (int i = threadIdx.x; i < cells; i += blockDim.x) hist[i] = 0u;
    __syncthreads();
    constexpr int V = 16 / (int)sizeof(T);
    typedef T vecT __attribute__((ext_vector_type(V)));
    const int steps = lut_steps_class<T>(td, 2);
    // chunks start at multiples of 8 points (host): 16-byte loads whenever the caller's arrays are 16-byte aligned
    const bool aligned = ((reinterpret_cast<size_t>(prm.p[0]) | reinterpret_cast<size_t>(prm.p[1])) & 15) == 0;
    // a chunk (<= 8192 points) is at most 2 (fp32) / 4 (fp64) vectors per lane and variable: the next chunk's are
    // requested before this chunk's are counted
    constexpr int VPL = wc_ppt<T>() / V;
    vecT a[VPL], bq[VPL];
    auto fetch = [&](int c) {
        const long long lo = (long long)c * bp.chunk, hi = lo + bp.chunk < N ? lo + bp.chunk : N;
        const int nvec = aligned ? (int)(hi - lo) / V : 0;
        const vecT *pu = reinterpret_cast<const vecT *>(prm.p[0] + lo), *pv = reinterpret_cast<const vecT *>(prm.p[1] + lo);
#pragma unroll
        for (int k = 0; k < VPL; ++k) {
            const int g = k * 1024 + (int)threadIdx.x;
            if (g < nvec) { a[k] = __builtin_nontemporal_load(pu + g); bq[k] = __builtin_nontemporal_load(pv + g); }
        }
    };
    int c = blockIdx.x;
    if (c < bp.chunks) fetch(c);
    for (; c < bp.chunks; c += gridDim.x) {
        const long long lo = (long long)c * bp.chunk, hi = lo + bp.chunk < N ? lo + bp.chunk : N;
        const int cnt = (int)(hi - lo);
        const int nvec = aligned ? cnt / V : 0;
        vecT ca[VPL], cb[VPL];
#pragma unroll
        for (int k = 0; k < VPL; ++k) { ca[k] = a[k]; cb[k] = bq[k]; }
        if (c + (int)gridDim.x < bp.chunks) fetch(c + (int)gridDim.x);
        // straight-line over the lane's points (a lane past the end classifies garbage and counts nothing): the searches'
        // dependent LDS reads interleave
        auto count = [&](auto S) {
            constexpr int ST = decltype(S)::value;
            int bn[VPL][V];
#pragma unroll
            for (int k = 0; k < VPL; ++k)
#pragma unroll
                for (int q = 0; q < V; ++q) bn[k][q] = bin_of<T, ST>(sl, d, td, bp, ca[k][q], cb[k][q]);
#pragma unroll
            for (int k = 0; k < VPL; ++k) {
                if (k * 1024 + (int)threadIdx.x < nvec) {
#pragma unroll
                    for (int q = 0; q < V; ++q) atomicAdd(&hist[min(max(bn[k][q], 0), cells - 1)], 1u);
                }
            }
        };
        if (steps == 1) count(cs_int<1>{}); else if (steps == 2) count(cs_int<2>{}); else count(cs_int<0>{});
        for (int i = nvec * V + (int)threadIdx.x; i < cnt; i += blockDim.x)
            atomicAdd(&hist[bin_of<T>(sl, d, td, bp, prm.p[0][lo + i], prm.p[1][lo + i])], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < cells; i += blockDim.x) rows[(size_t)blockIdx.x * cells + i] = hist[i];
}

// bin_starts: rows[w][b] -> exclusive prefix over the workgroups INSIDE every bin (in place), bin totals, and - by the
// last workgroup to finish (ticket) - the exclusive prefix of the totals over the bins: start[b].  The first slot of
// workgroup w's records of bin b is start[b] + rows[w][b].
// Workgroup = 64 bins x 16 groups of rows (1024 lanes): a lane scans its group's rows (16 in flight), the 16 groups of
// a bin are combined through LDS.
static __global__ __launch_bounds__(1024) void bin_starts(const int cells, const int G, unsigned *__restrict__ rows, unsigned *__restrict__ tot,
                                                    unsigned *__restrict__ start, unsigned *__restrict__ done)
{
    __shared__ unsigned part[16][65];
    __shared__ unsigned s_wave[16];
    __shared__ int s_last;
    const int bl = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int b = blockIdx.x * 64 + bl;
    const int rpg = (G + 15) / 16;                              // rows per group
    const int r0 = grp * rpg, r1 = r0 + rpg < G ? r0 + rpg : G;
    // pass 1: the group's sum (its first 16 rows stay in registers: the whole group when G <= 256)
    unsigned sum = 0, v0[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v0[k] = (b < cells && r0 + k < r1) ? rows[(size_t)(r0 + k) * cells + b] : 0u;
#pragma unroll
    for (int k = 0; k < 16; ++k) sum += v0[k];
    if (b < cells)
        for (int r = r0 + 16; r < r1; r += 16) {
            unsigned v[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = r + k < r1 ? rows[(size_t)(r + k) * cells + b] : 0u;
#pragma unroll
            for (int k = 0; k < 16; ++k) sum += v[k];
        }
    part[grp][bl] = sum;
    __syncthreads();
    unsigned run = 0, total = 0;
#pragma unroll
    for (int g2 = 0; g2 < 16; ++g2) {
        const unsigned p = part[g2][bl];
        run += g2 < grp ? p : 0u;
        total += p;
    }
    // pass 2: exclusive prefixes, written once
    if (b < cells) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (r0 + k < r1) rows[(size_t)(r0 + k) * cells + b] = run;
            run += v0[k];
        }
        for (int r = r0 + 16; r < r1; r += 16) {
            unsigned v[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = r + k < r1 ? rows[(size_t)(r + k) * cells + b] : 0u;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                if (r + k < r1) rows[(size_t)(r + k) * cells + b] = run;
                run += v[k];
            }
        }
        if (grp == 0) tot[b] = total;
    }
    // hand-over (MI355X_MICROARCH.md, inter-workgroup visibility): every storing wave drains its stores, the workgroup
    // meets, ONE lane releases at agent scope and takes the ticket (a fence per wave costs microseconds each)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        s_last = atomicAdd(done, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (!s_last) return;
    // last workgroup: the totals of the others were written and released (__threadfence) before their tickets
    constexpr int PER = BIN_MAX_CELLS / 1024;
    const int per = (cells + 1023) / 1024;
    unsigned v[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int k = (int)threadIdx.x * per + i;
        v[i] = (i < per && k < cells) ? atomicAdd(&tot[k], 0u) : 0u;       // read where device-scope atomics are served, not from this XCD's L2
    }
    unsigned first = block_excl_scan<PER>(v, s_wave);
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int k = (int)threadIdx.x * per + i;
        if (i < per && k < cells) { start[k] = first; first += v[i]; }
    }
    if (threadIdx.x == 0) atomicExch(done, 0u);
}

// dest | key << dest_bits in the record's spare word (fp64 records: the low 32 bits of it)
template <typename T>
__device__ __forceinline__ T tag_word(unsigned tag)
{
    using Tag = typename std::conditional<sizeof(T) == 4, unsigned, unsigned long long>::type;
    return __builtin_bit_cast(T, (Tag)tag);
}
template <typename T>
__device__ __forceinline__ unsigned word_tag(T w)
{
    using Tag = typename std::conditional<sizeof(T) == 4, unsigned, unsigned long long>::type;
    return (unsigned)__builtin_bit_cast(Tag, w);
}

// bin_scatter_tag: same grid as bin_totals.  Workgroup w owns, in every bin b, the slots from start[b] + rows[w][b]
// on: a private cursor per bin in LDS, bumped chunk by chunk - no global atomics, no per-chunk tables, and the slot
// of every record is deterministic.
// LDS: [counts: cells x u32][cursors: cells x u32][local bin starts: cells x u16][records: chunk x 16 / 32 B]
//      [bins: chunk x u16][knots + bucket tables of the three variables]
constexpr int TAG_PPT = BIN_MAX_CELLS / 1024;       // bins per lane of the 1024-lane sort kernels

template <typename T>
__global__ __launch_bounds__(1024) void bin_scatter_tag(const BinPlan bp, const Params<T> prm, const long long N, const long long base,
                                                        const unsigned *__restrict__ rows, const unsigned *__restrict__ start,
                                                        BinRec<T, 3> *__restrict__ rec,
                                                        unsigned short *__restrict__ lpos, const Desc<T> d, const TileDesc<T> td,
                                                        const T *__restrict__ gtab, const unsigned *__restrict__ glut,
                                                        const int dest_bits, unsigned long long *bad)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ unsigned s_wave[16];
    const int cells = bp.cells;
    unsigned *lcnt = reinterpret_cast<unsigned *>(smem);
    unsigned *cursor = lcnt + cells;
    unsigned short *locb = reinterpret_cast<unsigned short *>(cursor + cells);
    const size_t bins_b = (10 * (size_t)cells + 15) & ~(size_t)15;
    BinRec<T, 3> *srec = reinterpret_cast<BinRec<T, 3> *>(smem + bins_b);
    unsigned short *sbin = reinterpret_cast<unsigned short *>(srec + bp.chunk);
    char *tabs = smem + bins_b + (((size_t)bp.chunk * (sizeof(BinRec<T, 3>) + 2) + 15) & ~(size_t)15);
    const SpanLds<T> sl = span_lds_stage<T, 3>(tabs, d, td, gtab, glut);
    for (int i = threadIdx.x; i < cells; i += blockDim.x) cursor[i] = start[i] + rows[(size_t)blockIdx.x * cells + i];
    const int per = (cells + 1023) / 1024;
    const int steps = lut_steps_class<T>(td, 3);
    // the NEXT chunk's points are requested at the top of an iteration (second register set): they stream in while
    // this chunk is counted, ordered and stored
    T nu[wc_ppt<T>()], nv[wc_ppt<T>()], nw[wc_ppt<T>()];
    auto fetch = [&](int c) {
        const long long lo = (long long)c * bp.chunk, hi = lo + bp.chunk < N ? lo + bp.chunk : N;
        const int cnt = (int)(hi - lo);
#pragma unroll
        for (int k = 0; k < wc_ppt<T>(); ++k) {
            const int i = k * 1024 + (int)threadIdx.x;
            const long long nn = lo + (i < cnt ? i : cnt - 1);
            nu[k] = __builtin_nontemporal_load(&prm.p[0][nn]);
            nv[k] = __builtin_nontemporal_load(&prm.p[1][nn]);
            nw[k] = __builtin_nontemporal_load(&prm.p[2][nn]);
        }
    };
    int c = blockIdx.x;
    if (c < bp.chunks) fetch(c);
    for (; c < bp.chunks; c += gridDim.x) {
        const long long lo = (long long)c * bp.chunk, hi = lo + bp.chunk < N ? lo + bp.chunk : N;
        const int cnt = (int)(hi - lo);
        T pu[wc_ppt<T>()], pv[wc_ppt<T>()], pw[wc_ppt<T>()];
#pragma unroll
        for (int k = 0; k < wc_ppt<T>(); ++k) { pu[k] = nu[k]; pv[k] = nv[k]; pw[k] = nw[k]; }
        if (c + (int)gridDim.x < bp.chunks) fetch(c + (int)gridDim.x);
        __syncthreads();                                          // the previous chunk's store phase is done with LDS (first: tables, cursors staged)
        for (int i = threadIdx.x; i < cells; i += blockDim.x) lcnt[i] = 0u;
        __syncthreads();
        // bins and span keys of the lane's 8 points: straight-line code over all of them (a lane past the chunk's end
        // holds a copy of its last point), so that the 8 chains of dependent LDS reads interleave; only the domain
        // record and the histogram update are predicated
        unsigned ck[wc_ppt<T>()], rk[wc_ppt<T>()], tg[wc_ppt<T>()];
        auto classify = [&](auto S) {
            constexpr int ST = decltype(S)::value;
#pragma unroll
            for (int k = 0; k < wc_ppt<T>(); ++k) ck[k] = (unsigned)min(max(bin_of<T, ST>(sl, d, td, bp, pu[k], pv[k]), 0), cells - 1);
#pragma unroll
            for (int k = 0; k < wc_ppt<T>(); ++k)
                tg[k] = (unsigned)(find_span_lut_n<T, ST>(sl.kn[2], sl.lut, td, 2, d.lo[2], d.ncoef[2], pw[k]) - d.order[2]) << dest_bits;
        };
        if (steps == 1) classify(cs_int<1>{}); else if (steps == 2) classify(cs_int<2>{}); else classify(cs_int<0>{});
#pragma unroll
        for (int k = 0; k < wc_ppt<T>(); ++k) {
            rk[k] = 0u;
            if (k * 1024 + (int)threadIdx.x < cnt) {
                const bool outside = (pu[k] < d.lo[0]) | (pu[k] > d.hi[0]) | (pv[k] < d.lo[1]) | (pv[k] > d.hi[1]) | (pw[k] < d.lo[2]) | (pw[k] > d.hi[2]);
                if (outside) record_bad(bad, base + lo + k * 1024 + (long long)threadIdx.x);
                rk[k] = atomicAdd(&lcnt[ck[k]], 1u);
            }
        }
        __syncthreads();
        {   // exclusive scan of the bin counts -> first local position of every bin
            unsigned v[TAG_PPT];
#pragma unroll
            for (int i = 0; i < TAG_PPT; ++i) {
                const int b = (int)threadIdx.x * per + i;
                v[i] = i < per && b < cells ? lcnt[b] : 0u;
            }
            unsigned run = block_excl_scan<TAG_PPT>(v, s_wave);
#pragma unroll
            for (int i = 0; i < TAG_PPT; ++i) {
                const int b = (int)threadIdx.x * per + i;
                if (i < per && b < cells) { locb[b] = (unsigned short)run; run += v[i]; }
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < wc_ppt<T>(); ++k) {
            const int i = k * 1024 + (int)threadIdx.x;
            if (i < cnt) {
                const unsigned p = (unsigned)locb[ck[k]] + rk[k];
                BinRec<T, 3> r;
                r.v[0] = pu[k]; r.v[1] = pv[k]; r.v[2] = pw[k];
                r.v[3] = tag_word<T>((unsigned)(lo + p) | tg[k]);   // where bin_unpermute_stream finds the result, and the span key
                srec[p] = r;
                sbin[p] = (unsigned short)ck[k];
                lpos[lo + i] = (unsigned short)p;
            }
        }
        __syncthreads();
        for (int p = threadIdx.x; p < cnt; p += blockDim.x) {
            const unsigned b = sbin[p];
            const unsigned slot = cursor[b] + ((unsigned)p - (unsigned)locb[b]);
            rec[slot < (unsigned long long)N ? slot : (unsigned)(N - 1)] = srec[p];     // (clamped: cursor comes from other kernels' tables)
        }
        __syncthreads();                                          // every record of the chunk has read its cursor
        for (int i = threadIdx.x; i < cells; i += blockDim.x) cursor[i] += lcnt[i];
    }
}

template <typename T, int ND>
__global__ __launch_bounds__(1024) void bin_unpermute_stream(const BinPlan bp, const long long N, const unsigned short *__restrict__ lpos,
                                                             const BinOut<T, ND> *__restrict__ tmp, T *__restrict__ out,
                                                             const long long ostride)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    BinOut<T, ND> *sout = reinterpret_cast<BinOut<T, ND> *>(smem);
    typedef T gvec __attribute__((ext_vector_type(BinOut<T, ND>::WORDS)));
    const int G = (int)gridDim.x;
    gvec g[wc_ppt<T>()];
    unsigned lpn[wc_ppt<T>()];
    auto fetch = [&](int c) {
        const long long lo = (long long)c * bp.chunk, hi = lo + bp.chunk < N ? lo + bp.chunk : N;
        const int cnt = (int)(hi - lo);
#pragma unroll
        for (int k = 0; k < wc_ppt<T>(); ++k) {
            const int p0 = k * 1024 + (int)threadIdx.x, p = p0 < cnt ? p0 : cnt - 1;
            g[k] = *reinterpret_cast<const gvec *>(&tmp[lo + p]);
            lpn[k] = lpos[lo + p];
        }
    };
    int c = blockIdx.x;
    if (c < bp.chunks) fetch(c);
    for (; c < bp.chunks; c += G) {
        const long long lo = (long long)c * bp.chunk, hi = lo + bp.chunk < N ? lo + bp.chunk : N;
        const int cnt = (int)(hi - lo);
#pragma unroll
        for (int k = 0; k < wc_ppt<T>(); ++k) {
            const int p = k * 1024 + (int)threadIdx.x;
            if (p < cnt) *reinterpret_cast<gvec *>(&sout[p]) = g[k];
        }
        unsigned lp[wc_ppt<T>()];
#pragma unroll
        for (int k = 0; k < wc_ppt<T>(); ++k) lp[k] = lpn[k];
        __syncthreads();
        if (c + G < bp.chunks) fetch(c + G);                   // the next chunk streams in while this one's rows are stored
#pragma unroll
        for (int k = 0; k < wc_ppt<T>(); ++k) {
            const int i = k * 1024 + (int)threadIdx.x;
            if (i < cnt) {
                const gvec r = *reinterpret_cast<const gvec *>(&sout[lp[k]]);
#pragma unroll
                for (int dd = 0; dd < ND; ++dd) nt_store(&out[dd * ostride + lo + i], (T)r[dd]);
            }
        }
        __syncthreads();                                     // sout is free for the next chunk
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

// ---------------------------------------------------------------------------------------------
// eval_cellsort: three variables, cells grouped a second time INSIDE the evaluation workgroup.
//
// eval_binned_lds gathers every lane's own window from the LDS bundle: 2 KB of LDS reads per cfg5
// point with ~2-way bank conflicts (lanes of a wave sit in different spans of the third variable).
// Here a workgroup (512 lanes) sorts each tile of ~1970 records (fp64: ~940) of a (span0, span1) bin by the span of the THIRD
// variable in LDS (counting sort on the key the scatter kernel left in the record; segments padded to
// multiples of four lanes; two barriers per tile), so four consecutive lanes always share one cell
// (span0, span1, span2) and neighbouring lanes mostly do.  The contraction
// r[dep] = sum_{ijk} b0_i b1_j b2_k C[i][j][k][dep]  of a 4-lane block then shares its coefficient operand:
//   MFMA = true   v_mfma_f32_4x4x1_16b_f32: sixteen 4 x 4 outer products per instruction.  The steps k of the
//                 third variable sit in the ROWS of the product, four at a time, the B operand is the row weight
//                 b0_i b1_j of the lane's point (cs_contract, see there): per window row (i, j) a lane reads the four
//                 dependent variables of ONE control point with one aligned ds_read_b128 (bundle kept
//                 [k][row][dep]: the row offsets are immediates) and feeds one of them to each of ND instructions;
//                 the reads are issued CS_AHEAD rows before their use.  The lane combines its own partial sums
//                 with b2 at the end.  375 B of LDS reads per point instead of 2 KB, O^2 + 4 O vector
//                 multiply-adds per point instead of O^3.
//   MFMA = false  the same sorted lanes on the vector ALU: eval_gather's arithmetic, every control point
//                 one 16-byte LDS read that the lanes of a cell share (broadcast, conflict free).
// The span tables of the bin's first two variables are wave-uniform: read once per bin into scalar registers
// (SpanTab / basis_regs: basis_fixed's operations in the same order); the third variable's are fetched in one batch.
// Results go to tmp[slot] of the record's ORIGINAL slot, so bin_unpermute is unchanged.  Summation
// order differs from eval_gather (one chain over the window rows per accumulator): results agree to
// rounding, not bitwise.   LDS: [axis tables][bundle][sorted records][two histograms, segment starts]
// Measured steps (10 M cfg5 points): 297 us (first MFMA form: step weights b0 b1 b2 as B operand, 4-byte A reads)
// -> 260 (row weights) -> 252 (b128 rows read ahead) -> 237 (scalar span tables) -> 228 (two barriers, no register
// prefetch of the next tile); DESIGN.md section 8.
// ---------------------------------------------------------------------------------------------
constexpr int CS_BLOCK = 512;           // lanes of an eval_cellsort workgroup
// records sorted at a time: 4 per lane (fp32), 2 per lane (fp64: the 32-byte records must leave room for two workgroups)
template <typename T>
__host__ __device__ constexpr int cs_per() { return sizeof(T) == 4 ? 4 : 2; }
constexpr int CS_MAX_S2 = 256;          // spans of the third variable the LDS histogram holds
constexpr int CS_AHEAD = 4;             // rows of coefficient reads in flight ahead of the MFMAs that use them

typedef float cs_f4 __attribute__((ext_vector_type(4)));

// Table values of one span for the Cox-de Boor recursion: kn[j] = knots[ix - (O - 1) + j] (j < O - 1) and
// rc[D][j] = r_D[ix - D + j] (j < D) - the entries basis_fixed reads, fetched together.
template <typename T, int O>
struct SpanTab {
    T kn[O > 1 ? O - 1 : 1];
    T rc[O][O];
};

// basis_fixed's recursion (same operations in the same order: same bits) on a SpanTab
// DERIV = false: plain evaluation, wrt is 0 - straight-line code (the value / derivative branches of every level
// cost a third of the recursion's instructions in copies where they merge)
template <typename T, int O, bool DERIV>
__device__ __forceinline__ void basis_regs(const SpanTab<T, O> &t, T u, int wrt, T (&b)[O])
{
#pragma unroll
    for (int k = 0; k < O; ++k) b[k] = T(0);
    if (DERIV && wrt >= O) return;
    b[O - 1] = T(1);
#pragma unroll
    for (int degree = 1; degree < O; ++degree) {
        if (!DERIV || degree < O - wrt) {
#pragma unroll
            for (int j = 0; j < degree; ++j) {
                const int bi = O - degree + j;
                const T alpha = (u - t.kn[(O - 1) - degree + j]) * t.rc[degree][j];
                b[bi - 1] += (T(1) - alpha) * b[bi];
                b[bi] *= alpha;
            }
        } else {
#pragma unroll
            for (int j = 0; j < degree; ++j) {
                const int bi = O - degree + j;
                const T alpha = T(degree) * t.rc[degree][j];
                b[bi - 1] -= alpha * b[bi];
                b[bi] *= alpha;
            }
        }
    }
}

// wave-uniform value -> scalar register (fp32; wider types stay in vector registers)
template <typename T>
__device__ __forceinline__ T cs_uniform(T x)
{
    if constexpr (sizeof(T) == 4) return __builtin_bit_cast(T, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x)));
    else return x;
}

// The span's table values of a variable whose span is the same for the whole workgroup (the bin's first two
// variables): read once per bin, kept in scalar registers.
template <typename T, int O>
__device__ __forceinline__ void span_tab_uniform(const T *tab, int nk, int ix, SpanTab<T, O> &t)
{
#pragma unroll
    for (int j = 0; j < O - 1; ++j) t.kn[j] = cs_uniform(tab[ix - (O - 1) + j]);
#pragma unroll
    for (int D = 1; D < O; ++D)
#pragma unroll
        for (int j = 0; j < D; ++j) t.rc[D][j] = cs_uniform(tab[D * nk + ix - D + j]);
}

// All table reads of a lane's own span in flight together (explicit reads: hipcc otherwise fetches every level of
// the recursion right before it and waits for each); span_tab_wait completes them.
template <typename T, int O>
__device__ __forceinline__ void span_tab_issue(unsigned tab_addr, int nk, int ix, SpanTab<T, O> &t)
{
    if constexpr (O > 1) {
        lds_issue_n<T, O - 1, (O > 1 ? O - 1 : 1)>(tab_addr + (unsigned)(ix - (O - 1)) * (unsigned)sizeof(T), t.kn);
        basis_issue<T, O, 1>(tab_addr, nk, ix, t.rc);
    }
}
template <typename T, int O>
__device__ __forceinline__ void span_tab_wait(SpanTab<T, O> &t)
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int j = 0; j < O - 1; ++j) asm volatile("" : "+v"(t.kn[j]) :: "memory");
#pragma unroll
    for (int D = 1; D < O; ++D)
#pragma unroll
        for (int j = 0; j < D; ++j) asm volatile("" : "+v"(t.rc[D][j]) :: "memory");
}

// The contraction of eval_cellsort<..., MFMA>: explicit LDS reads, issued CS_AHEAD rows before the instructions that
// use them (left to itself hipcc loads every row into the same registers right before its first use and waits for
// each; __builtin_amdgcn_sched_group_barrier did not change that).  hipcc does not track asm operands: the value of
// a read is tied to the wait that completes it, and check_lds_hazards.py replays the assembly (Makefile).
template <int OFF>
__device__ __forceinline__ cs_f4 cs_lds_b128(unsigned addr)
{
    cs_f4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}

template <int O>
struct CsRow {                                              // the coefficient operands of one window row
    static constexpr int G = O / 4, R = O % 4;
    cs_f4 g[G > 0 ? G : 1];
    float r[R > 0 ? R : 1];
};

template <int O, int ROW>
__device__ __forceinline__ void cs_issue_row(unsigned a_addr, unsigned r_addr, CsRow<O> &row)
{
    constexpr int ROWS = O * O, G = O / 4, R = O % 4;
    static_assert(G <= 1 && ((G * 4 + R) * ROWS) * 16 < 65536, "offsets of the row reads");
    if constexpr (G >= 1) row.g[0] = cs_lds_b128<ROW * 16>(a_addr);
    if constexpr (R >= 1) row.r[0] = LdsRead<float>::template at<(0 * ROWS + ROW) * 16>(r_addr);
    if constexpr (R >= 2) row.r[1] = LdsRead<float>::template at<(1 * ROWS + ROW) * 16>(r_addr);
    if constexpr (R >= 3) row.r[2] = LdsRead<float>::template at<(2 * ROWS + ROW) * 16>(r_addr);
}

// wait until at most CNT younger reads are outstanding; the row's values become available here
template <int O, int CNT>
__device__ __forceinline__ void cs_wait_row(CsRow<O> &row)
{
    constexpr int G = O / 4, R = O % 4;
    static_assert(CNT >= 0 && CNT <= 15, "lgkmcnt is 4 bits");
    if constexpr (G == 1 && R == 0) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(row.g[0]) : "n"(CNT) : "memory");
    else if constexpr (G == 1 && R == 1) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(row.g[0]), "+v"(row.r[0]) : "n"(CNT) : "memory");
    else if constexpr (G == 1 && R == 2)
        asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(row.g[0]), "+v"(row.r[0]), "+v"(row.r[1]) : "n"(CNT) : "memory");
    else if constexpr (G == 0 && R == 1) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(row.r[0]) : "n"(CNT) : "memory");
    else if constexpr (G == 0 && R == 2) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(row.r[0]), "+v"(row.r[1]) : "n"(CNT) : "memory");
    else asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(row.r[0]), "+v"(row.r[1]), "+v"(row.r[2]) : "n"(CNT) : "memory");
}

template <int O, int ND, int ROW, int AHEAD>
__device__ __forceinline__ void cs_rows(unsigned a_addr, unsigned r_addr, const float (&b0)[O], const float (&b1)[O], float wgt,
                                        CsRow<O> (&buf)[AHEAD + 1], cs_f4 (&accg)[(O / 4) > 0 ? (O / 4) * ND : 1],
                                        cs_f4 (&accr)[(O % 4) > 0 ? (O % 4) : 1])
{
    constexpr int ROWS = O * O, G = O / 4, R = O % 4, NS = AHEAD + 1;
    if constexpr (ROW < ROWS) {
        if constexpr (ROW + AHEAD < ROWS) cs_issue_row<O, ROW + AHEAD>(a_addr, r_addr, buf[(ROW + AHEAD) % NS]);
        // the next row's weight is formed a row early (pinned above this row's wait): a product right in front of the
        // MFMA that reads it costs wait states
        float wnext = 0.f;
        if constexpr (ROW + 1 < ROWS) {
            wnext = b0[(ROW + 1) / O] * b1[(ROW + 1) % O];
            asm volatile("" : "+v"(wnext) :: "memory");
        }
        constexpr int later = ROWS - 1 - ROW < AHEAD ? ROWS - 1 - ROW : AHEAD;
        CsRow<O> &row = buf[ROW % NS];
        cs_wait_row<O, (G + R) * later>(row);
        if constexpr (G >= 1) {
#pragma unroll
            for (int dd = 0; dd < ND; ++dd) accg[dd] = __builtin_amdgcn_mfma_f32_4x4x1f32(row.g[0][dd], wgt, accg[dd], 0, 0, 0);
        }
#pragma unroll
        for (int r2 = 0; r2 < R; ++r2) accr[r2] = __builtin_amdgcn_mfma_f32_4x4x1f32(row.r[r2], wgt, accr[r2], 0, 0, 0);
        cs_rows<O, ND, ROW + 1, AHEAD>(a_addr, r_addr, b0, b1, wnext, buf, accg, accr);
    }
}

template <int O, int ND, int AHEAD = CS_AHEAD>
__device__ __forceinline__ void cs_contract(unsigned a_addr, unsigned r_addr, const float (&b0)[O], const float (&b1)[O],
                                            cs_f4 (&accg)[(O / 4) > 0 ? (O / 4) * ND : 1], cs_f4 (&accr)[(O % 4) > 0 ? (O % 4) : 1])
{
    CsRow<O> buf[AHEAD + 1];
    constexpr int ROWS = O * O;
    if constexpr (ROWS > 0 && AHEAD > 0) cs_issue_row<O, 0>(a_addr, r_addr, buf[0]);
    if constexpr (ROWS > 1 && AHEAD > 1) cs_issue_row<O, 1>(a_addr, r_addr, buf[1]);
    if constexpr (ROWS > 2 && AHEAD > 2) cs_issue_row<O, 2>(a_addr, r_addr, buf[2]);
    if constexpr (ROWS > 3 && AHEAD > 3) cs_issue_row<O, 3>(a_addr, r_addr, buf[3]);
    static_assert(AHEAD <= 4, "prologue");
    cs_rows<O, ND, 0, AHEAD>(a_addr, r_addr, b0, b1, b0[0] * b1[0], buf, accg, accr);
}

// value and first-derivative bases of one variable from ONE recursion: the levels below the last are common, the last
// level runs in its value and in its derivative form (the operations of basis_regs<.., true> with wrt 0 and wrt 1, in
// the same order: same bits)
template <typename T, int O>
__device__ __forceinline__ void basis_regs_vd(const SpanTab<T, O> &t, T u, T (&b)[O], T (&db)[O])
{
#pragma unroll
    for (int k = 0; k < O; ++k) b[k] = T(0);
    b[O - 1] = T(1);
#pragma unroll
    for (int degree = 1; degree < O - 1; ++degree) {
#pragma unroll
        for (int j = 0; j < degree; ++j) {
            const int bi = O - degree + j;
            const T alpha = (u - t.kn[(O - 1) - degree + j]) * t.rc[degree][j];
            b[bi - 1] += (T(1) - alpha) * b[bi];
            b[bi] *= alpha;
        }
    }
#pragma unroll
    for (int k = 0; k < O; ++k) db[k] = O > 1 ? b[k] : T(0);     // (order 1: the derivative of a piecewise constant is zero)
    if constexpr (O > 1) {
        constexpr int degree = O - 1;
#pragma unroll
        for (int j = 0; j < degree; ++j) {
            const int bi = O - degree + j;
            const T alpha = (u - t.kn[(O - 1) - degree + j]) * t.rc[degree][j];
            b[bi - 1] += (T(1) - alpha) * b[bi];
            b[bi] *= alpha;
            const T beta = T(degree) * t.rc[degree][j];
            db[bi - 1] -= beta * db[bi];
            db[bi] *= beta;
        }
    }
}

// The fused jacobian's contraction: every row of the window is read ONCE and feeds three sets of accumulators - the
// row weights db0_i b1_j, b0_i db1_j and b0_i b1_j (the third set is combined with db2 at the end).  One row in flight
// ahead of the one in use.
template <int O, int ND, int ROW>
__device__ __forceinline__ void cs_rows3(unsigned a_addr, unsigned r_addr, const float (&b0)[O], const float (&d0)[O],
                                         const float (&b1)[O], const float (&d1)[O], CsRow<O> (&buf)[2],
                                         cs_f4 (&acc)[3][(O / 4) * ND + (O % 4) > 0 ? (O / 4) * ND + (O % 4) : 1])
{
    constexpr int ROWS = O * O, G = O / 4, R = O % 4;
    if constexpr (ROW < ROWS) {
        if constexpr (ROW + 1 < ROWS) cs_issue_row<O, ROW + 1>(a_addr, r_addr, buf[(ROW + 1) % 2]);
        constexpr int i = ROW / O, j = ROW % O;
        float w[3] = {d0[i] * b1[j], b0[i] * d1[j], b0[i] * b1[j]};
        asm volatile("" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]) :: "memory");       // formed above the row's wait
        CsRow<O> &row = buf[ROW % 2];
        cs_wait_row<O, (ROW + 1 < ROWS) ? (G + R) : 0>(row);
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            if constexpr (G >= 1) {
#pragma unroll
                for (int dd = 0; dd < ND; ++dd) acc[s][dd] = __builtin_amdgcn_mfma_f32_4x4x1f32(row.g[0][dd], w[s], acc[s][dd], 0, 0, 0);
            }
#pragma unroll
            for (int r2 = 0; r2 < R; ++r2)
                acc[s][G * ND + r2] = __builtin_amdgcn_mfma_f32_4x4x1f32(row.r[r2], w[s], acc[s][G * ND + r2], 0, 0, 0);
        }
        cs_rows3<O, ND, ROW + 1>(a_addr, r_addr, b0, d0, b1, d1, buf, acc);
    }
}

// lanes of an eval_cellsort workgroup: the fused jacobian keeps three sets of accumulators and both kinds of bases -
// 161 registers, i.e. three waves per SIMD: three workgroups of four waves on a CU
template <bool JAC>
__host__ __device__ constexpr int cs_block() { return JAC ? 256 : CS_BLOCK; }

// JAC: the fused jacobian - one tile sort and one set of record reads for the three partials of a record, stored at
// tmp[j * N + dest].  Value and first-derivative bases of every variable come from one recursion (basis_regs_vd) and
// every window row is read once for three sets of accumulators (cs_rows3): 161 registers, three waves per SIMD.
// (Three passes of the plain body over the sorted tile at 125 registers / four waves per SIMD: 475 us per 10 M cfg5
// points against 445; the same sharing squeezed into 128 registers spilled 32 - 46 of them in three arrangements.)
template <typename T, int O, int ND, bool MFMA, bool DERIV = true, bool JAC = false>
__global__ __launch_bounds__(cs_block<JAC>()) __attribute__((amdgpu_waves_per_eu(JAC ? 3 : MFMA && O <= 5 ? 4 : 1, 8))) void eval_cellsort(const Desc<T> d, const BinPlan bp, const T *__restrict__ gtab,
                                                     const T *__restrict__ aos, const unsigned *__restrict__ start,
                                                     const BinRec<T, 3> *__restrict__ rec, const long long N,
                                                     BinOut<T, ND> *__restrict__ tmp, const Wrt wrt, const int dest_bits)
{
    static_assert(!MFMA || (sizeof(T) == 4 && ND <= 4), "the 4x4x1 fp32 MFMA form");
    static_assert(!JAC || DERIV, "the fused jacobian requests derivatives");
    static_assert(sizeof(BinRec<T, 3>) == 4 * sizeof(T), "record = u, v, w, tag");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int ncl = d.ncoef[2];
    const int S2 = ncl - d.order[2] + 1;
    constexpr int ROWS = O * O;
    constexpr int BLK = cs_block<JAC>();
    constexpr int PER = cs_per<T>();                        // records per lane and tile
    constexpr int TILE_CAP = PER * BLK;
    constexpr int DP = MFMA ? 4 : ND;                       // dependent-variable slots of a bundle row
    const size_t tab_b = (sizeof(T) * (size_t)d.tab_len + 15) & ~(size_t)15;
    const size_t bun_b = (sizeof(T) * (size_t)ROWS * DP * ncl + 15) & ~(size_t)15;
    T *stab = reinterpret_cast<T *>(smem);
    T *bun = reinterpret_cast<T *>(smem + tab_b);
    BinRec<T, 3> *srec = reinterpret_cast<BinRec<T, 3> *>(smem + tab_b + bun_b);
    unsigned *hist = reinterpret_cast<unsigned *>(smem + tab_b + bun_b + sizeof(BinRec<T, 3>) * (size_t)(TILE_CAP + 4 * S2));
    unsigned *segs = hist + 2 * CS_MAX_S2;                  // segment start of every span
    __shared__ int s_first[BLK / 64];
    int tile_no = 0;
    // Records per tile: with the padding of its S2 segments (0 .. 3 lanes each, 1.5 on average) a tile should fill the
    // wave passes of the workgroup (4 per wave) and not start one more that one wave runs while the others wait.
    const int tile = TILE_CAP - 2 * S2 - 8 >= TILE_CAP / 2 ? TILE_CAP - 2 * S2 - 8 : TILE_CAP / 2;
    for (int i = threadIdx.x; i < d.tab_len; i += blockDim.x) stab[i] = gtab[i];
    for (int i = threadIdx.x; i < 2 * CS_MAX_S2; i += blockDim.x) hist[i] = 0u;
    const long long per = (N + gridDim.x - 1) / gridDim.x;
    const long long lo = (long long)blockIdx.x * per, hi = lo + per < N ? lo + per : N;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    {
        // first bin of the range: the last c with start[c] <= lo (start is non-decreasing).  Every lane probes its share in
        // one round of independent loads (a binary search by one lane is a chain of a dozen dependent misses, ~10 us in
        // front of every workgroup's first tile)
        int best = 0;
        for (int c = threadIdx.x; c < bp.cells; c += blockDim.x)
            if ((long long)start[c] <= lo) best = c;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) best = max(best, __shfl_xor(best, off));
        if (lane == 0) s_first[wave] = best;
    }
    __syncthreads();
    int first = 0;
#pragma unroll
    for (int i = 0; i < BLK / 64; ++i) first = max(first, s_first[i]);
    const T *tab2 = stab + d.off[2];
    const int cs0 = d.ncoef[1] * d.ncoef[2], cs1 = d.ncoef[2];   // table strides (control points)
    for (int c = first; c < bp.cells && lo < hi; ++c) {
        const long long cb = start[c], ce = c + 1 < bp.cells ? (long long)start[c + 1] : N;
        if (cb >= hi) break;
        const long long sl = cb > lo ? cb : lo, sh = ce < hi ? ce : hi;
        if (sl >= sh) continue;
        const int q0 = c / bp.n1, q1 = c % bp.n1;          // window starts of the first two variables (bins are exact spans)
        SpanTab<T, O> st0, st1;                              // their table values: the same for every point of the bin
        span_tab_uniform<T, O>(stab + d.off[0], d.nk[0], q0 + O, st0);
        span_tab_uniform<T, O>(stab + d.off[1], d.nk[1], q1 + O, st1);
        __syncthreads();                                     // previous bin's readers are done
        // bundle: row r = (i, j) of the bin's O x O control-point rows along the third variable
        if constexpr (MFMA && ND == 4) {
            // flat over the bundle's 16-byte elements, two independent loads per lane in flight (a wave per row is a
            // chain of ROWS / 8 dependent round trips in front of the bin's first tile)
            const int tot4 = ROWS * ncl;
            for (int e0 = threadIdx.x; e0 < tot4; e0 += 2 * BLK) {
                float4 v[2];
                int at[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int e = e0 + u * BLK, ee = e < tot4 ? e : tot4 - 1;
                    const int r = ee / ncl, k = ee - r * ncl, i = r / O, j = r - i * O;
                    v[u] = *reinterpret_cast<const float4 *>(aos + ((long long)(q0 + i) * cs0 + (long long)(q1 + j) * cs1 + k) * 4);
                    at[u] = e < tot4 ? (k * ROWS + r) * 4 : -1;
                }
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    if (at[u] >= 0) *reinterpret_cast<float4 *>(bun + at[u]) = v[u];
            }
        } else
        for (int r = wave; r < ROWS; r += BLK / 64) {
            const int i = r / O, j = r - i * O;
            const T *__restrict__ src = aos + ((long long)(q0 + i) * cs0 + (long long)(q1 + j) * cs1) * ND;
            if constexpr (MFMA) {
                // [k][row][4]: the rows of one control-point index k of the third variable side by side, so that a
                // lane's window rows sit at compile-time offsets from ONE address; dependent variables beyond ND zero
                if constexpr (ND == 4) {
                    for (int k = lane; k < ncl; k += 64)
                        *reinterpret_cast<float4 *>(bun + ((size_t)k * ROWS + r) * 4) = *reinterpret_cast<const float4 *>(src + 4 * k);
                } else {
                    for (int e = lane; e < 4 * ncl; e += 64) {
                        const int k = e >> 2, dep = e & 3;
                        bun[((size_t)k * ROWS + r) * 4 + dep] = dep < ND ? src[k * ND + dep] : T(0);
                    }
                }
            } else {
                for (int e = lane; e < ncl * ND; e += 64) bun[(size_t)r * ncl * ND + e] = src[e];
            }
        }
        const unsigned dest_mask = (1u << dest_bits) - 1u;      // record tag = dest | span key << dest_bits (bin_scatter_tag)
        for (long long t0 = sl; t0 < sh; t0 += tile) {
            const int cnt = (int)((sh - t0) < tile ? (sh - t0) : tile);
            // Two barriers per tile.  The histograms alternate: this tile counts in `hc` (zeroed one tile ago), and
            // clears `hn` for the next one after barrier (A).
            unsigned *hc = hist + (tile_no & 1) * CS_MAX_S2, *hn = hist + ((tile_no & 1) ^ 1) * CS_MAX_S2;
            ++tile_no;
            // the tile's records (their lines were touched one tile ago: L2 hits; keeping them in registers across the
            // evaluation of the previous tile costs 16 registers and, with them, the fourth wave per SIMD)
            typedef T cs_rec4 __attribute__((ext_vector_type(4)));
            cs_rec4 rc[PER];
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const long long p = t0 + i * BLK + (long long)threadIdx.x;
                // read once: non-temporal (214 -> 207 us)
                rc[i] = __builtin_nontemporal_load(reinterpret_cast<const cs_rec4 *>(&rec[p < sh ? p : sh - 1]));
            }
            // --- rank inside the span of the third variable
            int key[PER];
            unsigned rank[PER];
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int idx = i * BLK + (int)threadIdx.x;
                key[i] = -1;
                if (idx < cnt) {
                    // found by the scatter kernel.  (Clamped: a tag is DATA from another kernel; no index formed from it
                    // may leave its array - histogram here, bundle and tmp below.)
                    key[i] = min((int)(word_tag<T>((T)rc[i][3]) >> dest_bits), S2 - 1);
                    rank[i] = atomicAdd(&hc[key[i]], 1u);
                }
            }
            __syncthreads();                                 // (A) counts complete; bundle staged; previous tile's readers of srec done
            // --- segment starts: exclusive scan of the counts rounded up to multiples of four.  EVERY wave scans and
            // writes the same values (a wave reads back its own LDS writes in order: no barrier before the look-ups)
            unsigned carry = 0;
            for (int k0 = 0; k0 < S2; k0 += 64) {
                const int k = k0 + lane;
                const unsigned v = k < S2 ? ((hc[k] + 3u) & ~3u) : 0u;
                unsigned inc = v;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const unsigned o = __shfl_up(inc, off);
                    if (lane >= off) inc += o;
                }
                if (k < S2) segs[k] = carry + inc - v;
                carry += __shfl(inc, 63);
            }
            const int total = (int)__builtin_amdgcn_readfirstlane((int)carry);
            for (int k = threadIdx.x; k < S2; k += blockDim.x) hn[k] = 0u;
            // --- records into span order (the tag travels with the record: destination of the result and the span)
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                if (key[i] >= 0) *reinterpret_cast<cs_rec4 *>(&srec[segs[key[i]] + rank[i]]) = rc[i];
            }
            // touch the lines of the next tile's records (one dword per 128-byte line), consumed after the evaluation
            T touch = T(0);
            if (t0 + tile < sh) {
                const long long p = t0 + tile + (long long)threadIdx.x * (128 / (long long)sizeof(BinRec<T, 3>));
                if (threadIdx.x < TILE_CAP * sizeof(BinRec<T, 3>) / 128 && p < sh) touch = rec[p].v[0];
            }
            // padding lanes of a segment: a point inside the same cell, tagged invalid (no store)
            for (int k = threadIdx.x; k < S2; k += blockDim.x) {
                const unsigned have = hc[k], want = (have + 3u) & ~3u;
                if (have != want) {
                    BinRec<T, 3> r;
                    r.v[0] = d.lo[0];
                    r.v[1] = d.lo[1];
                    r.v[2] = tab2[k + d.order[2] - 1];        // left knot of span k: inside the cell
                    r.v[3] = tag_word<T>(dest_mask | ((unsigned)k << dest_bits));
                    for (unsigned p = have; p < want; ++p) srec[segs[k] + p] = r;
                }
            }
            __syncthreads();                                 // (B) tile in span order
            // --- evaluation in span order
            for (int g = wave * 64; g < total; g += BLK) {
                const int q = g + lane;
                const bool live = q < total;
                const BinRec<T, 3> r = srec[live ? q : total - 1];
                const unsigned tag = word_tag<T>(r.v[3]);
                const int ix2 = min((int)(tag >> dest_bits), S2 - 1);
                const unsigned dest = tag & dest_mask;
                // (the fused jacobian has its own body below)
                auto body = [&]() __attribute__((always_inline)) {
                const int w0 = wrt.w[0], w1 = wrt.w[1], w2 = wrt.w[2];
                T b[3][O];
                SpanTab<T, O> st2;
                span_tab_issue<T, O>((unsigned)(size_t)tab2, d.nk[2], ix2 + O, st2);
                basis_regs<T, O, DERIV>(st0, r.v[0], w0, b[0]);
                basis_regs<T, O, DERIV>(st1, r.v[1], w1, b[1]);
                span_tab_wait<T, O>(st2);
                basis_regs<T, O, DERIV>(st2, r.v[2], w2, b[2]);
                T res[ND];
                if constexpr (MFMA) {
                    // The steps m of the THIRD variable go four at a time into the ROWS of the outer product:
                    //   group g, dependent variable dd:  D[m'][point] += C[i][j][4g + m'][dd] * (b0_i b1_j)[point]
                    //   remaining step m = 4G + r:       D[dd][point] += C[i][j][m][dd]      * (b0_i b1_j)[point]
                    // Lane l = 4 b + m' of block b feeds control point ix2 + 4g + m' of its block's cell: ALL its dependent
                    // variables are one aligned 16-byte LDS read (ds_read_b128: 256 B/clk, a 4-byte read 128 B/clk and
                    // a misaligned wide read less - tools/lds_unaligned_probe.hip), used by the ND instructions of the
                    // group; a remaining step reads 4 bytes.  O = 5, ND = 4: 5 MFMAs and 6 LDS cycles per row (was 10);
                    // ND < 4 needs ND instead of 4 instructions per group.  The B operand is the row weight alone (O^2
                    // products per point instead of O^3), the chains are independent, and the lane combines its own
                    // sums with b2 at the end.
                    constexpr int G = O / 4, R = O % 4;
                    cs_f4 accg[G > 0 ? G * ND : 1], accr[R > 0 ? R : 1];
#pragma unroll
                    for (int q2 = 0; q2 < G * ND; ++q2) accg[q2] = cs_f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int q2 = 0; q2 < R; ++q2) accr[q2] = cs_f4{0.f, 0.f, 0.f, 0.f};
                    const unsigned a_addr = (unsigned)(size_t)bun + (unsigned)((ix2 + (lane & 3)) * (ROWS * 16));
                    const unsigned r_addr = (unsigned)(size_t)bun + (unsigned)((ix2 + 4 * G) * (ROWS * 16) + (lane & 3) * 4);
                    cs_contract<O, ND, CS_AHEAD>(a_addr, r_addr, b[0], b[1], accg, accr);
#pragma unroll
                    for (int dd = 0; dd < ND; ++dd) {
                        T sum = T(0);
#pragma unroll
                        for (int g2 = 0; g2 < G; ++g2)
#pragma unroll
                            for (int m = 0; m < 4; ++m) sum += b[2][4 * g2 + m] * accg[g2 * ND + dd][m];
#pragma unroll
                        for (int r2 = 0; r2 < R; ++r2) sum += b[2][4 * G + r2] * accr[r2][dd];
                        res[dd] = sum;
                    }
                } else {
                    int pad[3] = {0, 0, 0};
                    window_contract<T, 3, O, ND, false>(bun + (long long)ix2 * ND, O * ncl, ncl, pad, b, res);
                }
                if (live && (long long)dest < N) {             // padding lanes carry dest_mask >= N; nothing else can pass either
                    BinOut<T, ND> o;
#pragma unroll
                    for (int k = 0; k < BinOut<T, ND>::WORDS; ++k) o.v[k] = k < ND ? res[k < ND ? k : 0] : T(0);
                    tmp[dest] = o;   // chunk order (bin_unpermute_stream); (un-sorting the tile's results through LDS for whole-line stores: 297 -> 394 us)
                }
                };
                if constexpr (JAC && MFMA) {
                    // fused jacobian: value and first-derivative bases of the three variables (three recursions with a
                    // doubled last level), every window row read once for three sets of accumulators (cs_rows3)
                    T bv[3][O], bd[3][O];
                    basis_regs_vd<T, O>(st0, r.v[0], bv[0], bd[0]);
                    basis_regs_vd<T, O>(st1, r.v[1], bv[1], bd[1]);
                    constexpr int G = O / 4, R = O % 4, NA = G * ND + R > 0 ? G * ND + R : 1;
                    cs_f4 acc[3][NA];
#pragma unroll
                    for (int s3 = 0; s3 < 3; ++s3)
#pragma unroll
                        for (int q2 = 0; q2 < NA; ++q2) acc[s3][q2] = cs_f4{0.f, 0.f, 0.f, 0.f};
                    const unsigned a_addr = (unsigned)(size_t)bun + (unsigned)((ix2 + (lane & 3)) * (ROWS * 16));
                    const unsigned r_addr = (unsigned)(size_t)bun + (unsigned)((ix2 + 4 * G) * (ROWS * 16) + (lane & 3) * 4);
                    CsRow<O> buf[2];
                    cs_issue_row<O, 0>(a_addr, r_addr, buf[0]);
                    cs_rows3<O, ND, 0>(a_addr, r_addr, bv[0], bd[0], bv[1], bd[1], buf, acc);
                    {   // the third variable's bases after the contraction: ten registers fewer while 15 accumulators are live
                        SpanTab<T, O> st2;
                        span_tab_issue<T, O>((unsigned)(size_t)tab2, d.nk[2], ix2 + O, st2);
                        span_tab_wait<T, O>(st2);
                        basis_regs_vd<T, O>(st2, r.v[2], bv[2], bd[2]);
                    }
#pragma unroll
                    for (int s3 = 0; s3 < 3; ++s3) {
                        const T (&b2)[O] = s3 == 2 ? bd[2] : bv[2];
                        BinOut<T, ND> o;
#pragma unroll
                        for (int k = 0; k < BinOut<T, ND>::WORDS; ++k) o.v[k] = T(0);
#pragma unroll
                        for (int dd = 0; dd < ND; ++dd) {
                            T sum = T(0);
#pragma unroll
                            for (int g2 = 0; g2 < G; ++g2)
#pragma unroll
                                for (int m = 0; m < 4; ++m) sum += b2[4 * g2 + m] * acc[s3][g2 * ND + dd][m];
#pragma unroll
                            for (int r2 = 0; r2 < R; ++r2) sum += b2[4 * G + r2] * acc[s3][G * ND + r2][dd];
                            o.v[dd] = sum;
                        }
                        if (live && (long long)dest < N) tmp[(long long)s3 * N + dest] = o;
                    }
                } else {
                    body();
                }
            }
            asm volatile("" :: "v"(touch));
        }
    }
}

// out[dep][n] = tmp[slot[n]][dep]; four independent gathers per lane and iteration.
template <typename T, int ND>
__global__ __launch_bounds__(256) void bin_unpermute(const long long N, const unsigned *__restrict__ slot,
                                                     const BinOut<T, ND> *__restrict__ tmp, T *__restrict__ out,
                                                     const long long ostride)
{
    const long long stride = (long long)gridDim.x * blockDim.x * BIN_ILP;
    for (long long n0 = (long long)blockIdx.x * blockDim.x * BIN_ILP + threadIdx.x; n0 < N; n0 += stride) {
        unsigned p[BIN_ILP];
#pragma unroll
        for (int k = 0; k < BIN_ILP; ++k) {
            const long long n = n0 + (long long)k * blockDim.x;
            p[k] = slot[n < N ? n : N - 1];
        }
        BinOut<T, ND> r[BIN_ILP];
#pragma unroll
        for (int k = 0; k < BIN_ILP; ++k) r[k] = tmp[p[k]];
#pragma unroll
        for (int k = 0; k < BIN_ILP; ++k) {
            const long long n = n0 + (long long)k * blockDim.x;
            if (n < N) {
#pragma unroll
                for (int dd = 0; dd < ND; ++dd) nt_store(&out[dd * ostride + n], r[k].v[dd]);
            }
        }
    }
}

}  // namespace bsk
