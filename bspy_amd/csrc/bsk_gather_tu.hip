// Large-table evaluation paths of libbspy_amd.so in their own translation unit (two thirds of the
// device code; compiled in parallel with bsk_api.hip): eval_gather (control-point-major gather
// from L2) and the cell-order pipeline (bsk_binned.hpp).  Entry: gather_or_binned_any<T>.
#include "bsk_host.hpp"
#include "bsk_tile.hpp"
#include "bsk_gather.hpp"
#include "bsk_binned.hpp"

template <typename T, int NIND, int O, bool MIXED>
static bsk_status launch_eval_gather(bsk_spline s, const Params<T> &prm, long long n, T *out, long long ostride,
                                     const Wrt &w, hipStream_t st)
{
    const Desc<T> &d = desc_of<T>(s);
    const size_t lds = sizeof(T) * (size_t)d.tab_len;
    if (lds > s->lds_max / 2) return BSK_ERR_UNSUPPORTED;        // axis tables too large for LDS: the caller falls back
    const int block = 256;
    const long long blocks = (n + block - 1) / block;
    const int grid = (int)std::max<long long>(1, std::min<long long>(blocks, (long long)s->num_cu * 8));
    const T *tab = static_cast<const T *>(s->tab);
    const T *aos = static_cast<const T *>(s->coef_aos);
#define GATHER_ND(ND)                                                                                             \
    case ND:                                                                                                      \
        HIPCHK(allow_lds(eval_gather<T, NIND, O, ND, MIXED>, lds));                                               \
        s->last_kernel = "eval_gather";                                                                           \
        hipLaunchKernelGGL((eval_gather<T, NIND, O, ND, MIXED>), dim3(grid), dim3(block), lds, st, d, tab, aos, prm, n,  \
                           out, ostride, w, s->bad);                                                              \
        break;
    switch (s->nDep) {
        GATHER_ND(1) GATHER_ND(2) GATHER_ND(3) GATHER_ND(4)
        default: return fail(BSK_ERR_INVALID, "internal: eval_gather needs nDep <= 4");
    }
#undef GATHER_ND
    HIPCHK(hipGetLastError());
    return BSK_OK;
}

// Cell-order evaluation of large batches on L2-resident tables (bsk_binned.hpp).  Returns
// BSK_ERR_UNSUPPORTED when it does not apply (the caller then gathers in batch order).
constexpr long long BIN_MIN_POINTS = 1 << 18;
#ifndef BIN_CHUNK_POINTS
#define BIN_CHUNK_POINTS 8192
#endif

// Three variables of one order on an L2-resident table (BASELINE cfg5): the round-3 pipeline of bsk_binned.hpp,
//   bin_totals -> bin_scatter_tag -> eval_cellsort -> bin_unpermute_stream
// (4 launches; round 2: 6).  A record's tag holds its destination in chunk order and its span key in 32 bits, so a
// launch takes at most 2^26 - 1 points (span keys < 64) or 2^24 - 1 (< 256): larger batches run piece by piece.
template <typename T, int O>
static bsk_status launch_cellsort_pipeline(bsk_spline s, BinPlan bp, const Params<T> &prm, long long n, T *out, long long ostride,
                                           const Wrt &w, hipStream_t st, size_t cs_lds_mfma, size_t cs_lds_valu, bool jac = false)
{
    // jac: the fused jacobian (eval_cellsort<..., JAC>): three result arrays in the workspace, three streaming un-permutes
    // into out[(dep * 3 + j) * n + i] (the caller passes ostride = 3 n); MFMA form only
    if (jac && (sizeof(T) != 4 || s->variant == 12)) return BSK_ERR_UNSUPPORTED;
    const Desc<T> &d = desc_of<T>(s);
    const TileDesc<T> &td = tile_of<T>(s);
    const int S2 = s->ncoef[2] - s->order[2] + 1;
    const int dest_bits = S2 <= 64 ? 26 : 24;
    // chunk of the scatter / un-permute: records (+ a 16-bit bin each) beside the bin tables and the span tables
    const size_t tabs_b = (span_lds_bytes<T, 3>(d, td) + 15) & ~(size_t)15;
    const size_t bins_b = (10 * (size_t)bp.cells + 15) & ~(size_t)15;
    if (bins_b + tabs_b + 1024 > s->lds_max || bp.cells > BIN_MAX_CELLS) return BSK_ERR_UNSUPPORTED;
    const size_t out_sz = (4 * sizeof(T) + 15) / 16 * 16;          // BinOut<T, ND <= 4>
    long long chunk_max = (long long)((s->lds_max - bins_b - tabs_b - 256) / (sizeof(BinRec<T, 3>) + 2)) / 1024 * 1024;
    chunk_max = std::min<long long>(chunk_max, (long long)(s->lds_max / out_sz) / 1024 * 1024);
    chunk_max = std::min<long long>(chunk_max, 1024 * wc_ppt<T>());
    static const int env_wc = getenv("BSK_WC_CHUNK") ? atoi(getenv("BSK_WC_CHUNK")) : 0;                  // measurement knobs
    static const int env_cs = getenv("BSK_CS_GRID") ? atoi(getenv("BSK_CS_GRID")) : 0;
    if (env_wc > 0) chunk_max = std::min<long long>(chunk_max, std::max(1024, env_wc / 1024 * 1024));
    if (chunk_max < 2048) return BSK_ERR_UNSUPPORTED;
    long long piece_max = (((1ll << dest_bits) - 1) / 1024) * 1024;
    const char *env_piece_s = getenv("BSK_CS_PIECE");                                                     // test knob: short pieces (read per call)
    const long long env_piece = env_piece_s ? atoll(env_piece_s) : 0;
    if (env_piece > 0) piece_max = std::min(piece_max, std::max(8192ll, env_piece / 1024 * 1024));
    const long long pieces = (n + piece_max - 1) / piece_max;
    const long long piece = std::min(n, (((n + pieces - 1) / pieces) + 1023) / 1024 * 1024);
    const size_t lds_tot = ((span_lds_bytes<T, 2>(d, td) + 15) & ~(size_t)15) + 4 * (size_t)bp.cells;
    if (lds_tot > s->lds_max / 2) return BSK_ERR_UNSUPPORTED;

    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t o_lpos = 0, o_rec = o_lpos + up(2 * (size_t)piece), o_tmp = o_rec + up(sizeof(BinRec<T, 3>) * (size_t)piece);
    const size_t o_tot = o_tmp + up(out_sz * (size_t)piece * (jac ? 3 : 1)), o_start = o_tot + up(4 * (size_t)(bp.cells + 1));
    const size_t o_rows = o_start + up(4 * (size_t)bp.cells), total = o_rows + up(4 * (size_t)bp.cells * s->num_cu);
    HIPCHK(s->bin_ws.reserve(total));
    char *ws = static_cast<char *>(s->bin_ws.p);
    unsigned short *lpos = reinterpret_cast<unsigned short *>(ws + o_lpos);
    BinRec<T, 3> *rec = reinterpret_cast<BinRec<T, 3> *>(ws + o_rec);
    unsigned *tot = reinterpret_cast<unsigned *>(ws + o_tot);
    if (!s->ticket) {                                             // the tickets start at zero and every user leaves them at zero
        HIPCHK(hipMalloc(reinterpret_cast<void **>(&s->ticket), 256));
        HIPCHK(hipMemset(s->ticket, 0, 256));
    }
    unsigned *done = s->ticket;
    unsigned *start = reinterpret_cast<unsigned *>(ws + o_start), *rows = reinterpret_cast<unsigned *>(ws + o_rows);
    const T *tab = static_cast<const T *>(s->tab);
    const T *aos = static_cast<const T *>(s->coef_aos);
    const bool reuse = s->bin_reuse && pieces == 1;               // later derivative passes of a jacobian find the batch sorted
    bool deriv = false;
    for (int iv = 0; iv < 3; ++iv) deriv |= w.w[iv] != 0;
    constexpr bool MF = sizeof(T) == 4;
    const bool mfma = MF && s->variant != 12;
    s->last_kernel = jac ? "cell-order pipeline (eval_cellsort, MFMA, fused jacobian)"
                         : mfma ? "cell-order pipeline (eval_cellsort, MFMA)" : "cell-order pipeline (eval_cellsort, VALU)";

    for (long long off = 0; off < n; off += piece) {
        const long long np = std::min(piece, n - off);
        Params<T> pp = prm;
        for (int iv = 0; iv < 3; ++iv) pp.p[iv] = prm.p[iv] + off;
        bp.chunks = (int)((np + chunk_max - 1) / chunk_max);
        bp.chunk = ((np + bp.chunks - 1) / bp.chunks + 7) & ~7ll;        // multiples of 8 points: chunks keep the batch's 16-byte alignment
        bp.ranges = bp.rlen = 1;
        const int pgrid = std::min(bp.chunks, s->num_cu);
        stage_mark(s, st, "start", true);
        if (!reuse) {
            HIPCHK(allow_lds(bin_totals<T>, lds_tot));
            hipLaunchKernelGGL((bin_totals<T>), dim3(pgrid), dim3(1024), lds_tot, st, d, td, bp, tab, s->lut, pp, np, rows);
            stage_mark(s, st, "bin_totals");
            hipLaunchKernelGGL(bin_starts, dim3((bp.cells + 63) / 64), dim3(1024), 0, st, bp.cells, pgrid, rows, tot, start, done);
            stage_mark(s, st, "bin_starts");
            const size_t lds_s = bins_b + (((size_t)bp.chunk * (sizeof(BinRec<T, 3>) + 2) + 15) & ~(size_t)15) + tabs_b;
            HIPCHK(allow_lds(bin_scatter_tag<T>, lds_s));
            hipLaunchKernelGGL((bin_scatter_tag<T>), dim3(pgrid), dim3(1024), lds_s, st, bp, pp, np, off, rows, start, rec, lpos, d, td, tab,
                               s->lut, dest_bits, s->bad);
            stage_mark(s, st, "bin_scatter_tag");
        }
        const size_t cs_lds_jac = cs_lds_mfma - 4 * sizeof(T) * (size_t)cs_per<T>() * (CS_BLOCK - cs_block<true>());   // shorter tiles
        const int cgrid = (int)std::max<long long>(1, std::min<long long>((np + 4 * CS_BLOCK - 1) / (4 * CS_BLOCK), (long long)s->num_cu * (env_cs > 0 ? env_cs : jac ? 6 : 4)));   // (fused jacobian: three workgroups resident per CU)
#define CS_ND(ND)                                                                                                        \
    case ND: {                                                                                                           \
        BinOut<T, ND> *tmp = reinterpret_cast<BinOut<T, ND> *>(ws + o_tmp);                                              \
        if constexpr (MF) if (jac) {                                                                                     \
            HIPCHK(allow_lds(eval_cellsort<T, O, ND, MF, true, MF>, cs_lds_jac));                                        \
            hipLaunchKernelGGL((eval_cellsort<T, O, ND, MF, true, MF>), dim3(cgrid), dim3(cs_block<true>()), cs_lds_jac, st, d,  \
                               bp, tab, aos, start, rec, np, tmp, w, dest_bits);                                         \
        }                                                                                                                \
        if (jac) {                                                                                                       \
        } else if (mfma) {                                                                                               \
            if (deriv) {                                                                                                 \
                HIPCHK(allow_lds(eval_cellsort<T, O, ND, MF, true>, cs_lds_mfma));                                       \
                hipLaunchKernelGGL((eval_cellsort<T, O, ND, MF, true>), dim3(cgrid), dim3(CS_BLOCK), cs_lds_mfma, st, d, \
                                   bp, tab, aos, start, rec, np, tmp, w, dest_bits);                                     \
            } else {                /* plain evaluation: the recursion without its derivative branches */              \
                HIPCHK(allow_lds(eval_cellsort<T, O, ND, MF, !MF>, cs_lds_mfma));                                        \
                hipLaunchKernelGGL((eval_cellsort<T, O, ND, MF, !MF>), dim3(cgrid), dim3(CS_BLOCK), cs_lds_mfma, st, d,  \
                                   bp, tab, aos, start, rec, np, tmp, w, dest_bits);                                     \
            }                                                                                                            \
        } else {                                                                                                         \
            HIPCHK(allow_lds(eval_cellsort<T, O, ND, false>, cs_lds_valu));                                              \
            hipLaunchKernelGGL((eval_cellsort<T, O, ND, false>), dim3(cgrid), dim3(CS_BLOCK), cs_lds_valu, st, d, bp,    \
                               tab, aos, start, rec, np, tmp, w, dest_bits);                                             \
        }                                                                                                                \
        stage_mark(s, st, "eval_cellsort");                                                                              \
        const size_t lds_u = (size_t)bp.chunk * sizeof(BinOut<T, ND>);                                                   \
        HIPCHK(allow_lds(bin_unpermute_stream<T, ND>, lds_u));                                                           \
        for (int j = 0; j < (jac ? 3 : 1); ++j)                                                                          \
            hipLaunchKernelGGL((bin_unpermute_stream<T, ND>), dim3(pgrid), dim3(1024), lds_u, st, bp, np, lpos, tmp + (size_t)j * np, \
                               out + (long long)j * n + off, ostride);                                                   \
        stage_mark(s, st, "bin_unpermute_stream");                                                                       \
    } break;
        switch (s->nDep) {
            CS_ND(1) CS_ND(2) CS_ND(3) CS_ND(4)
            default: return BSK_ERR_UNSUPPORTED;
        }
#undef CS_ND
        HIPCHK(hipGetLastError());
    }
    return BSK_OK;
}

template <typename T, int NIND, int O, bool MIXED>
static bsk_status launch_eval_binned(bsk_spline s, const Params<T> &prm, long long n, T *out, long long ostride,
                                     const Wrt &w, hipStream_t st, bool jac = false)
{
    if constexpr (NIND < 2) {
        return BSK_ERR_UNSUPPORTED;
    } else {
        if (n < BIN_MIN_POINTS || n > 0xffffffffll || !s->coef_aos || s->variant == 7 || s->nDep > 4) return BSK_ERR_UNSUPPORTED;
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) {
            (void)hipGetLastError();
            return BSK_ERR_UNSUPPORTED;                       // the workspace may have to be (re)allocated
        }
        const Desc<T> &d = desc_of<T>(s);
        BinPlan bp;
        const int s0 = s->ncoef[0] - s->order[0] + 1, s1 = s->ncoef[1] - s->order[1] + 1;   // spans (O = largest order)
        bp.sh0 = bp.sh1 = 0;
        auto cells_of = [&](int sh0, int sh1) { return (((s0 - 1) >> sh0) + 1) * (((s1 - 1) >> sh1) + 1); };
        while (cells_of(bp.sh0, bp.sh1) > BIN_MAX_CELLS) {
            if ((s0 >> bp.sh0) >= (s1 >> bp.sh1)) ++bp.sh0; else ++bp.sh1;
        }
        bp.n1 = ((s1 - 1) >> bp.sh1) + 1;
        bp.cells = cells_of(bp.sh0, bp.sh1);
        bp.chunks = (int)std::max<long long>(1, std::min<long long>(BIN_MAX_CHUNKS, (n + BIN_CHUNK_POINTS - 1) / BIN_CHUNK_POINTS));
        static const int env_chunks = getenv("BSK_BIN_CHUNKS") ? atoi(getenv("BSK_BIN_CHUNKS")) : 0;      // measurement knobs
        static const int env_block = getenv("BSK_BIN_BLOCK") ? atoi(getenv("BSK_BIN_BLOCK")) : 0;
        static const int env_unp = getenv("BSK_UNP_GRID") ? atoi(getenv("BSK_UNP_GRID")) : 0;
        if (env_chunks > 0) bp.chunks = (int)std::min<long long>(std::min(env_chunks, BIN_MAX_CHUNKS), std::max<long long>(1, n / 64));
        // write-combining scatter / un-permute (bsk_binned.hpp): the chunk's records and results must fit LDS
        // beside three (two) bin tables; BSK_VARIANT 14 keeps the direct forms
        const size_t rec_sz = sizeof(BinRec<T, NIND>), out_sz_max = (4 * sizeof(T) + 15) / 16 * 16;
        long long wc_chunk = 0;
        if (bp.cells <= BIN_MAX_WC_CELLS && s->variant != 14 && s->variant != 13) {
            const size_t room = s->lds_max - 12 * (size_t)bp.cells - 256;
            wc_chunk = (long long)(room / (std::max(rec_sz, out_sz_max) + 2)) / 1024 * 1024;
            wc_chunk = std::min<long long>(wc_chunk, 1024 * WC_PPT);
            static const int env_wc = getenv("BSK_WC_CHUNK") ? atoi(getenv("BSK_WC_CHUNK")) : 0;          // measurement knob
            if (env_wc > 0) wc_chunk = std::min<long long>(wc_chunk, env_wc);
            if (wc_chunk < 2048 || (n + wc_chunk - 1) / wc_chunk > 65535) wc_chunk = 0;
        }
        const bool wc = wc_chunk > 0;
        // (a whole number of rounds of the persistent kernels - 1280 slightly smaller chunks on 256 CUs instead of 1221 -
        // measured slower: 83 -> 88 and 93 -> 96 us; the shorter runs cost more than the even rounds win)
        if (wc) bp.chunks = (int)((n + wc_chunk - 1) / wc_chunk);
        const int bin_block = env_block > 0 ? env_block : BIN_BLOCK;
        const int ugrid = env_unp > 0 ? env_unp : 2048;

        bp.chunk = (n + bp.chunks - 1) / bp.chunks;
        bp.ranges = std::min(bp.chunks, BIN_MAX_RANGES);
        bp.rlen = (bp.chunks + bp.ranges - 1) / bp.ranges;
        bp.ranges = (bp.chunks + bp.rlen - 1) / bp.rlen;
        const size_t tab_b = (sizeof(T) * (size_t)d.tab_len + 15) & ~(size_t)15;
        const TileDesc<T> &td = tile_of<T>(s);
        const size_t lds_count = tab_b + sizeof(unsigned) * (size_t)(bp.cells + ((td.lut_len + 3) & ~3));
        if (lds_count > s->lds_max / 2) return BSK_ERR_UNSUPPORTED;

        auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
        const size_t o_cell = 0, o_slot = o_cell + up(2 * (size_t)n), o_rec = o_slot + up(4 * (size_t)n);
        const size_t o_tmp = o_rec + up(sizeof(BinRec<T, NIND>) * (size_t)n);
        size_t o_M = 0, o_tot = 0, o_Tr = 0, o_start = 0, o_pbin = 0, o_Lb = 0, total = 0;
        auto layout = [&](size_t out_bytes) {
            o_M = o_tmp + up(out_bytes * (size_t)n);
            o_tot = o_M + up(4 * (size_t)bp.cells * bp.chunks);
            o_Tr = o_tot + up(4 * (size_t)bp.cells);
            o_start = o_Tr + up(4 * (size_t)bp.cells * bp.ranges);
            o_pbin = o_start + up(4 * (size_t)bp.cells);
            o_Lb = o_pbin + (wc ? up(2 * (size_t)n) : 0);
            total = o_Lb + (wc ? up(4 * (size_t)bp.cells * bp.chunks) : 0);
        };
        const T *tab = static_cast<const T *>(s->tab);
        const T *aos = static_cast<const T *>(s->coef_aos);
        const int egrid = (int)std::max<long long>(1, std::min<long long>((n + 255) / 256, (long long)s->num_cu * 8));
        // rows of one cell staged in LDS (eval_binned_lds) when they fit beside the axis tables twice per CU
        const size_t rows_b = (size_t)((1 << bp.sh0) + s->order[0] - 1) * (NIND == 3 ? (size_t)((1 << bp.sh1) + s->order[1] - 1) : 1);
        const size_t bundle_b = ((rows_b * (size_t)s->ncoef[NIND - 1] * s->nDep * sizeof(T)) + 15) & ~(size_t)15;
        if (tab_b + bundle_b > s->lds_max / 2) return BSK_ERR_UNSUPPORTED;   // rows of a cell must fit LDS twice per CU
        // three variables of one order, bins = exact (span0, span1): second grouping by span2 inside the
        // evaluation workgroup (eval_cellsort); BSK_VARIANT 13 keeps eval_binned_lds, 12 the VALU form
        bool cellsort = false;
        size_t cs_lds_mfma = 0, cs_lds_valu = 0;
        if constexpr (NIND == 3 && !MIXED) {
            const int S2 = s->ncoef[2] - s->order[2] + 1;
            const size_t rec_b = 4 * sizeof(T) * (size_t)(cs_per<T>() * CS_BLOCK + 4 * S2) + sizeof(unsigned) * (size_t)(3 * CS_MAX_S2 + 4);
            auto up16 = [](size_t b) { return (b + 15) & ~(size_t)15; };
            cs_lds_mfma = tab_b + up16(sizeof(T) * (size_t)O * O * 4 * s->ncoef[2]) + rec_b;
            cs_lds_valu = tab_b + up16(sizeof(T) * (size_t)O * O * s->nDep * s->ncoef[2]) + rec_b;
            cellsort = bp.sh0 == 0 && bp.sh1 == 0 && S2 <= CS_MAX_S2 && s->variant != 13 && s->variant != 14 &&
                       std::max(cs_lds_mfma, cs_lds_valu) <= s->lds_max / 2;
            if (cellsort) {
                const bsk_status r = launch_cellsort_pipeline<T, O>(s, bp, prm, n, out, ostride, w, st, cs_lds_mfma, cs_lds_valu, jac);
                if (r != BSK_ERR_UNSUPPORTED) return r;
                cellsort = false;                                 // falls back to the round-2 sort + eval_binned_lds
            }
        }
        if (jac) return BSK_ERR_UNSUPPORTED;                      // the fused jacobian exists on the eval_cellsort pipeline only
#define BINNED_ND(ND)                                                                                                    \
    case ND: {                                                                                                           \
        layout(sizeof(BinOut<T, ND>));                                                                                   \
        HIPCHK(s->bin_ws.reserve(total));                                                                                \
        char *ws = static_cast<char *>(s->bin_ws.p);                                                                     \
        unsigned short *cell = reinterpret_cast<unsigned short *>(ws + o_cell);                                          \
        unsigned *slot = reinterpret_cast<unsigned *>(ws + o_slot);                                                      \
        BinRec<T, NIND> *rec = reinterpret_cast<BinRec<T, NIND> *>(ws + o_rec);                                          \
        BinOut<T, ND> *tmp = reinterpret_cast<BinOut<T, ND> *>(ws + o_tmp);                                              \
        unsigned *M = reinterpret_cast<unsigned *>(ws + o_M);                                                            \
        unsigned *Tr = reinterpret_cast<unsigned *>(ws + o_Tr);                                                          \
        unsigned *start = reinterpret_cast<unsigned *>(ws + o_start);                                                    \
        HIPCHK(allow_lds(bin_count<T, NIND, O>, lds_count));                                                             \
        s->last_kernel = "cell-order pipeline (eval_binned_lds)";                                                        \
        unsigned short *pbin = reinterpret_cast<unsigned short *>(ws + o_pbin);                                          \
        unsigned *Lb = reinterpret_cast<unsigned *>(ws + o_Lb);                                                          \
        if (!s->bin_reuse) {         /* later derivative passes of a jacobian find the batch sorted (dispatch_jac) */    \
        hipLaunchKernelGGL((bin_count<T, NIND, O>), dim3(bp.chunks), dim3(bin_block), lds_count, st, d, td, bp, tab,     \
                           s->lut, prm, n, cell, M, s->bad);                                                             \
        hipLaunchKernelGGL(bin_scan_ranges, dim3((bp.cells + 255) / 256, bp.ranges), dim3(256), 0, st, bp, M, Tr);       \
        hipLaunchKernelGGL(bin_scan_top, dim3(1), dim3(1024), 0, st, bp, Tr, start);                                     \
        if (wc) {                                                                                                        \
            size_t lds_s = ((12 * (size_t)bp.cells + 15) & ~(size_t)15) + (size_t)bp.chunk * (sizeof(BinRec<T, NIND>) + 2); \
            HIPCHK(allow_lds(bin_scatter_wc<T, NIND>, lds_s));                                                           \
            hipLaunchKernelGGL((bin_scatter_wc<T, NIND>), dim3(std::min(bp.chunks, s->num_cu)), dim3(1024), lds_s, st, bp, prm, n, cell, M,   \
                               Tr, start, rec, reinterpret_cast<unsigned short *>(slot), pbin, Lb, d, td, tab, s->lut, s->bad);                        \
        } else                                                                                                           \
        hipLaunchKernelGGL((bin_scatter<T, NIND>), dim3(bp.chunks), dim3(bin_block), sizeof(unsigned) * (size_t)bp.cells, \
                           st, bp, prm, n, cell, M, Tr, start, rec, slot, d, td, tab, s->lut, s->bad);             \
        }                                                                                                                \
        HIPCHK(allow_lds(eval_binned_lds<T, NIND, O, ND, MIXED>, tab_b + bundle_b));                                     \
        hipLaunchKernelGGL((eval_binned_lds<T, NIND, O, ND, MIXED>), dim3(egrid), dim3(256), tab_b + bundle_b, st, d,    \
                           bp, tab, aos, start, rec, n, tmp, w);                                                         \
        if (wc) {                                                                                                        \
            const size_t lds_u = ((8 * (size_t)bp.cells + 15) & ~(size_t)15) + (size_t)bp.chunk * sizeof(BinOut<T, ND>); \
            HIPCHK(allow_lds(bin_unpermute_wc<T, ND>, lds_u));                                                           \
            hipLaunchKernelGGL((bin_unpermute_wc<T, ND>), dim3(std::min(bp.chunks, s->num_cu)), dim3(1024), lds_u, st, bp, n,    \
                               reinterpret_cast<const unsigned short *>(slot), M, Tr, start, Lb, pbin, tmp, out, ostride);                                                      \
        } else                                                                                                           \
        hipLaunchKernelGGL((bin_unpermute<T, ND>), dim3(ugrid), dim3(256), 0, st, n, slot, tmp, out, ostride);           \
    } break;
        switch (s->nDep) {
            BINNED_ND(1) BINNED_ND(2) BINNED_ND(3) BINNED_ND(4)
            default: return BSK_ERR_UNSUPPORTED;
        }
#undef BINNED_ND
        HIPCHK(hipGetLastError());
        return BSK_OK;
    }
}

// MIXED: variables of different orders (O = the largest); chosen by the call site so that every
// (NIND, O) instantiates one form only
template <typename T, int NIND, int O, bool MIXED>
static bsk_status gather_or_binned(bsk_spline s, const Params<T> &prm, long long n, T *out, long long ostride,
                                   const Wrt &w, hipStream_t st)
{
    if constexpr (NIND == 2) {                                    // surfaces: table streamed through LDS (bsk_slab_tu.hip)
        const bsk_status r2 = slab2_any<T>(s, MIXED, prm, n, out, ostride, w, st);
        if (r2 != BSK_ERR_UNSUPPORTED) return r2;
    }
    const bsk_status r = launch_eval_binned<T, NIND, O, MIXED>(s, prm, n, out, ostride, w, st);
    if (r != BSK_ERR_UNSUPPORTED) return r;
    return launch_eval_gather<T, NIND, O, MIXED>(s, prm, n, out, ostride, w, st);
}


template <typename T>
bsk_status gather_or_binned_any(bsk_spline s, bool mixed, const Params<T> &prm, long long n, T *out, long long ostride,
                                const Wrt &w, hipStream_t st)
{
    int omax = 0;
    for (int iv = 0; iv < s->nInd; ++iv) omax = std::max(omax, s->order[iv]);
    if (s->nInd < 1 || s->nInd > 3 || omax < 1 || omax > 6 || !s->coef_aos) return BSK_ERR_UNSUPPORTED;
#define GB_CASE(NIND, O, MIXED) case O: return gather_or_binned<T, NIND, O, MIXED>(s, prm, n, out, ostride, w, st);
    if (!mixed) {
        if (s->nInd == 1) switch (omax) { GB_CASE(1, 1, false) GB_CASE(1, 2, false) GB_CASE(1, 3, false) GB_CASE(1, 4, false) GB_CASE(1, 5, false) GB_CASE(1, 6, false) default: break; }
        else if (s->nInd == 2) switch (omax) { GB_CASE(2, 1, false) GB_CASE(2, 2, false) GB_CASE(2, 3, false) GB_CASE(2, 4, false) GB_CASE(2, 5, false) GB_CASE(2, 6, false) default: break; }
        else switch (omax) { GB_CASE(3, 1, false) GB_CASE(3, 2, false) GB_CASE(3, 3, false) GB_CASE(3, 4, false) GB_CASE(3, 5, false) GB_CASE(3, 6, false) default: break; }
    } else {
        if (s->nInd == 2) switch (omax) { GB_CASE(2, 2, true) GB_CASE(2, 3, true) GB_CASE(2, 4, true) GB_CASE(2, 5, true) GB_CASE(2, 6, true) default: break; }
        else if (s->nInd == 3) switch (omax) { GB_CASE(3, 2, true) GB_CASE(3, 3, true) GB_CASE(3, 4, true) GB_CASE(3, 5, true) GB_CASE(3, 6, true) default: break; }
    }
#undef GB_CASE
    return BSK_ERR_UNSUPPORTED;
}

// Fused jacobian of a large batch on an L2-resident table (three variables of one order, fp32: the eval_cellsort
// pipeline).  out[(dep * 3 + j) * n + i].  BSK_ERR_UNSUPPORTED when the shape is not covered: the caller then runs
// nInd derivative passes (which share one sort of the batch).
template <typename T>
bsk_status cellsort_jacobian_any(bsk_spline s, const Params<T> &prm, long long n, T *out, hipStream_t st)
{
    if (s->nInd != 3 || !s->same_order || !s->coef_aos || s->order[0] < 1 || s->order[0] > 6) return BSK_ERR_UNSUPPORTED;
    Wrt w;
    for (int iv = 0; iv < MAXI; ++iv) w.w[iv] = 0;
    switch (s->order[0]) {
#define CJ_CASE(O) case O: return launch_eval_binned<T, 3, O, false>(s, prm, n, out, 3 * n, w, st, true);
        CJ_CASE(1) CJ_CASE(2) CJ_CASE(3) CJ_CASE(4) CJ_CASE(5) CJ_CASE(6)
#undef CJ_CASE
        default: return BSK_ERR_UNSUPPORTED;
    }
}
template bsk_status cellsort_jacobian_any<float>(bsk_spline, const Params<float> &, long long, float *, hipStream_t);
template bsk_status cellsort_jacobian_any<double>(bsk_spline, const Params<double> &, long long, double *, hipStream_t);

template bsk_status gather_or_binned_any<float>(bsk_spline, bool, const Params<float> &, long long, float *, long long,
                                                const Wrt &, hipStream_t);
template bsk_status gather_or_binned_any<double>(bsk_spline, bool, const Params<double> &, long long, double *, long long,
                                                 const Wrt &, hipStream_t);
