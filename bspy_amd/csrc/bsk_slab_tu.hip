// eval_slab2 (bsk_slab.hpp) in its own translation unit: 80 instantiations (fp32 / fp64 x orders 2 - 6 x nDep 1 - 4 x
// one common order / mixed orders) compile beside bsk_gather_tu.hip instead of behind it.  Entry: slab2_any<T>.
#include "bsk_host.hpp"
#include "bsk_tile.hpp"
#include "bsk_gather.hpp"
#include "bsk_binned.hpp"
#include "bsk_slab.hpp"

// Surfaces whose table can be streamed through LDS once per chunk of 8192 points (bsk_slab.hpp): no global sort.
// (tools/mixed_small.py on the TomsNasty shape: a launch takes 53 us for any batch up to 1.2 M points - every slab staged by every
//  workgroup - against 14 - 34 us of the gather kernel up to 300 k points; even at 600 k: 54 against 65 us)
constexpr long long SLAB_MIN_POINTS = 1 << 19;

template <typename T, int O, bool MIXED>
static bsk_status launch_eval_slab2(bsk_spline s, const Params<T> &prm, long long n, T *out, long long ostride, const Wrt &w,
                                    hipStream_t st)
{
    if (s->nInd != 2 || s->nDep > 4 || !s->coef_aos || n < SLAB_MIN_POINTS) return BSK_ERR_UNSUPPORTED;
    if (s->variant == 7 || s->variant == 13 || s->variant == 14) return BSK_ERR_UNSUPPORTED;     // pinned to the other large-table paths
    const Desc<T> &d = desc_of<T>(s);
    const TileDesc<T> &td = tile_of<T>(s);
    if ((size_t)d.coef_len * sizeof(T) > SLAB_MAX_TABLE || s->ncoef[0] > 65535) return BSK_ERR_UNSUPPORTED;
    auto up16 = [](size_t b) { return (b + 15) & ~(size_t)15; };
    SlabPlan sp;
    size_t off = up16(sizeof(T) * (size_t)d.nk[1] * s->order[1]);
    sp.off_lut = (unsigned)off;
    off += up16(4 * (size_t)td.lut_len);
    sp.off_wcnt = (unsigned)off;
    off += up16(4 * (size_t)SLAB_WAVES * SLAB_MAX_PASS);
    sp.off_pstart = (unsigned)off;
    off += up16(4 * (size_t)(SLAB_MAX_PASS + 1) * (1 + SLAB_ROUND));   // pass starts of the chunk being ordered + of every chunk of the round
    sp.off_tab0 = (unsigned)off;
    const size_t lds_wg = s->lds_max;                             // one workgroup per CU
    if (off + 4096 > lds_wg) return BSK_ERR_UNSUPPORTED;
    // a pass of spp spans needs (spp + order0 - 1) table rows and order0 x (spp + order0) axis-table entries
    const size_t rowbytes = (size_t)s->ncoef[1] * s->nDep * sizeof(T), slicebytes = (size_t)s->order[0] * sizeof(T);
    const int S0 = s->ncoef[0] - s->order[0] + 1;
    const long long room = (long long)lds_wg - (long long)off - 256 - (long long)(s->order[0] - 1) * (long long)rowbytes -
                           (long long)s->order[0] * (long long)slicebytes;
    int spp = (int)std::min<long long>(room / (long long)(rowbytes + slicebytes), S0);
    if (spp < 1) return BSK_ERR_UNSUPPORTED;
    sp.npass = (S0 + spp - 1) / spp;
    if (sp.npass > SLAB_MAX_PASS) return BSK_ERR_UNSUPPORTED;
    spp = (S0 + sp.npass - 1) / sp.npass;                         // even passes
    sp.spp = spp;
    sp.rows = spp + s->order[0] - 1;
    sp.snk = spp + s->order[0];
    sp.off_slab = (unsigned)(off + up16((size_t)sp.snk * slicebytes));
    sp.total = (unsigned)(sp.off_slab + up16((size_t)sp.rows * rowbytes));
    if (sp.total > lds_wg) return BSK_ERR_UNSUPPORTED;
    // chunk-private scratch of the order: batch position | span of every point
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) {
        (void)hipGetLastError();
        return BSK_ERR_UNSUPPORTED;                               // the workspace may have to be (re)allocated
    }
    HIPCHK(s->bin_ws.reserve(4 * (size_t)n + 256));
    SlabPt<T> *spts = nullptr;                                    // (the points are read again from the caller's arrays)
    unsigned *sidx = reinterpret_cast<unsigned *>(s->bin_ws.p);
    // one slab (no ordering phase): chunks may be as short as 1024 points, so that a small batch still reaches two chunks per CU
    sp.chunk = SLAB_CHUNK;
    // (an LDS-resident table has eval_mixed as well: within 15 % either way up to 1 M points, this form 1.4 x faster at 10 M)
    if (sp.npass == 1 && s->aos_small && n < (1ll << 20)) return BSK_ERR_UNSUPPORTED;
    if (sp.npass == 1 && MIXED)
        sp.chunk = (int)std::max<long long>(1024, std::min<long long>(SLAB_CHUNK, ((n + 2 * s->num_cu - 1) / (2 * s->num_cu) + 1023) / 1024 * 1024));
    const long long nchunks = (n + sp.chunk - 1) / sp.chunk;
    int grid = (int)std::min<long long>(nchunks, (long long)s->num_cu);
    if (const char *g = getenv("BSK_SLAB_GRID")) grid = std::max(1, std::min(grid, atoi(g)));          // test knob: few workgroups = several rounds each
    static const int slab_dbg = getenv("BSK_SLAB_DBG") ? atoi(getenv("BSK_SLAB_DBG")) : 0;     // timing-only switches (tools/)
    const T *tab = static_cast<const T *>(s->tab);
    const T *aos = static_cast<const T *>(s->coef_aos);
    bool deriv = false;
    for (int iv = 0; iv < 2; ++iv) deriv |= w.w[iv] != 0;
#define SLAB_ND(ND)                                                                                               \
    case ND:                                                                                                      \
        if (deriv) {                                                                                              \
            HIPCHK(allow_lds(eval_slab2<T, O, ND, MIXED, true>, sp.total));                                       \
            hipLaunchKernelGGL((eval_slab2<T, O, ND, MIXED, true>), dim3(grid), dim3(SLAB_BLOCK), sp.total, st, d, td, sp, tab, s->lut, aos, \
                               prm, n, 0ll, spts, sidx, out, ostride, w, s->bad, slab_dbg);                       \
        } else {          /* plain evaluation: the recursions without their derivative branches */              \
            HIPCHK(allow_lds(eval_slab2<T, O, ND, MIXED, false>, sp.total));                                      \
            hipLaunchKernelGGL((eval_slab2<T, O, ND, MIXED, false>), dim3(grid), dim3(SLAB_BLOCK), sp.total, st, d, td, sp, tab, s->lut, aos, \
                               prm, n, 0ll, spts, sidx, out, ostride, w, s->bad, slab_dbg);                       \
        }                                                                                                         \
        break;
    switch (s->nDep) {
        SLAB_ND(1) SLAB_ND(2) SLAB_ND(3) SLAB_ND(4)
        default: return BSK_ERR_UNSUPPORTED;
    }
#undef SLAB_ND
    s->last_kernel = "eval_slab2";
    HIPCHK(hipGetLastError());
    return BSK_OK;
}

template <typename T>
bsk_status slab2_any(bsk_spline s, bool mixed, const Params<T> &prm, long long n, T *out, long long ostride, const Wrt &w,
                     hipStream_t st)
{
    int omax = 0;
    for (int iv = 0; iv < s->nInd; ++iv) omax = std::max(omax, s->order[iv]);
    if (s->nInd != 2 || omax < 1 || omax > 6) return BSK_ERR_UNSUPPORTED;
#define SLAB_CASE(O, M) case O: return launch_eval_slab2<T, O, M>(s, prm, n, out, ostride, w, st);
    if (!mixed) switch (omax) { SLAB_CASE(1, false) SLAB_CASE(2, false) SLAB_CASE(3, false) SLAB_CASE(4, false) SLAB_CASE(5, false) SLAB_CASE(6, false) default: break; }
    else switch (omax) { SLAB_CASE(2, true) SLAB_CASE(3, true) SLAB_CASE(4, true) SLAB_CASE(5, true) SLAB_CASE(6, true) default: break; }
#undef SLAB_CASE
    return BSK_ERR_UNSUPPORTED;
}

template bsk_status slab2_any<float>(bsk_spline, bool, const Params<float> &, long long, float *, long long, const Wrt &, hipStream_t);
template bsk_status slab2_any<double>(bsk_spline, bool, const Params<double> &, long long, double *, long long, const Wrt &, hipStream_t);
