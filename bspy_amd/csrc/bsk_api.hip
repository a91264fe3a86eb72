// C ABI of libbspy_amd.so (see include/bspy_amd.h).  Host side: table construction,
// kernel selection and launch, host<->device staging for BSK_HOST buffers.
#include "bsk_kernels.hpp"
#include "bsk_tile.hpp"
#include "bsk_stream.hpp"
#include "bsk_rowrot.hpp"
#include "bsk_uniform.hpp"
#include "bsk_rec32.hpp"
#include "bsk_host.hpp"

#include <condition_variable>
#include <mutex>
#include <thread>

struct HostPipe;
static void pipe_destroy(HostPipe *p);      // defined with HostPipe (pipelined host path)

static int ceil_log2(int x)
{
    int s = 0;
    while ((1 << s) < x) ++s;
    return s;
}

// Host construction of the axis table of one variable (layout: see Desc).
template <typename T>
static void build_axis_table(const T *knots, int order, int nk, std::vector<T> &tab)
{
    const size_t base = tab.size();
    tab.resize(base + (size_t)order * nk, T(0));
    for (int i = 0; i < nk; ++i) tab[base + i] = knots[i];
    for (int dgr = 1; dgr < order; ++dgr)
        for (int i = 0; i + dgr < nk; ++i) {
            const T den = knots[i + dgr] - knots[i];     // formed in T, as the reference does
            tab[base + (size_t)dgr * nk + i] = den > T(0) ? T(1.0 / (double)den) : T(0);
        }
}

// searchsorted(knots, x, 'right') clamped to [order, ncoef]: the reference's span rule
// (bspy/_spline_evaluation.py:6-8) on the host, for the bucket tables.
template <typename T>
static int host_span(const T *knots, int order, int ncoef, double x)
{
    int lo = order, hi = ncoef;
    while (lo < hi) {
        const int mid = (lo + hi) / 2;
        if ((double)knots[mid] <= x) lo = mid + 1; else hi = mid;
    }
    return lo;
}

static int lut_buckets(int order, int ncoef)
{
    int m = 16;
    while (m < 4 * (ncoef - order + 1) && m < 4096) m <<= 1;
    return m;
}

// Bucket table of one variable: bucket b covers parameters with int((u - lo) * scale) == b;
// entry = lo_ix | hi_ix << 16 brackets the span of every such u.  The bracket is widened by
// 1 % of a bucket on both sides, far more than the rounding of the device's (u - lo) * scale.
template <typename T>
static void build_lut(const T *knots, int order, int ncoef, int m, T &scale, int &steps, std::vector<unsigned> &lut)
{
    const double lo = (double)knots[order - 1], hi = (double)knots[ncoef];
    const double w = hi - lo;
    scale = T((double)m / w);
    const double inv = 1.0 / (double)scale;
    int widest = 0;
    for (int b = 0; b < m; ++b) {
        const int l = host_span<T>(knots, order, ncoef, lo + ((double)b - 0.01) * inv);
        const int h = (b == m - 1) ? ncoef : host_span<T>(knots, order, ncoef, lo + ((double)b + 1.01) * inv);
        widest = std::max(widest, h - l);
        lut.push_back((unsigned)l | ((unsigned)h << 16));
    }
    steps = ceil_log2(widest + 1);
}

template <typename T>
static bsk_status upload_tables(bsk_spline s, const void *const *knots, const void *coefs)
{
    Desc<T> &d = desc_of<T>(s);
    TileDesc<T> &td = tile_of<T>(s);
    std::vector<T> tab;
    std::vector<unsigned> lut;
    for (int iv = 0; iv < s->nInd; ++iv) {
        d.off[iv] = (int)tab.size();
        const T *k = static_cast<const T *>(knots[iv]);
        build_axis_table<T>(k, s->order[iv], d.nk[iv], tab);
        d.lo[iv] = k[s->order[iv] - 1];          // domain: reference _spline_evaluation.py:135-138
        d.hi[iv] = k[s->ncoef[iv]];
        // bucket table: lut_buckets() entries are reserved (4 x spans); the smallest power of two >= spans is
        // used when its brackets still hold at most two spans (every near-uniform knot vector): fewer
        // distinct entries per LDS bank = fewer bank conflicts in the span search
        td.lut_off[iv] = (int)lut.size();
        const int m_alloc = lut_buckets(s->order[iv], s->ncoef[iv]);
        int m_small = 16;
        while (m_small < s->ncoef[iv] - s->order[iv] + 1) m_small <<= 1;
        std::vector<unsigned> seg;
        int m_use = m_alloc;
        if (m_small < m_alloc) {
            build_lut<T>(k, s->order[iv], s->ncoef[iv], m_small, td.lut_scale[iv], td.lut_steps[iv], seg);
            if (td.lut_steps[iv] <= 1) m_use = m_small;
        }
        if (m_use == m_alloc) {
            seg.clear();
            build_lut<T>(k, s->order[iv], s->ncoef[iv], m_alloc, td.lut_scale[iv], td.lut_steps[iv], seg);
        }
        td.lut_m[iv] = m_use;
        seg.resize((size_t)m_alloc, seg.empty() ? 0u : seg.back());
        lut.insert(lut.end(), seg.begin(), seg.end());
    }
    if ((int)tab.size() != d.tab_len) return fail(BSK_ERR_INVALID, "internal: axis table size changed");
    if ((int)lut.size() != td.lut_len) return fail(BSK_ERR_INVALID, "internal: bucket table size changed");
    HIPCHK(hipSetDevice(s->device));
    if (d.tab_len) HIPCHK(hipMemcpy(s->tab, tab.data(), sizeof(T) * tab.size(), hipMemcpyHostToDevice));
    s->tab_host.assign(reinterpret_cast<const unsigned char *>(tab.data()),
                       reinterpret_cast<const unsigned char *>(tab.data()) + sizeof(T) * tab.size());
    if (td.lut_len) HIPCHK(hipMemcpy(s->lut, lut.data(), sizeof(unsigned) * lut.size(), hipMemcpyHostToDevice));
    if (d.coef_len) HIPCHK(hipMemcpy(s->coef, coefs, sizeof(T) * (size_t)d.coef_len, hipMemcpyHostToDevice));
    if (s->coef_aos) {
        // (nDep, cells) -> (cells, nDep)
        const size_t cells = (size_t)d.coef_len / (size_t)s->nDep;
        std::vector<T> aos((size_t)d.coef_len);
        const T *src = static_cast<const T *>(coefs);
        for (int dep = 0; dep < s->nDep; ++dep)
            for (size_t c = 0; c < cells; ++c) aos[c * s->nDep + dep] = src[(size_t)dep * cells + c];
        HIPCHK(hipMemcpy(s->coef_aos, aos.data(), sizeof(T) * aos.size(), hipMemcpyHostToDevice));
    }
    return BSK_OK;
}

template <typename T>
static bsk_status init_desc(bsk_spline s)
{
    Desc<T> &d = desc_of<T>(s);
    memset(&d, 0, sizeof(d));
    d.nInd = s->nInd;
    d.nDep = s->nDep;
    long long clen = 1;
    long long tlen = 0;
    for (int iv = 0; iv < s->nInd; ++iv) {
        d.order[iv] = s->order[iv];
        d.ncoef[iv] = s->ncoef[iv];
        d.nk[iv] = s->order[iv] + s->ncoef[iv];
        d.steps[iv] = ceil_log2(s->ncoef[iv] - s->order[iv] + 1);
        tlen += (long long)s->order[iv] * d.nk[iv];
        clen *= s->ncoef[iv];
    }
    d.cstride[s->nInd] = 1;
    for (int iv = s->nInd - 1; iv >= 0; --iv) {
        const long long st = (long long)d.cstride[iv + 1] * s->ncoef[iv];
        if (st * std::max(1, s->nDep) > 0x7fffffffLL) return fail(BSK_ERR_UNSUPPORTED, "coefficient table exceeds 2^31 elements");
        d.cstride[iv] = (int)st;
    }
    if (tlen > 0x7fffffffLL) return fail(BSK_ERR_UNSUPPORTED, "knot table too large");
    d.tab_len = (int)tlen;
    d.coef_len = (int)(clen * s->nDep);
    TileDesc<T> &td = tile_of<T>(s);
    memset(&td, 0, sizeof(td));
    for (int iv = 0; iv < s->nInd; ++iv) {
        td.lut_m[iv] = lut_buckets(s->order[iv], s->ncoef[iv]);   // reserved size; upload_tables may use fewer buckets
        td.lut_len += td.lut_m[iv];
    }
    auto up16 = [](size_t b) { return (unsigned)((b + 15) & ~(size_t)15); };
    td.tab_bytes = up16(sizeof(T) * (size_t)d.tab_len);
    td.lut_bytes = up16(sizeof(unsigned) * (size_t)td.lut_len);
    td.coef_bytes = up16(sizeof(T) * (size_t)d.coef_len);
    return BSK_OK;
}

template <typename T>
static bsk_status upload_uniform(bsk_spline s, const void *const *knots, const void *coefs);   // uniform-knot surface path
template <typename T>
static bsk_status upload_uniform_nd(bsk_spline s, const void *const *knots, const void *coefs);   // ... of the other LDS-resident shapes

extern "C" int bsk_version(void) { return BSK_VERSION; }
extern "C" const char *bsk_last_error(void) { return g_err.c_str(); }

extern "C" bsk_status bsk_device_count(int *count)
{
    if (!count) return fail(BSK_ERR_INVALID, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return fail(BSK_ERR_NO_DEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
    }
    *count = n;
    return BSK_OK;
}

extern "C" bsk_status bsk_spline_create(bsk_dtype dtype, int device, int nInd, int nDep, const int *order,
                                        const int *nCoef, const void *const *knots, const void *coefs,
                                        bsk_spline *out)
{
    if (!out) return fail(BSK_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (dtype != BSK_F32 && dtype != BSK_F64) return fail(BSK_ERR_INVALID, "dtype must be BSK_F32 or BSK_F64");
    if (nInd < 1 || nInd > MAXI) return fail(BSK_ERR_UNSUPPORTED, "nInd must be in [1, BSK_MAX_NIND]");
    if (nDep < 1) return fail(BSK_ERR_INVALID, "nDep must be >= 1");
    if (!order || !nCoef || !knots || !coefs) return fail(BSK_ERR_INVALID, "NULL argument");
    for (int iv = 0; iv < nInd; ++iv) {
        if (order[iv] < 1 || order[iv] > MAXO) return fail(BSK_ERR_UNSUPPORTED, "order must be in [1, BSK_MAX_ORDER]");
        if (nCoef[iv] < order[iv]) return fail(BSK_ERR_INVALID, "nCoef < order");
        if (!knots[iv]) return fail(BSK_ERR_INVALID, "NULL knots pointer");
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(BSK_ERR_NO_DEVICE, "no HIP device");
    if (device < 0 || device >= ndev) return fail(BSK_ERR_INVALID, "device index out of range");

    bsk_spline s = new bsk_spline_s();
    s->dtype = dtype;
    s->device = device;
    s->nInd = nInd;
    s->nDep = nDep;
    s->esize = dtype == BSK_F32 ? 4 : 8;
    s->same_order = true;
    for (int iv = 0; iv < nInd; ++iv) {
        s->order[iv] = order[iv];
        s->ncoef[iv] = nCoef[iv];
        if (order[iv] != order[0]) s->same_order = false;
    }
    bsk_status st = dtype == BSK_F32 ? init_desc<float>(s) : init_desc<double>(s);
    if (st != BSK_OK) { delete s; return st; }
    const int tab_len = dtype == BSK_F32 ? s->d32.tab_len : s->d64.tab_len;
    const int coef_len = dtype == BSK_F32 ? s->d32.coef_len : s->d64.coef_len;

    auto cleanup = [&]() {
        if (s->tab) (void)hipFree(s->tab);
        if (s->coef) (void)hipFree(s->coef);
        if (s->bad) (void)hipFree(s->bad);
        if (s->lut) (void)hipFree(s->lut);
        if (s->coef_aos) (void)hipFree(s->coef_aos);
        delete s;
    };
#define HIPCHK_C(expr)                                                                        \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) {                                                               \
            cleanup();                                                                        \
            return fail(BSK_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));      \
        }                                                                                     \
    } while (0)
    HIPCHK_C(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK_C(hipGetDeviceProperties(&prop, device));
    s->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    s->lds_max = std::min<size_t>(160 * 1024, prop.maxSharedMemoryPerMultiProcessor > 0
                                                  ? (size_t)prop.maxSharedMemoryPerMultiProcessor
                                                  : (size_t)64 * 1024);
    HIPCHK_C(hipMalloc(&s->tab, std::max<size_t>(16, s->esize * (size_t)tab_len)));
    HIPCHK_C(hipMalloc(&s->coef, std::max<size_t>(16, s->esize * (size_t)coef_len)));
    {
        const size_t tbytes = s->esize * (size_t)tab_len + s->esize * (size_t)coef_len;
        int omax = 0;
        for (int iv = 0; iv < nInd; ++iv) omax = std::max(omax, order[iv]);
        bool fixed = nInd <= 3 && omax <= 6;                  // gather / cell-order kernels: any mix of orders <= 6
        if (fixed && nDep <= 4 && tbytes + 8192 > s->lds_max)
            HIPCHK_C(hipMalloc(&s->coef_aos, std::max<size_t>(16, s->esize * (size_t)coef_len)));
        else if (fixed && nDep <= 4 && nInd == 2 && !s->same_order && omax >= 2) {
            // LDS-resident surface of mixed orders: eval_slab2 in one pass (explicit LDS reads) reads the same layout
            HIPCHK_C(hipMalloc(&s->coef_aos, std::max<size_t>(16, s->esize * (size_t)coef_len)));
            s->aos_small = true;
        }
    }
    HIPCHK_C(hipMalloc((void **)&s->lut, std::max<size_t>(16, sizeof(unsigned) * (size_t)(dtype == BSK_F32 ? s->t32.lut_len : s->t64.lut_len))));
    HIPCHK_C(hipMalloc((void **)&s->bad, sizeof(unsigned long long)));
    if (const char *v = getenv("BSK_VARIANT")) s->variant = atoi(v);
    HIPCHK_C(hipMemset(s->bad, 0xff, sizeof(unsigned long long)));
#undef HIPCHK_C
    st = dtype == BSK_F32 ? upload_tables<float>(s, knots, coefs) : upload_tables<double>(s, knots, coefs);
    if (st == BSK_OK) st = dtype == BSK_F32 ? upload_uniform<float>(s, knots, coefs) : upload_uniform<double>(s, knots, coefs);
    if (st == BSK_OK) st = dtype == BSK_F32 ? upload_uniform_nd<float>(s, knots, coefs) : upload_uniform_nd<double>(s, knots, coefs);
    if (st != BSK_OK) { s->uni_img.release(); cleanup(); return st; }
    *out = s;
    return BSK_OK;
}

extern "C" bsk_status bsk_spline_update(bsk_spline s, const void *const *knots, const void *coefs)
{
    if (!s || !knots || !coefs) return fail(BSK_ERR_INVALID, "NULL argument");
    HIPCHK(hipSetDevice(s->device));
    HIPCHK(hipDeviceSynchronize());
    const bsk_status st = s->dtype == BSK_F32 ? upload_tables<float>(s, knots, coefs) : upload_tables<double>(s, knots, coefs);
    if (st != BSK_OK) return st;
    const bsk_status su = s->dtype == BSK_F32 ? upload_uniform<float>(s, knots, coefs) : upload_uniform<double>(s, knots, coefs);
    if (su != BSK_OK) return su;
    return s->dtype == BSK_F32 ? upload_uniform_nd<float>(s, knots, coefs) : upload_uniform_nd<double>(s, knots, coefs);
}

extern "C" bsk_status bsk_spline_destroy(bsk_spline s)
{
    if (!s) return BSK_OK;
    (void)hipSetDevice(s->device);
    (void)hipDeviceSynchronize();
    if (s->tab) (void)hipFree(s->tab);
    if (s->coef) (void)hipFree(s->coef);
    if (s->bad) (void)hipFree(s->bad);
    if (s->lut) (void)hipFree(s->lut);
    if (s->coef_aos) (void)hipFree(s->coef_aos);
    s->in_ws.release();
    s->out_ws.release();
    s->aux_ws.release();
    s->curv_ws.release();
    s->bin_ws.release();
    if (s->ticket) (void)hipFree(s->ticket);
    s->ticket = nullptr;
    s->uni_img.release();
    if (s->pin) (void)hipHostFree(s->pin);
    s->pin = nullptr;
    pipe_destroy(s->pipe);
    s->pipe = nullptr;
    for (hipEvent_t &e : s->stage_ev)
        if (e) { (void)hipEventDestroy(e); e = nullptr; }
    delete s;
    return BSK_OK;
}

// ------------------------------------------------------------------------------------
// launch plans
// ------------------------------------------------------------------------------------
struct Plan {
    bool lds_coefs;
    size_t lds_bytes;
    int block, grid;
};

// LDS budget: the whole axis table always, the coefficient table when both fit in one
// CU's 160 KiB.  Workgroups are persistent: as many as fit per CU by LDS and by the
// 2048-lane limit, each striding over the batch.
template <typename T>
static Plan make_plan(bsk_spline s, long long n)
{
    const Desc<T> &d = desc_of<T>(s);
    Plan p;
    const size_t tab_b = sizeof(T) * (size_t)((d.tab_len + 1) & ~1);
    const size_t coef_b = sizeof(T) * (size_t)d.coef_len;
    const size_t limit = s->lds_max - 1024;   // keep clear of the hard limit
    p.lds_coefs = tab_b + coef_b <= limit;
    p.lds_bytes = tab_b + (p.lds_coefs ? coef_b : 0);
    int per_cu;   // workgroups per CU
    if (p.lds_bytes > limit / 2) { p.block = 1024; per_cu = 1; }
    else if (p.lds_bytes > limit / 4) { p.block = 1024; per_cu = 2; }
    else if (p.lds_bytes > limit / 8) { p.block = 512; per_cu = 4; }
    else { p.block = 256; per_cu = 8; }
    long long blocks = (n + p.block - 1) / p.block;
    p.grid = (int)std::max<long long>(1, std::min<long long>(blocks, (long long)s->num_cu * per_cu));
    return p;
}

template <typename T, int NIND, int O>
static bsk_status launch_eval_fixed(bsk_spline s, const Plan &p, const Params<T> &prm, long long n, T *out,
                                    long long ostride, const Wrt &w, hipStream_t st)
{
    const Desc<T> &d = desc_of<T>(s);
    const T *tab = static_cast<const T *>(s->tab);
    const T *coef = static_cast<const T *>(s->coef);
    if (p.lds_coefs) {
        HIPCHK(allow_lds(eval_fixed<T, NIND, O, true>, p.lds_bytes));
        s->last_kernel = "eval_fixed";
        hipLaunchKernelGGL((eval_fixed<T, NIND, O, true>), dim3(p.grid), dim3(p.block), p.lds_bytes, st, d, tab, coef,
                           prm, n, out, ostride, w, s->bad);
    } else {
        HIPCHK(allow_lds(eval_fixed<T, NIND, O, false>, p.lds_bytes));
        s->last_kernel = "eval_fixed";
        hipLaunchKernelGGL((eval_fixed<T, NIND, O, false>), dim3(p.grid), dim3(p.block), p.lds_bytes, st, d, tab, coef,
                           prm, n, out, ostride, w, s->bad);
    }
    HIPCHK(hipGetLastError());
    return BSK_OK;
}

template <typename T>
static bsk_status launch_eval_generic(bsk_spline s, const Params<T> &prm, long long n, T *out, long long ostride,
                                      const Wrt &w, hipStream_t st);

// Variables of different orders (or one order beyond the fixed-order kernels): eval_mixed at OMAX = the
// largest order - surfaces up to order 8, volumes up to 6, curves 9..12.
template <typename T, int NIND, int OMAX>
static bsk_status launch_eval_mixed(bsk_spline s, const Plan &p, const Params<T> &prm, long long n, T *out,
                                    long long ostride, const Wrt &w, hipStream_t st)
{
    const Desc<T> &d = desc_of<T>(s);
    const T *tab = static_cast<const T *>(s->tab);
    const T *coef = static_cast<const T *>(s->coef);
    if (p.lds_coefs) {
        HIPCHK(allow_lds(eval_mixed<T, NIND, OMAX, true>, p.lds_bytes));
        s->last_kernel = "eval_mixed";
        hipLaunchKernelGGL((eval_mixed<T, NIND, OMAX, true>), dim3(p.grid), dim3(p.block), p.lds_bytes, st, d, tab, coef,
                           prm, n, out, ostride, w, s->bad);
    } else if constexpr (NIND >= 4) {
        // four and five variables: only the LDS-resident form is instantiated (build time); larger tables
        // stay on the generic kernel
        return launch_eval_generic<T>(s, prm, n, out, ostride, w, st);
    } else {
        HIPCHK(allow_lds(eval_mixed<T, NIND, OMAX, false>, p.lds_bytes));
        s->last_kernel = "eval_mixed";
        hipLaunchKernelGGL((eval_mixed<T, NIND, OMAX, false>), dim3(p.grid), dim3(p.block), p.lds_bytes, st, d, tab, coef,
                           prm, n, out, ostride, w, s->bad);
    }
    HIPCHK(hipGetLastError());
    return BSK_OK;
}

template <typename T, int NIND, int O>
static bsk_status launch_jac_fixed(bsk_spline s, const Plan &p, const Params<T> &prm, long long n, T *out,
                                   hipStream_t st)
{
    const Desc<T> &d = desc_of<T>(s);
    const T *tab = static_cast<const T *>(s->tab);
    const T *coef = static_cast<const T *>(s->coef);
    if (p.lds_coefs) {
        HIPCHK(allow_lds(jac_fixed<T, NIND, O, true>, p.lds_bytes));
        s->last_kernel = "jac_fixed";
        hipLaunchKernelGGL((jac_fixed<T, NIND, O, true>), dim3(p.grid), dim3(p.block), p.lds_bytes, st, d, tab, coef,
                           prm, n, out, s->bad);
    } else {
        HIPCHK(allow_lds(jac_fixed<T, NIND, O, false>, p.lds_bytes));
        s->last_kernel = "jac_fixed";
        hipLaunchKernelGGL((jac_fixed<T, NIND, O, false>), dim3(p.grid), dim3(p.block), p.lds_bytes, st, d, tab, coef,
                           prm, n, out, s->bad);
    }
    HIPCHK(hipGetLastError());
    return BSK_OK;
}

template <typename T>
static bsk_status launch_eval_generic(bsk_spline s, const Params<T> &prm, long long n, T *out, long long ostride,
                                      const Wrt &w, hipStream_t st)
{
    const Desc<T> &d = desc_of<T>(s);
    const int block = 256;
    const long long blocks = (n + block - 1) / block;
    const int grid = (int)std::max<long long>(1, std::min<long long>(blocks, (long long)s->num_cu * 8));
    s->last_kernel = "eval_generic";
    hipLaunchKernelGGL((eval_generic<T>), dim3(grid), dim3(block), 0, st, d, static_cast<const T *>(s->tab),
                       static_cast<const T *>(s->coef), prm, n, out, ostride, w, s->bad);
    HIPCHK(hipGetLastError());
    return BSK_OK;
}

// LDS-resident kernels: bytes of the table image (axis tables, bucket tables, coefficients),
// or 0 when it does not fit in one CU's LDS.
template <typename T>
static size_t tile_lds_bytes(bsk_spline s, bool /*unused*/)
{
    const TileDesc<T> &td = tile_of<T>(s);
    const size_t b = (size_t)td.tab_bytes + td.lut_bytes + td.coef_bytes;
    for (int iv = 0; iv < s->nInd; ++iv)
        if (s->ncoef[iv] > 65535) return 0;
    return b <= s->lds_max ? b : 0;
}

static bool has_fixed_path(bsk_spline s);

// eval_rowrot / jac_rowrot / fused normal: surfaces of order 2 or 4 whose odd-stride image fits LDS
// Points per launch of the rowrot kernels (32-bit indices).  BSK_RR_CHUNK lowers it so that the
// chunk loop can be exercised by tests without a 2^28-point batch.
static long long rr_chunk_points()
{
    static const long long v = [] {
        const char *e = getenv("BSK_RR_CHUNK");
        const long long x = e ? atoll(e) : 0;
        return x > 0 && x < (long long)RR_MAX_CHUNK ? x : (long long)RR_MAX_CHUNK;
    }();
    return v;
}

template <typename T>
static size_t rowrot_lds_bytes(bsk_spline s)
{
    const TileDesc<T> &tdr = tile_of<T>(s);
    const size_t rs = (size_t)(s->ncoef[1] | 1);
    const size_t coef_b = ((size_t)s->nDep * s->ncoef[0] * rs * sizeof(T) + 15) & ~(size_t)15;
    const int nk0 = s->order[0] + s->ncoef[0], nk1 = s->order[1] + s->ncoef[1];
    const size_t rec_b = s->order[0] == 4 ? rr_records_bytes<T, 4>(nk0, nk1) : rr_records_bytes<T, 2>(nk0, nk1);
    return rec_b + rr_lut_bytes<T>(tdr.lut_len) + coef_b + TILE * sizeof(unsigned);
}

template <typename T>
static bool rowrot_applies(bsk_spline s)
{
    return has_fixed_path(s) && s->nInd == 2 && (s->order[0] == 2 || s->order[0] == 4) &&
           (s->variant == 0 || s->variant == 9) && rowrot_lds_bytes<T>(s) <= s->lds_max;
}

// ------------------------------------------------------------------------------------
// uniform-knot surface path (bsk_uniform.hpp): detection, unclamping, LDS image
// ------------------------------------------------------------------------------------
// Domain knots equally spaced, each end either clamped (order equal knots) or continuing the uniform spacing.
// "Equally" is relative to the SPAN: the kernels take the local coordinate from the stored knot and the nominal
// span width h, the reference from the stored knots alone, so a knot that sits d away from lo + j h moves the
// result by ~d / h.  Accepted: d <= 1024 ulp of h in fp64 (2e-13 h: linspace knots of a domain near the origin
// are 10 - 100 times closer), 32 ulp in fp32.  Equally spaced knots far from the origin relative to their
// spacing (lo = 1e6, h = 1e-3: d / h ~ 1e-7) are therefore NOT taken by this path.
template <typename T>
static bool axis_is_uniform(const T *k, int order, int ncoef, bool &clamp_lo, bool &clamp_hi)
{
    const int ns = ncoef - order + 1;
    const long double lo = k[order - 1], hi = k[ncoef];
    if (!(hi > lo) || ns < 1) return false;
    const long double h = (hi - lo) / ns;
    // "Equally spaced" = every stored knot is lo + j h up to the ROUNDING of the stored value (4 ulp of the largest
    // knot: np.linspace is within 1.5), and that rounding is small against the span (1024 ulp of h: the uniform
    // kernels form alpha from the nominal width, the reference from the stored knots; a deviation d of a knot moves
    // the result by d / h).  A user's perturbation of a few hundred ulp of h is NOT rounding: such knots keep the
    // general kernels (test_uniform_path_declines_perturbed_knots).
    const long double eps = std::numeric_limits<T>::epsilon();
    const long double mag = std::max(std::fabs(lo), std::fabs(hi));
    const long double tol = std::min((sizeof(T) == 8 ? 1024.0L : 32.0L) * eps * h, 4.0L * eps * mag);
    for (int j = 0; j <= ns; ++j)
        if (std::fabs((long double)k[order - 1 + j] - (lo + j * h)) > tol) return false;
    auto side = [&](bool low, bool &clamped) {
        bool cl = true, un = true;
        for (int i = 1; i < order; ++i) {
            const long double v = low ? (long double)k[order - 1 - i] : (long double)k[ncoef + i];
            const long double e = low ? lo : hi;
            if (v != e) cl = false;
            if (std::fabs(v - (low ? lo - i * h : hi + i * h)) > tol) un = false;
        }
        clamped = cl && order > 1;
        return cl || un;
    };
    return side(true, clamp_lo) && side(false, clamp_hi);
}

// Non-zero B-spline basis values at z on the span [kn[ix - 1], kn[ix]) (reference
// bspy/_spline_evaluation.py:11-18), extended precision.
static void host_basis(const long double *kn, int O, int ix, long double z, long double *b)
{
    for (int i = 0; i < O; ++i) b[i] = 0;
    b[O - 1] = 1;
    for (int degree = 1; degree < O; ++degree) {
        int bi = O - degree;
        for (int i = ix - degree; i < ix; ++i, ++bi) {
            const long double alpha = (z - kn[i]) / (kn[i + degree] - kn[i]);
            b[bi - 1] += (1 - alpha) * b[bi];
            b[bi] *= alpha;
        }
    }
}

// Unclamping matrix of one end for order O: the first O control points Q of the uniform-knot form
// from the first O control points P of the clamped form, Q = M P (only the first O - 1 change).
// Both bases span the polynomials of degree < O on the first span, so M = U^-1 C with the two bases
// sampled at O points; unit spans (the matrix does not depend on h).  The other end is the mirror image.
static bool unclamp_matrix(int O, std::vector<long double> &M)
{
    std::vector<long double> kc(2 * O + 1), ku(2 * O + 1);
    for (int i = 0; i <= 2 * O; ++i) { kc[i] = i < O ? 0 : i - O + 1; ku[i] = i - (O - 1); }
    std::vector<long double> U((size_t)O * O), C((size_t)O * O), b(O);
    for (int r = 0; r < O; ++r) {
        const long double z = (r + 0.5L) / O;
        host_basis(ku.data(), O, O, z, b.data());
        for (int i = 0; i < O; ++i) U[(size_t)r * O + i] = b[i];
        host_basis(kc.data(), O, O, z, b.data());
        for (int i = 0; i < O; ++i) C[(size_t)r * O + i] = b[i];
    }
    // Gauss-Jordan with partial pivoting on [U | C]
    for (int c = 0; c < O; ++c) {
        int piv = c;
        for (int r = c + 1; r < O; ++r) if (std::fabs(U[(size_t)r * O + c]) > std::fabs(U[(size_t)piv * O + c])) piv = r;
        if (U[(size_t)piv * O + c] == 0) return false;
        if (piv != c) for (int i = 0; i < O; ++i) { std::swap(U[(size_t)piv * O + i], U[(size_t)c * O + i]); std::swap(C[(size_t)piv * O + i], C[(size_t)c * O + i]); }
        const long double d = U[(size_t)c * O + c];
        for (int i = 0; i < O; ++i) { U[(size_t)c * O + i] /= d; C[(size_t)c * O + i] /= d; }
        for (int r = 0; r < O; ++r) {
            if (r == c) continue;
            const long double f = U[(size_t)r * O + c];
            if (f == 0) continue;
            for (int i = 0; i < O; ++i) { U[(size_t)r * O + i] -= f * U[(size_t)c * O + i]; C[(size_t)r * O + i] -= f * C[(size_t)c * O + i]; }
        }
    }
    M = C;
    return true;
}

// Apply Q = M P to the first (low) or last (mirrored) O control points of every line along variable iv
// of the (lines_outer, nc, inner) tensor w.
static void unclamp_axis(std::vector<long double> &w, size_t outer, int nc, size_t inner, int O,
                         const std::vector<long double> &M, bool low)
{
    std::vector<long double> p(O);
    for (size_t a = 0; a < outer; ++a)
        for (size_t c = 0; c < inner; ++c) {
            long double *line = w.data() + a * (size_t)nc * inner + c;
            for (int i = 0; i < O; ++i) p[i] = line[(size_t)(low ? i : nc - 1 - i) * inner];
            for (int i = 0; i < O; ++i) {
                long double q = 0;
                for (int j = 0; j < O; ++j) q += M[(size_t)i * O + j] * p[j];
                line[(size_t)(low ? i : nc - 1 - i) * inner] = q;
            }
        }
}

template <typename T>
static bool rowrot_applies(bsk_spline s);

template <typename T>
static bsk_status upload_uniform(bsk_spline s, const void *const *knots, const void *coefs)
{
    s->uni = false;
    if (!rowrot_applies<T>(s) || s->variant == 9) return BSK_OK;
    bool cl[2][2];
    for (int iv = 0; iv < 2; ++iv) {
        if (s->ncoef[iv] < 2 * s->order[iv]) return BSK_OK;           // the two ends must not overlap
        if (!axis_is_uniform<T>(static_cast<const T *>(knots[iv]), s->order[iv], s->ncoef[iv], cl[iv][0], cl[iv][1])) return BSK_OK;
    }
    const int O = s->order[0];
    // Unclamping multiplies the boundary control points by up to 6 (order 4) per variable and the corner
    // cells' rounding errors with them (13^2 ulp): harmless in fp64, not in fp32 (order 2 changes nothing).
    if (sizeof(T) == 4 && O > 2) return BSK_OK;
    std::vector<long double> M;
    if (!unclamp_matrix(O, M)) return BSK_OK;
    const int nc0 = s->ncoef[0], nc1 = s->ncoef[1], nDep = s->nDep;
    std::vector<long double> w((size_t)nDep * nc0 * nc1);
    const T *src = static_cast<const T *>(coefs);
    for (size_t i = 0; i < w.size(); ++i) w[i] = src[i];
    if (cl[0][0]) unclamp_axis(w, (size_t)nDep, nc0, (size_t)nc1, O, M, true);
    if (cl[0][1]) unclamp_axis(w, (size_t)nDep, nc0, (size_t)nc1, O, M, false);
    if (cl[1][0]) unclamp_axis(w, (size_t)nDep * nc0, nc1, 1, O, M, true);
    if (cl[1][1]) unclamp_axis(w, (size_t)nDep * nc0, nc1, 1, O, M, false);

    UniDesc<T> &ud = uni_of<T>(s);
    memset(&ud, 0, sizeof(ud));
    const bool cp_major = nDep <= 3;                              // [i0][i1][dep] (see bsk_uniform.hpp)
    const int rs = uni_row_stride(nc1, nDep, O);
    ud.rs = rs;
    unsigned off = 0;
    int nsmax = 1;
    for (int iv = 0; iv < 2; ++iv) {
        const T *k = static_cast<const T *>(knots[iv]);
        ud.ns[iv] = s->ncoef[iv] - s->order[iv] + 1;
        ud.ncoef[iv] = s->ncoef[iv];
        ud.lo[iv] = k[s->order[iv] - 1];
        ud.hi[iv] = k[s->ncoef[iv]];
        ud.inv_h[iv] = T((long double)ud.ns[iv] / ((long double)ud.hi[iv] - (long double)ud.lo[iv]));
        ud.kn_off[iv] = off;
        off += (unsigned)((ud.ns[iv] + 1) * sizeof(T));
        nsmax = std::max(nsmax, ud.ns[iv]);
    }
    ud.nDep = nDep;
    // bias of the span estimate: above the rounding of (u - lo) * inv_h (2 ulp of a value < ns), far below 1
    ud.eps = T(std::max<long double>(sizeof(T) == 8 ? 0x1p-30L : 0x1p-12L, 16.0L * nsmax * std::numeric_limits<T>::epsilon()));
    off = (off + 15u) & ~15u;
    ud.coef_off = off;
    off += (unsigned)((cp_major ? (size_t)nc0 * rs : (size_t)nDep * nc0 * rs) * sizeof(T));
    off = (off + 15u) & ~15u;
    ud.img_bytes = off;
    if ((size_t)off + TILE * sizeof(unsigned) > s->lds_max) return BSK_OK;
    std::vector<unsigned char> img(off, 0);
    for (int iv = 0; iv < 2; ++iv) {
        const T *k = static_cast<const T *>(knots[iv]);
        memcpy(img.data() + ud.kn_off[iv], k + s->order[iv] - 1, (size_t)(ud.ns[iv] + 1) * sizeof(T));
    }
    T *ic = reinterpret_cast<T *>(img.data() + ud.coef_off);
    for (int dep = 0; dep < nDep; ++dep)
        for (int i0 = 0; i0 < nc0; ++i0)
            for (int i1 = 0; i1 < nc1; ++i1) {
                const T v = T(w[((size_t)dep * nc0 + i0) * nc1 + i1]);
                if (cp_major) ic[(size_t)i0 * rs + (size_t)i1 * nDep + dep] = v;              // [i0][i1][dep]
                else ic[((size_t)dep * nc0 + i0) * rs + i1] = v;                               // [dep][i0][i1]
            }
    HIPCHK(hipSetDevice(s->device));
    HIPCHK(s->uni_img.reserve(off));
    HIPCHK(hipMemcpy(s->uni_img.p, img.data(), off, hipMemcpyHostToDevice));
    s->uni = true;
    return BSK_OK;
}

// The same image for curves, surfaces of order 1 / 3 / 5 and volumes (eval_stream_uni / jac_stream_uni):
// [domain knots of every variable][coefficients, unclamped, in the reference's (nDep, nCoef...) layout].
// Unclamping a variable multiplies the rounding errors of its boundary control points by the largest absolute row
// sum of its matrix; the path is taken while the product over the variables stays below 5000 (three variables of
// order 5 exceed it) and, as for surfaces, in fp64 only unless nothing has to be unclamped (order <= 2).
static size_t tile_lds_bytes_any(bsk_spline s);
template <typename T>
static bsk_status upload_uniform_nd(bsk_spline s, const void *const *knots, const void *coefs)
{
    s->uniN = false;
    if (s->uni || !has_fixed_path(s) || s->variant != 0 || s->order[0] > 5 || tile_lds_bytes_any(s) == 0) return BSK_OK;
    const int O = s->order[0], nInd = s->nInd, nDep = s->nDep;
    bool cl[3][2];
    for (int iv = 0; iv < nInd; ++iv) {
        if (s->ncoef[iv] < 2 * O) return BSK_OK;                       // the two ends must not overlap
        if (!axis_is_uniform<T>(static_cast<const T *>(knots[iv]), O, s->ncoef[iv], cl[iv][0], cl[iv][1])) return BSK_OK;
    }
    if (sizeof(T) == 4 && O > 2) return BSK_OK;
    std::vector<long double> M;
    if (!unclamp_matrix(O, M)) return BSK_OK;
    long double growth = 0, total = 1;
    for (int i = 0; i < O; ++i) {
        long double rsum = 0;
        for (int j = 0; j < O; ++j) rsum += std::fabs(M[(size_t)i * O + j]);
        growth = std::max(growth, rsum);
    }
    for (int iv = 0; iv < nInd; ++iv) if (cl[iv][0] || cl[iv][1]) total *= growth;
    if (total > 5000.0L) return BSK_OK;
    size_t ncp = 1;
    for (int iv = 0; iv < nInd; ++iv) ncp *= (size_t)s->ncoef[iv];
    std::vector<long double> w((size_t)nDep * ncp);
    const T *src = static_cast<const T *>(coefs);
    for (size_t i = 0; i < w.size(); ++i) w[i] = src[i];
    size_t outer = (size_t)nDep, inner = ncp;
    for (int iv = 0; iv < nInd; ++iv) {
        inner /= (size_t)s->ncoef[iv];
        if (cl[iv][0]) unclamp_axis(w, outer, s->ncoef[iv], inner, O, M, true);
        if (cl[iv][1]) unclamp_axis(w, outer, s->ncoef[iv], inner, O, M, false);
        outer *= (size_t)s->ncoef[iv];
    }
    UniDescN<T> &un = uniN_of<T>(s);
    memset(&un, 0, sizeof(un));
    unsigned off = 0;
    int nsmax = 1;
    for (int iv = 0; iv < nInd; ++iv) {
        const T *k = static_cast<const T *>(knots[iv]);
        un.ns[iv] = s->ncoef[iv] - O + 1;
        un.lo[iv] = k[O - 1];
        un.hi[iv] = k[s->ncoef[iv]];
        un.inv_h[iv] = T((long double)un.ns[iv] / ((long double)un.hi[iv] - (long double)un.lo[iv]));
        un.kn_off[iv] = off;
        off += (unsigned)((un.ns[iv] + 1) * sizeof(T));
        nsmax = std::max(nsmax, un.ns[iv]);
    }
    un.eps = T(std::max<long double>(sizeof(T) == 8 ? 0x1p-30L : 0x1p-12L, 16.0L * nsmax * std::numeric_limits<T>::epsilon()));
    off = (off + 15u) & ~15u;
    un.coef_off = off;
    const size_t total_b = (size_t)off + (((size_t)nDep * ncp * sizeof(T) + 15) & ~(size_t)15);
    if (total_b > s->lds_max) return BSK_OK;
    un.img_bytes = (unsigned)total_b;
    std::vector<unsigned char> img(total_b, 0);
    for (int iv = 0; iv < nInd; ++iv)
        memcpy(img.data() + un.kn_off[iv], static_cast<const T *>(knots[iv]) + O - 1, (size_t)(un.ns[iv] + 1) * sizeof(T));
    T *ic = reinterpret_cast<T *>(img.data() + un.coef_off);
    for (size_t i = 0; i < w.size(); ++i) ic[i] = T(w[i]);
    HIPCHK(hipSetDevice(s->device));
    HIPCHK(s->uni_img.reserve(total_b));
    HIPCHK(hipMemcpy(s->uni_img.p, img.data(), total_b, hipMemcpyHostToDevice));
    s->uniN = true;
    return BSK_OK;
}

template <typename T, bool NORMAL>
static bsk_status launch_jac_rowrot(bsk_spline s, const Params<T> &prm, long long n, T *out, int normalize, int negate,
                                    hipStream_t st)
{
    const Desc<T> &d = desc_of<T>(s);
    const TileDesc<T> &tdr = tile_of<T>(s);
    const size_t lds_rr = rowrot_lds_bytes<T>(s);
    const T *tab = static_cast<const T *>(s->tab);
    const T *coef = static_cast<const T *>(s->coef);
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(2, s->lds_max / lds_rr));
    if (s->uni) {
        // equally spaced knots: table-free front end on the unclamped image (bsk_uniform.hpp)
        const UniDesc<T> &ud = uni_of<T>(s);
        const size_t lds_u = (size_t)ud.img_bytes + TILE * sizeof(unsigned);
        const int per_cu_u = (int)std::max<size_t>(1, std::min<size_t>(2, s->lds_max / lds_u));
#define BSK_JUNI(O_, ND_)                                                                                               \
    do {                                                                                                                 \
        s->last_kernel = "jac_uni";                                        \
        HIPCHK(allow_lds(jac_uni<T, O_, NORMAL, ND_>, lds_u));                                                          \
        hipLaunchKernelGGL((jac_uni<T, O_, NORMAL, ND_>), dim3(g), dim3(TILE), lds_u, st, ud, s->uni_img.p, cp,         \
                           (unsigned)m, n0, out + n0, n, s->bad, normalize, negate);                                    \
    } while (0)
        const int ndu = (NORMAL || s->nDep > 3) ? 0 : s->nDep;
        const long long cmaxu = rr_chunk_points();
        for (long long n0 = 0; n0 < n; n0 += cmaxu) {
            const long long m = std::min<long long>(n - n0, cmaxu);
            const int g = (int)std::max<long long>(1, std::min<long long>((m + TILE - 1) / TILE, (long long)s->num_cu * per_cu_u));
            Params<T> cp = prm;
            for (int iv = 0; iv < s->nInd; ++iv) cp.p[iv] = prm.p[iv] + n0;
            if (s->order[0] == 4) {
                if constexpr (NORMAL) BSK_JUNI(4, 3);
                else switch (ndu) { case 1: BSK_JUNI(4, 1); break; case 2: BSK_JUNI(4, 2); break; case 3: BSK_JUNI(4, 3); break; default: BSK_JUNI(4, 0); }
            } else {
                if constexpr (NORMAL) BSK_JUNI(2, 3);
                else switch (ndu) { case 1: BSK_JUNI(2, 1); break; case 2: BSK_JUNI(2, 2); break; case 3: BSK_JUNI(2, 3); break; default: BSK_JUNI(2, 0); }
            }
            HIPCHK(hipGetLastError());
        }
#undef BSK_JUNI
        return BSK_OK;
    }
#define BSK_JROT(O_, ND_)                                                                                               \
    do {                                                                                                                 \
        s->last_kernel = "jac_rowrot";                                        \
        HIPCHK(allow_lds(jac_rowrot<T, O_, NORMAL, ND_>, lds_rr));                                                      \
        hipLaunchKernelGGL((jac_rowrot<T, O_, NORMAL, ND_>), dim3(g), dim3(TILE), lds_rr, st, d, tdr, tab, s->lut, coef, \
                           cp, (unsigned)m, n0, out + n0, n, s->bad, normalize, negate);                                \
    } while (0)
    // the number of dependent variables is a template constant up to 3 (0 = run-time loop)
    const int nd = (NORMAL || s->nDep > 3) ? 0 : s->nDep;
    // 32-bit point indices inside a launch: chunks of at most RR_MAX_CHUNK points
    const long long cmax = rr_chunk_points();
    for (long long n0 = 0; n0 < n; n0 += cmax) {
        const long long m = std::min<long long>(n - n0, cmax);
        const int g = (int)std::max<long long>(1, std::min<long long>((m + TILE - 1) / TILE, (long long)s->num_cu * per_cu));
        Params<T> cp = prm;
        for (int iv = 0; iv < s->nInd; ++iv) cp.p[iv] = prm.p[iv] + n0;
        if (s->order[0] == 4) {
            if constexpr (NORMAL) BSK_JROT(4, 0);
            else switch (nd) { case 1: BSK_JROT(4, 1); break; case 2: BSK_JROT(4, 2); break; case 3: BSK_JROT(4, 3); break; default: BSK_JROT(4, 0); }
        } else {
            if constexpr (NORMAL) BSK_JROT(2, 0);
            else switch (nd) { case 1: BSK_JROT(2, 1); break; case 2: BSK_JROT(2, 2); break; case 3: BSK_JROT(2, 3); break; default: BSK_JROT(2, 0); }
        }
        HIPCHK(hipGetLastError());
    }
#undef BSK_JROT
    return BSK_OK;
}

template <typename T, int NIND, int O>
static bsk_status launch_eval_lds(bsk_spline s, size_t lds, const Params<T> &prm, long long n, T *out, long long ostride,
                                  const Wrt &w, hipStream_t st)
{
    const Desc<T> &d = desc_of<T>(s);
    const TileDesc<T> &td = tile_of<T>(s);
    const T *tab = static_cast<const T *>(s->tab);
    const T *coef = static_cast<const T *>(s->coef);
    bool deriv = false;
    for (int iv = 0; iv < s->nInd; ++iv) deriv |= w.w[iv] != 0;
    const long long ntiles = (n + TILE - 1) / TILE;
    if constexpr (NIND == 2 && (O == 2 || O == 4)) {
        // surfaces of order 2 / 4: row rotation on an odd-stride LDS image
        if (rowrot_applies<T>(s)) {
            if (s->uni) {
                // equally spaced knots: table-free front end on the unclamped image (bsk_uniform.hpp)
                const UniDesc<T> &ud = uni_of<T>(s);
                const size_t lds_u = (size_t)ud.img_bytes + TILE * sizeof(unsigned);
                const int per_cu_u = (int)std::max<size_t>(1, std::min<size_t>(2, s->lds_max / lds_u));
#define BSK_UNI(DERIV_, ND_)                                                                                             \
    do {                                                                                                                 \
        s->last_kernel = "eval_uni";                                        \
        HIPCHK(allow_lds(eval_uni<T, O, DERIV_, ND_>, lds_u));                                                          \
        hipLaunchKernelGGL((eval_uni<T, O, DERIV_, ND_>), dim3(g), dim3(TILE), lds_u, st, ud, s->uni_img.p, cp,         \
                           (unsigned)m, n0, out + n0, ostride, w, s->bad);                                              \
    } while (0)
#define BSK_UNI_ND(DERIV_)                                                                                               \
    switch (s->nDep) {                                                                                                   \
    case 1: BSK_UNI(DERIV_, 1); break;                                                                                   \
    case 2: BSK_UNI(DERIV_, 2); break;                                                                                   \
    case 3: BSK_UNI(DERIV_, 3); break;                                                                                   \
    default: BSK_UNI(DERIV_, 0); break;                                                                                  \
    }
                const long long cmaxu = rr_chunk_points();
                for (long long n0 = 0; n0 < n; n0 += cmaxu) {
                    const long long m = std::min<long long>(n - n0, cmaxu);
                    const int g = (int)std::max<long long>(1, std::min<long long>((m + TILE - 1) / TILE, (long long)s->num_cu * per_cu_u));
                    Params<T> cp = prm;
                    for (int iv = 0; iv < s->nInd; ++iv) cp.p[iv] = prm.p[iv] + n0;
                    if (deriv) { BSK_UNI_ND(true); } else { BSK_UNI_ND(false); }
                    HIPCHK(hipGetLastError());
                }
#undef BSK_UNI_ND
#undef BSK_UNI
                return BSK_OK;
            }
            if constexpr (sizeof(T) == 4 && O == 4) {
                // fp32 bicubics: everything in 16-byte LDS reads (bsk_rec32.hpp); BSK_VARIANT=9 keeps eval_rowrot
                const size_t lds_r = r32_lds_bytes(s->ncoef[0] - 3, s->ncoef[1] - 3, td.lut_len, s->ncoef[0], s->ncoef[1]);
                if (s->variant == 0 && s->nDep >= 1 && s->nDep <= 4 && lds_r + 256 <= s->lds_max) {
                    const int per_cu_r = (int)std::max<size_t>(1, std::min<size_t>(2, s->lds_max / lds_r));
#define BSK_R32(DERIV_, ND_)                                                                                             \
    do {                                                                                                                 \
        HIPCHK(allow_lds(eval_rec32<DERIV_, ND_>, lds_r));                                                              \
        hipLaunchKernelGGL((eval_rec32<DERIV_, ND_>), dim3(g), dim3(TILE), lds_r, st, d, td, tab, s->lut, coef, cp,     \
                           (unsigned)m, n0, out + n0, ostride, w, s->bad);                                              \
    } while (0)
#define BSK_R32_ND(DERIV_)                                                                                               \
    switch (s->nDep) {                                                                                                   \
    case 1: BSK_R32(DERIV_, 1); break;                                                                                   \
    case 2: BSK_R32(DERIV_, 2); break;                                                                                   \
    case 3: BSK_R32(DERIV_, 3); break;                                                                                   \
    default: BSK_R32(DERIV_, 4); break;                                                                                  \
    }
                    s->last_kernel = "eval_rec32";
                    const long long cmaxr = rr_chunk_points();
                    for (long long n0 = 0; n0 < n; n0 += cmaxr) {
                        const long long m = std::min<long long>(n - n0, cmaxr);
                        const int g = (int)std::max<long long>(1, std::min<long long>((m + TILE - 1) / TILE, (long long)s->num_cu * per_cu_r));
                        Params<T> cp = prm;
                        for (int iv = 0; iv < s->nInd; ++iv) cp.p[iv] = prm.p[iv] + n0;
                        if (deriv) { BSK_R32_ND(true); } else { BSK_R32_ND(false); }
                        HIPCHK(hipGetLastError());
                    }
#undef BSK_R32_ND
#undef BSK_R32
                    return BSK_OK;
                }
            }
            const size_t lds_rr = rowrot_lds_bytes<T>(s);
            const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(2, s->lds_max / lds_rr));
            // the number of dependent variables is a template constant up to 3 (0 = run-time loop)
#define BSK_ROWROT(DERIV_, ND_)                                                                                          \
    do {                                                                                                                 \
        s->last_kernel = "eval_rowrot";                                        \
        HIPCHK(allow_lds(eval_rowrot<T, O, DERIV_, ND_>, lds_rr));                                                      \
        hipLaunchKernelGGL((eval_rowrot<T, O, DERIV_, ND_>), dim3(g), dim3(TILE), lds_rr, st, d, td, tab, s->lut, coef, \
                           cp, (unsigned)m, n0, out + n0, ostride, w, s->bad);                                          \
    } while (0)
#define BSK_ROWROT_ND(DERIV_)                                                                                            \
    switch (s->nDep) {                                                                                                   \
    case 1: BSK_ROWROT(DERIV_, 1); break;                                                                                \
    case 2: BSK_ROWROT(DERIV_, 2); break;                                                                                \
    case 3: BSK_ROWROT(DERIV_, 3); break;                                                                                \
    default: BSK_ROWROT(DERIV_, 0); break;                                                                               \
    }
            // 32-bit point indices inside a launch: chunks of at most RR_MAX_CHUNK points
            const long long cmax = rr_chunk_points();
            for (long long n0 = 0; n0 < n; n0 += cmax) {
                const long long m = std::min<long long>(n - n0, cmax);
                const int g = (int)std::max<long long>(1, std::min<long long>((m + TILE - 1) / TILE, (long long)s->num_cu * per_cu));
                Params<T> cp = prm;
                for (int iv = 0; iv < s->nInd; ++iv) cp.p[iv] = prm.p[iv] + n0;
                if (deriv) { BSK_ROWROT_ND(true); } else { BSK_ROWROT_ND(false); }
                HIPCHK(hipGetLastError());
            }
#undef BSK_ROWROT_ND
#undef BSK_ROWROT
            return BSK_OK;
        }
    }
    if constexpr (O <= 5) {
        if (s->uniN) {
            // equally spaced knots: table-free front end on the unclamped image (bsk_uniform.hpp)
            const UniDescN<T> &un = uniN_of<T>(s);
            const size_t lds_u = un.img_bytes;
            const int per_cu_u = (int)std::max<size_t>(1, std::min<size_t>(2, s->lds_max / lds_u));
            const int grid_u = (int)std::max<long long>(1, std::min<long long>(ntiles, (long long)s->num_cu * per_cu_u));
            s->last_kernel = "eval_stream_uni";
            if (deriv) {
                HIPCHK(allow_lds(eval_stream_uni<T, NIND, O, true>, lds_u));
                hipLaunchKernelGGL((eval_stream_uni<T, NIND, O, true>), dim3(grid_u), dim3(STREAM_BLOCK), lds_u, st, d, un, s->uni_img.p,
                                   prm, n, out, ostride, w, s->bad);
            } else {
                HIPCHK(allow_lds(eval_stream_uni<T, NIND, O, false>, lds_u));
                hipLaunchKernelGGL((eval_stream_uni<T, NIND, O, false>), dim3(grid_u), dim3(STREAM_BLOCK), lds_u, st, d, un, s->uni_img.p,
                                   prm, n, out, ostride, w, s->bad);
            }
            HIPCHK(hipGetLastError());
            return BSK_OK;
        }
    }
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(2, s->lds_max / lds));
    const int grid = (int)std::max<long long>(1, std::min<long long>(ntiles, (long long)s->num_cu * per_cu));
    if (deriv) {
        HIPCHK(allow_lds(eval_stream<T, NIND, O, true>, lds));
        s->last_kernel = "eval_stream";
        hipLaunchKernelGGL((eval_stream<T, NIND, O, true>), dim3(grid), dim3(STREAM_BLOCK), lds, st, d, td, tab, s->lut, coef, prm,
                           n, out, ostride, w, s->bad);
    } else {
        HIPCHK(allow_lds(eval_stream<T, NIND, O, false>, lds));
        s->last_kernel = "eval_stream";
        hipLaunchKernelGGL((eval_stream<T, NIND, O, false>), dim3(grid), dim3(STREAM_BLOCK), lds, st, d, td, tab, s->lut, coef, prm,
                           n, out, ostride, w, s->bad);
    }
    HIPCHK(hipGetLastError());
    return BSK_OK;
}

// Fast-path coverage: nInd 1..3, one common order 1..6.
static bool has_fixed_path(bsk_spline s)
{
    return s->same_order && s->nInd >= 1 && s->nInd <= 3 && s->order[0] >= 1 && s->order[0] <= 6;
}

static size_t tile_lds_bytes_any(bsk_spline s)
{
    return s->dtype == BSK_F32 ? tile_lds_bytes<float>(s, false) : tile_lds_bytes<double>(s, false);
}

// The LDS-staging kernels (eval_fixed / jac_fixed / eval_mixed) keep at least the axis tables in LDS.
static bool axis_tables_fit_lds(bsk_spline s)
{
    const size_t tab_len = s->dtype == BSK_F32 ? (size_t)s->d32.tab_len : (size_t)s->d64.tab_len;
    return s->esize * (tab_len + 2) + 1024 <= s->lds_max;
}

#define BSK_ORDER_SWITCH(NIND, CALL)                        \
    switch (s->order[0]) {                                  \
        case 1: return CALL(NIND, 1);                       \
        case 2: return CALL(NIND, 2);                       \
        case 3: return CALL(NIND, 3);                       \
        case 4: return CALL(NIND, 4);                       \
        case 5: return CALL(NIND, 5);                       \
        case 6: return CALL(NIND, 6);                       \
        default: break;                                     \
    }

template <typename T, int NIND, int O>
static bsk_status launch_jac_stream(bsk_spline s, size_t lds, const Params<T> &prm, long long n, T *out, hipStream_t st)
{
    const Desc<T> &d = desc_of<T>(s);
    const TileDesc<T> &td = tile_of<T>(s);
    const long long ntiles = (n + STREAM_BLOCK - 1) / STREAM_BLOCK;
    if constexpr (O <= 5) {
        if (s->uniN) {
            const UniDescN<T> &un = uniN_of<T>(s);
            const size_t lds_u = un.img_bytes;
            const int per_cu_u = (int)std::max<size_t>(1, std::min<size_t>(2, s->lds_max / lds_u));
            const int grid_u = (int)std::max<long long>(1, std::min<long long>(ntiles, (long long)s->num_cu * per_cu_u));
            HIPCHK(allow_lds(jac_stream_uni<T, NIND, O>, lds_u));
            s->last_kernel = "jac_stream_uni";
            hipLaunchKernelGGL((jac_stream_uni<T, NIND, O>), dim3(grid_u), dim3(STREAM_BLOCK), lds_u, st, d, un, s->uni_img.p, prm, n,
                               out, s->bad);
            HIPCHK(hipGetLastError());
            return BSK_OK;
        }
    }
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(2, s->lds_max / lds));
    const int grid = (int)std::max<long long>(1, std::min<long long>(ntiles, (long long)s->num_cu * per_cu));
    HIPCHK(allow_lds(jac_stream<T, NIND, O>, lds));
    s->last_kernel = "jac_stream";
    hipLaunchKernelGGL((jac_stream<T, NIND, O>), dim3(grid), dim3(STREAM_BLOCK), lds, st, d, td, static_cast<const T *>(s->tab),
                       s->lut, static_cast<const T *>(s->coef), prm, n, out, s->bad);
    HIPCHK(hipGetLastError());
    return BSK_OK;
}

// The asm-LDS kernels (tile / stream) cover orders 1..5: at order 6 their register windows
// (36-value slabs, 20 table values per variable) no longer fit without spilling, which those
// kernels must never do (check_spills.py); order 6 runs on eval_fixed - except curves, whose single
// window row fits up to order 8 (dispatched before this switch).
#define BSK_ORDER_SWITCH5(NIND, CALL)                       \
    switch (s->order[0]) {                                  \
        case 1: return CALL(NIND, 1);                       \
        case 2: return CALL(NIND, 2);                       \
        case 3: return CALL(NIND, 3);                       \
        case 4: return CALL(NIND, 4);                       \
        case 5: return CALL(NIND, 5);                       \
        default: break;                                     \
    }

template <typename T>
static bsk_status dispatch_eval(bsk_spline s, const Params<T> &prm, long long n, T *out, long long ostride,
                                const Wrt &w, hipStream_t st)
{
    if (n <= 0) return BSK_OK;
    if (s->nInd == 1 && s->order[0] >= 6 && s->order[0] <= 8 && s->variant != 1) {
        // curves of order 6..8: one window row per dependent variable still fits the registers
        const size_t lds = tile_lds_bytes<T>(s, false);
        if (lds != 0) switch (s->order[0]) {
            case 6: return launch_eval_lds<T, 1, 6>(s, lds, prm, n, out, ostride, w, st);
            case 7: return launch_eval_lds<T, 1, 7>(s, lds, prm, n, out, ostride, w, st);
            default: return launch_eval_lds<T, 1, 8>(s, lds, prm, n, out, ostride, w, st);
        }
    }
    if (has_fixed_path(s) && s->variant != 1 && s->order[0] <= 5) {
        // table image fits in LDS: eval_rowrot (surfaces of order 2 / 4) or eval_stream
        const size_t lds = tile_lds_bytes<T>(s, false);
        if (lds != 0) {
#define CALL_LDS(NIND, O) launch_eval_lds<T, NIND, O>(s, lds, prm, n, out, ostride, w, st)
            if (s->nInd == 1) { BSK_ORDER_SWITCH5(1, CALL_LDS) }
            else if (s->nInd == 2) { BSK_ORDER_SWITCH5(2, CALL_LDS) }
            else { BSK_ORDER_SWITCH5(3, CALL_LDS) }
#undef CALL_LDS
        }
    }
    if (has_fixed_path(s) && s->coef_aos && s->variant != 1) {
        // table too large for LDS: control-point-major gather / cell-order pipeline (bsk_gather_tu.hip)
        const bsk_status r = gather_or_binned_any<T>(s, false, prm, n, out, ostride, w, st);
        if (r != BSK_ERR_UNSUPPORTED) return r;
    }
    if (has_fixed_path(s) && axis_tables_fit_lds(s)) {
        const Plan p = make_plan<T>(s, n);
#define CALL_EVAL(NIND, O) launch_eval_fixed<T, NIND, O>(s, p, prm, n, out, ostride, w, st)
        if (s->nInd == 1) { BSK_ORDER_SWITCH(1, CALL_EVAL) }
        else if (s->nInd == 2) { BSK_ORDER_SWITCH(2, CALL_EVAL) }
        else { BSK_ORDER_SWITCH(3, CALL_EVAL) }
#undef CALL_EVAL
    }
    // What is left: variables of different orders, or orders beyond the fixed-order kernels.
    if (s->nInd >= 1 && s->nInd <= 5 && s->variant != 1 && axis_tables_fit_lds(s)) {
        int omax = 0;
        for (int iv = 0; iv < s->nInd; ++iv) omax = std::max(omax, s->order[iv]);
        if (!s->same_order && omax >= 2 && omax <= 6 && s->coef_aos) {
            // table too large for LDS: control-point-major gather / cell-order pipeline at O = omax; an LDS-resident
            // surface: eval_slab2 with the whole table as its one slab (batches below 65536 points: eval_mixed)
            const bsk_status r = s->aos_small ? slab2_any<T>(s, true, prm, n, out, ostride, w, st)
                                              : gather_or_binned_any<T>(s, true, prm, n, out, ostride, w, st);
            if (r != BSK_ERR_UNSUPPORTED) return r;
        }
        // eval_mixed: surfaces up to order 8, volumes up to 6, curves up to 12, four variables up to order 4, five up to 3
        const Plan p = make_plan<T>(s, n);
#define MIXED_CASE(NIND, OM) case OM: return launch_eval_mixed<T, NIND, OM>(s, p, prm, n, out, ostride, w, st);
        if (s->nInd == 1 && omax >= 9 && omax <= 12) {
            switch (omax) { MIXED_CASE(1, 9) MIXED_CASE(1, 10) MIXED_CASE(1, 11) MIXED_CASE(1, 12) default: break; }
        } else if (s->nInd == 2 && omax >= 2 && omax <= 8) {
            switch (omax) { MIXED_CASE(2, 2) MIXED_CASE(2, 3) MIXED_CASE(2, 4) MIXED_CASE(2, 5) MIXED_CASE(2, 6) MIXED_CASE(2, 7) MIXED_CASE(2, 8) default: break; }
        } else if (s->nInd == 3 && omax >= 2 && omax <= 6) {
            switch (omax) { MIXED_CASE(3, 2) MIXED_CASE(3, 3) MIXED_CASE(3, 4) MIXED_CASE(3, 5) MIXED_CASE(3, 6) default: break; }
        } else if (s->nInd == 4 && omax >= 2 && omax <= 4) {
            switch (omax) { MIXED_CASE(4, 2) MIXED_CASE(4, 3) MIXED_CASE(4, 4) default: break; }
        } else if (s->nInd == 5 && omax >= 2 && omax <= 3) {
            switch (omax) { MIXED_CASE(5, 2) MIXED_CASE(5, 3) default: break; }
        }
#undef MIXED_CASE
    }
    return launch_eval_generic<T>(s, prm, n, out, ostride, w, st);
}

template <typename T>
static bsk_status dispatch_jac(bsk_spline s, const Params<T> &prm, long long n, T *out, hipStream_t st)
{
    if (n <= 0) return BSK_OK;
    if (rowrot_applies<T>(s)) return launch_jac_rowrot<T, false>(s, prm, n, out, 0, 0, st);
    if (s->nInd == 1 && s->order[0] >= 6 && s->order[0] <= 8 && s->variant != 1) {
        const size_t lds = tile_lds_bytes<T>(s, false);
        if (lds != 0) switch (s->order[0]) {
            case 6: return launch_jac_stream<T, 1, 6>(s, lds, prm, n, out, st);
            case 7: return launch_jac_stream<T, 1, 7>(s, lds, prm, n, out, st);
            default: return launch_jac_stream<T, 1, 8>(s, lds, prm, n, out, st);
        }
    }
    if (has_fixed_path(s) && s->variant != 1 && s->order[0] <= 5) {
        const size_t lds = tile_lds_bytes<T>(s, false);
        if (lds != 0) {
#define CALL_JACS(NIND, O) launch_jac_stream<T, NIND, O>(s, lds, prm, n, out, st)
            if (s->nInd == 1) { BSK_ORDER_SWITCH5(1, CALL_JACS) }
            else if (s->nInd == 2) { BSK_ORDER_SWITCH5(2, CALL_JACS) }
            else {
                switch (s->order[0]) {      // three variables, order 5: two 25-value windows spill -> jac_fixed
                    case 1: return CALL_JACS(3, 1);
                    case 2: return CALL_JACS(3, 2);
                    case 3: return CALL_JACS(3, 3);
                    case 4: return CALL_JACS(3, 4);
                    default: break;
                }
            }
#undef CALL_JACS
        }
    }
    // L2-resident table and a batch large enough for the cell-order pipeline: nInd derivative passes through
    // it beat jac_fixed's batch-order gathers (cfg5 shape, 10 M points: 4.8 -> 2.7 ms)
    const bool passes_in_cell_order = s->coef_aos && s->nInd >= 2 && n >= (1ll << 18) && s->variant != 7 && s->variant != 1;
    if (has_fixed_path(s) && axis_tables_fit_lds(s) && !passes_in_cell_order) {
        const Plan p = make_plan<T>(s, n);
#define CALL_JAC(NIND, O) launch_jac_fixed<T, NIND, O>(s, p, prm, n, out, st)
        if (s->nInd == 1) { BSK_ORDER_SWITCH(1, CALL_JAC) }
        else if (s->nInd == 2) { BSK_ORDER_SWITCH(2, CALL_JAC) }
        else { BSK_ORDER_SWITCH(3, CALL_JAC) }
#undef CALL_JAC
    }
    // variables of different orders (or orders beyond jac_fixed) with LDS-resident tables: jac_mixed
    // (mixed-order surface, 10 M points, order (3,4): jac_mixed 309 us; two derivative passes of eval_slab2 in one slab 286 us -
    //  measured, not worth a second path)
    if ((s->nInd == 2 || s->nInd == 3) && s->variant != 1 && axis_tables_fit_lds(s)) {
        int omax = 0;
        for (int iv = 0; iv < s->nInd; ++iv) omax = std::max(omax, s->order[iv]);
        const Plan p = make_plan<T>(s, n);
        if (p.lds_coefs && omax >= 2 && omax <= (s->nInd == 2 ? 8 : 6)) {
            const Desc<T> &d = desc_of<T>(s);
            const T *tab = static_cast<const T *>(s->tab);
            const T *coef = static_cast<const T *>(s->coef);
#define JMIX(NIND, OM)                                                                                              \
    case OM:                                                                                                        \
        s->last_kernel = "jac_mixed";                                        \
        HIPCHK(allow_lds(jac_mixed<T, NIND, OM>, p.lds_bytes));                                                     \
        hipLaunchKernelGGL((jac_mixed<T, NIND, OM>), dim3(p.grid), dim3(p.block), p.lds_bytes, st, d, tab, coef, prm, n, \
                           out, s->bad);                                                                            \
        HIPCHK(hipGetLastError());                                                                                  \
        return BSK_OK;
            if (s->nInd == 2) switch (omax) { JMIX(2, 2) JMIX(2, 3) JMIX(2, 4) JMIX(2, 5) JMIX(2, 6) JMIX(2, 7) JMIX(2, 8) default: break; }
            else switch (omax) { JMIX(3, 2) JMIX(3, 3) JMIX(3, 4) JMIX(3, 5) JMIX(3, 6) default: break; }
#undef JMIX
        }
    }
    // large batches on L2-resident tables of three variables: the fused jacobian of the cell-order pipeline
    if (s->nInd == 3 && has_fixed_path(s) && s->coef_aos && s->variant != 1) {
        const bsk_status r = cellsort_jacobian_any<T>(s, prm, n, out, st);
        if (r != BSK_ERR_UNSUPPORTED) return r;
    }
    // otherwise nInd unit-derivative passes, as the reference does (_spline_evaluation.py:205-213), each on the
    // best evaluation kernel for the shape (mixed orders: eval_mixed / gather; else eval_generic)
    for (int j = 0; j < s->nInd; ++j) {
        Wrt w;
        for (int iv = 0; iv < MAXI; ++iv) w.w[iv] = (iv == j);
        s->bin_reuse = j > 0;           // cell-order pipeline: the batch is counted, scanned and scattered once
        bsk_status r = dispatch_eval<T>(s, prm, n, out + (long long)j * n, (long long)s->nInd * n, w, st);
        s->bin_reuse = false;
        if (r != BSK_OK) return r;
    }
    return BSK_OK;
}

// ------------------------------------------------------------------------------------
// out-of-domain record
// ------------------------------------------------------------------------------------
static bsk_status read_bad(bsk_spline s, hipStream_t st, int64_t *first_bad)
{
    unsigned long long v = NO_BAD;
    HIPCHK(hipMemcpyAsync(&v, s->bad, sizeof(v), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (v != NO_BAD) {
        HIPCHK(hipMemsetAsync(s->bad, 0xff, sizeof(v), st));
        HIPCHK(hipStreamSynchronize(st));
        if (first_bad) *first_bad = (int64_t)v;
        return fail(BSK_ERR_DOMAIN, "parameter outside the spline's domain at flat index " + std::to_string(v));
    }
    if (first_bad) *first_bad = -1;
    return BSK_OK;
}

// Grow a per-handle device workspace.  Growth frees and allocates (a device-wide synchronisation) and cannot be
// recorded by a stream capture: a capturing stream gets a clear error instead of a broken capture (run the
// call once outside the capture, the workspace then has its size).  The workspaces belong to the handle:
// one handle is used from one stream at a time (include/bspy_amd.h).
static bsk_status ws_reserve(DevBuf &b, size_t bytes, hipStream_t st)
{
    if (bytes <= b.cap) return BSK_OK;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone)
        return fail(BSK_ERR_INVALID, "a workspace of this spline handle has to grow, which a stream capture cannot record: "
                                     "run the same call once outside the capture first");
    (void)hipGetLastError();
    HIPCHK(b.reserve(bytes));
    return BSK_OK;
}

extern "C" bsk_status bsk_domain_status(bsk_spline s, void *stream, int64_t *first_bad)
{
    if (!s) return fail(BSK_ERR_INVALID, "NULL handle");
    HIPCHK(hipSetDevice(s->device));
    return read_bad(s, static_cast<hipStream_t>(stream), first_bad);
}

// ------------------------------------------------------------------------------------
// evaluate / derivative / jacobian
// ------------------------------------------------------------------------------------
// Small BSK_HOST calls (the reference's single-point API lands here; up to 65536 points, the measured
// crossover with the staged path - BSK_SMALL_POINTS overrides): no staging copies.  The
// parameters are written into a pinned, device-mapped host buffer that the kernels read directly
// over PCIe; results and the out-of-domain record come back through the same buffer; one
// synchronisation.  (Five small hipMemcpyAsync calls plus the record query cost ~70 us per call;
// this path ~20 us.)
static long long small_call_points()
{
    static const long long v = getenv("BSK_SMALL_POINTS") ? atoll(getenv("BSK_SMALL_POINTS")) : 65536;
    return v;
}

__global__ void publish_bad(unsigned long long *bad, unsigned long long *slot)
{
    const unsigned long long v = *bad;
    *slot = v;
    if (v != NO_BAD) *bad = NO_BAD;
}

static bsk_status reserve_pin(bsk_spline s, size_t bytes)
{
    if (bytes <= s->pin_cap) return BSK_OK;
    if (s->pin) (void)hipHostFree(s->pin);
    s->pin = nullptr;
    s->pin_cap = 0;
    HIPCHK(hipHostMalloc(&s->pin, bytes, hipHostMallocMapped));
    s->pin_cap = bytes;
    return BSK_OK;
}

// layout of the pinned buffer: [first_bad u64][pad to 64][inputs: rows_in x n][outputs: rows_out x n]
template <typename T, typename Launch>
static bsk_status run_small(bsk_spline s, const void *const *uvw, long long n, int rows_out, void *out, hipStream_t st,
                            int64_t *first_bad, Launch launch)
{
    const size_t in_b = sizeof(T) * (size_t)n * s->nInd, out_b = sizeof(T) * (size_t)n * rows_out;
    bsk_status r = reserve_pin(s, 64 + ((in_b + 63) & ~(size_t)63) + out_b + 64);
    if (r != BSK_OK) return r;
    char *hp = static_cast<char *>(s->pin);
    void *dp = nullptr;
    HIPCHK(hipHostGetDevicePointer(&dp, s->pin, 0));
    char *dbase = static_cast<char *>(dp);
    unsigned long long *hslot = reinterpret_cast<unsigned long long *>(hp);
    T *hin = reinterpret_cast<T *>(hp + 64), *din = reinterpret_cast<T *>(dbase + 64);
    const size_t ooff = 64 + ((in_b + 63) & ~(size_t)63);
    T *hout = reinterpret_cast<T *>(hp + ooff), *dout = reinterpret_cast<T *>(dbase + ooff);
    Params<T> prm;
    for (int iv = 0; iv < MAXI; ++iv) prm.p[iv] = nullptr;
    for (int iv = 0; iv < s->nInd; ++iv) {
        memcpy(hin + (size_t)iv * n, uvw[iv], sizeof(T) * (size_t)n);
        prm.p[iv] = din + (size_t)iv * n;
    }
    r = launch(prm, dout);
    if (r != BSK_OK) return r;
    hipLaunchKernelGGL(publish_bad, dim3(1), dim3(1), 0, st, s->bad, reinterpret_cast<unsigned long long *>(dbase));
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));
    const unsigned long long v = *hslot;
    if (v != NO_BAD) {
        if (first_bad) *first_bad = (int64_t)v;
        return fail(BSK_ERR_DOMAIN, "parameter outside the spline's domain at flat index " + std::to_string(v));
    }
    memcpy(out, hout, out_b);
    if (first_bad) *first_bad = -1;
    return BSK_OK;
}

// ------------------------------------------------------------------------------------
// Large BSK_HOST batches: pipelined staging.
// hipMemcpy from / to pageable memory stages through the runtime's own pinned buffers on one thread
// (~15 GB/s: 26.5 ms for the 400 MB of a 10 M-point cfg2 call, around a 0.11 ms kernel).  Here the
// batch is cut into chunks; a small pool of threads copies a chunk between the caller's arrays and
// pinned staging buffers while the DMA engines and the kernel work on the previous chunk (two
// slots; H2D and D2H on their own streams so both PCIe directions are busy).
// ------------------------------------------------------------------------------------
class CopyPool {
public:
    struct Job { char *dst; const char *src; size_t bytes; };
    explicit CopyPool(int threads)
    {
        for (int i = 0; i < threads; ++i) th_.emplace_back([this] { work(); });
    }
    ~CopyPool()
    {
        { std::lock_guard<std::mutex> l(m_); stop_ = true; }
        cv_.notify_all();
        for (auto &t : th_) t.join();
    }
    int threads() const { return (int)th_.size(); }
    // copy `bytes` in slices on all workers (and the caller); returns when done
    void copy(void *dst, const void *src, size_t bytes)
    {
        const size_t parts = (size_t)th_.size() + 1;
        const size_t slice = ((bytes / parts) + 4095) & ~(size_t)4095;
        std::vector<Job> mine;
        {
            std::lock_guard<std::mutex> l(m_);
            size_t off = slice;                              // the caller takes the first slice
            while (off < bytes) {
                const size_t b = std::min(slice, bytes - off);
                jobs_.push_back(Job{static_cast<char *>(dst) + off, static_cast<const char *>(src) + off, b});
                ++pending_;
                off += b;
            }
        }
        cv_.notify_all();
        memcpy(dst, src, std::min(slice, bytes));
        std::unique_lock<std::mutex> l(m_);
        done_.wait(l, [this] { return pending_ == 0; });
    }

private:
    void work()
    {
        for (;;) {
            Job j;
            {
                std::unique_lock<std::mutex> l(m_);
                cv_.wait(l, [this] { return stop_ || !jobs_.empty(); });
                if (stop_ && jobs_.empty()) return;
                j = jobs_.back();
                jobs_.pop_back();
            }
            memcpy(j.dst, j.src, j.bytes);
            {
                std::lock_guard<std::mutex> l(m_);
                if (--pending_ == 0) done_.notify_all();
            }
        }
    }
    std::vector<std::thread> th_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    std::vector<Job> jobs_;
    size_t pending_ = 0;
    bool stop_ = false;
};

// BSK_HOST_THREADS: copy threads of the pipelined host path (default 7 workers + the caller; 0 = staged path)
static CopyPool *copy_pool()
{
    static const int want = [] {
        const char *e = getenv("BSK_HOST_THREADS");
        if (e) return std::max(0, atoi(e));
        const unsigned hc = std::thread::hardware_concurrency();
        return (int)std::min<unsigned>(7, hc > 2 ? hc / 2 : 0);
    }();
    if (want <= 0) return nullptr;
    static CopyPool pool(want);
    return &pool;
}

struct HostPipe {
    static constexpr int SLOTS = 2;
    void *pin_in[SLOTS] = {nullptr, nullptr}, *pin_out[SLOTS] = {nullptr, nullptr};
    size_t in_cap = 0, out_cap = 0;
    DevBuf din[SLOTS], dout[SLOTS];
    unsigned long long *pin_bad = nullptr;                   // SLOTS entries, device mapped
    hipStream_t s_in = nullptr, s_k = nullptr, s_out = nullptr;
    hipEvent_t e_in[SLOTS] = {}, e_k[SLOTS] = {}, e_out[SLOTS] = {};
    bool ready = false;
    ~HostPipe()
    {
        for (int i = 0; i < SLOTS; ++i) {
            if (pin_in[i]) (void)hipHostFree(pin_in[i]);
            if (pin_out[i]) (void)hipHostFree(pin_out[i]);
            din[i].release();
            dout[i].release();
            if (e_in[i]) (void)hipEventDestroy(e_in[i]);
            if (e_k[i]) (void)hipEventDestroy(e_k[i]);
            if (e_out[i]) (void)hipEventDestroy(e_out[i]);
        }
        if (pin_bad) (void)hipHostFree(pin_bad);
        if (s_in) (void)hipStreamDestroy(s_in);
        if (s_k) (void)hipStreamDestroy(s_k);
        if (s_out) (void)hipStreamDestroy(s_out);
    }
};

static void pipe_destroy(HostPipe *p) { delete p; }

static constexpr long long PIPE_CHUNK = 1ll << 20;           // points per pipeline chunk
static constexpr long long PIPE_MIN_POINTS = 1ll << 21;      // below this the staged path is as fast

static bsk_status pipe_prepare(bsk_spline s, size_t in_b, size_t out_b)
{
    if (!s->pipe) s->pipe = new HostPipe();
    HostPipe &p = *s->pipe;
    if (!p.ready) {
        HIPCHK(hipStreamCreateWithFlags(&p.s_in, hipStreamNonBlocking));
        HIPCHK(hipStreamCreateWithFlags(&p.s_k, hipStreamNonBlocking));
        HIPCHK(hipStreamCreateWithFlags(&p.s_out, hipStreamNonBlocking));
        for (int i = 0; i < HostPipe::SLOTS; ++i) {
            HIPCHK(hipEventCreateWithFlags(&p.e_in[i], hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&p.e_k[i], hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&p.e_out[i], hipEventDisableTiming));
        }
        HIPCHK(hipHostMalloc((void **)&p.pin_bad, sizeof(unsigned long long) * HostPipe::SLOTS, hipHostMallocMapped));
        p.ready = true;
    }
    for (int i = 0; i < HostPipe::SLOTS; ++i) {
        if (in_b > p.in_cap) {
            if (p.pin_in[i]) (void)hipHostFree(p.pin_in[i]);
            p.pin_in[i] = nullptr;
            HIPCHK(hipHostMalloc(&p.pin_in[i], in_b, hipHostMallocDefault));
        }
        if (out_b > p.out_cap) {
            if (p.pin_out[i]) (void)hipHostFree(p.pin_out[i]);
            p.pin_out[i] = nullptr;
            HIPCHK(hipHostMalloc(&p.pin_out[i], out_b, hipHostMallocDefault));
        }
        HIPCHK(p.din[i].reserve(in_b));
        HIPCHK(p.dout[i].reserve(out_b));
    }
    p.in_cap = std::max(p.in_cap, in_b);
    p.out_cap = std::max(p.out_cap, out_b);
    return BSK_OK;
}

// rows_out result rows of n values each at out (row stride n); launch(prm, m, dout, stream) enqueues the
// kernels of one chunk of m points writing rows of stride m
template <typename T, typename Launch>
static bsk_status run_piped(bsk_spline s, CopyPool *pool, const void *const *uvw, long long n, int rows_out, void *out,
                            int64_t *first_bad, Launch launch)
{
    const long long C = PIPE_CHUNK;
    const long long K = (n + C - 1) / C;
    bsk_status r = pipe_prepare(s, sizeof(T) * (size_t)C * s->nInd, sizeof(T) * (size_t)C * rows_out);
    if (r != BSK_OK) return r;
    HostPipe &p = *s->pipe;
    void *dbadv = nullptr;
    HIPCHK(hipHostGetDevicePointer(&dbadv, p.pin_bad, 0));
    unsigned long long *dbad = static_cast<unsigned long long *>(dbadv);
    bsk_status result = BSK_OK;
    auto drain = [&](long long k) -> bsk_status {            // results of chunk k -> caller's arrays
        const int slot = (int)(k % HostPipe::SLOTS);
        const long long start = k * C, m = std::min(C, n - start);
        HIPCHK(hipEventSynchronize(p.e_out[slot]));
        const unsigned long long v = p.pin_bad[slot];
        if (v != NO_BAD) {
            if (first_bad) *first_bad = start + (int64_t)v;
            return fail(BSK_ERR_DOMAIN, "parameter outside the spline's domain at flat index " + std::to_string(start + (long long)v));
        }
        const T *src = static_cast<const T *>(p.pin_out[slot]);
        for (int row = 0; row < rows_out; ++row)
            pool->copy(static_cast<T *>(out) + (size_t)row * n + start, src + (size_t)row * m, sizeof(T) * (size_t)m);
        return BSK_OK;
    };
    for (long long k = 0; k < K && result == BSK_OK; ++k) {
        const int slot = (int)(k % HostPipe::SLOTS);
        const long long start = k * C, m = std::min(C, n - start);
        if (k >= HostPipe::SLOTS) {
            result = drain(k - HostPipe::SLOTS);               // also frees this slot's buffers
            if (result != BSK_OK) break;
        }
        T *hin = static_cast<T *>(p.pin_in[slot]);
        T *din = static_cast<T *>(p.din[slot].p), *dout = static_cast<T *>(p.dout[slot].p);
        Params<T> prm;
        for (int iv = 0; iv < MAXI; ++iv) prm.p[iv] = nullptr;
        for (int iv = 0; iv < s->nInd; ++iv) {
            pool->copy(hin + (size_t)iv * m, static_cast<const T *>(uvw[iv]) + start, sizeof(T) * (size_t)m);
            prm.p[iv] = din + (size_t)iv * m;
        }
        HIPCHK(hipMemcpyAsync(din, hin, sizeof(T) * (size_t)m * s->nInd, hipMemcpyHostToDevice, p.s_in));
        HIPCHK(hipEventRecord(p.e_in[slot], p.s_in));
        HIPCHK(hipStreamWaitEvent(p.s_k, p.e_in[slot], 0));
        r = launch(prm, m, dout, p.s_k);
        if (r != BSK_OK) { result = r; break; }
        hipLaunchKernelGGL(publish_bad, dim3(1), dim3(1), 0, p.s_k, s->bad, dbad + slot);
        HIPCHK(hipEventRecord(p.e_k[slot], p.s_k));
        HIPCHK(hipStreamWaitEvent(p.s_out, p.e_k[slot], 0));
        HIPCHK(hipMemcpyAsync(p.pin_out[slot], dout, sizeof(T) * (size_t)m * rows_out, hipMemcpyDeviceToHost, p.s_out));
        HIPCHK(hipEventRecord(p.e_out[slot], p.s_out));
    }
    if (result == BSK_OK)
        for (long long k = std::max<long long>(0, K - HostPipe::SLOTS); k < K && result == BSK_OK; ++k) result = drain(k);
    if (result != BSK_OK) {                                  // leave nothing in flight behind an error
        (void)hipStreamSynchronize(p.s_in);
        (void)hipStreamSynchronize(p.s_k);
        (void)hipStreamSynchronize(p.s_out);
        (void)hipMemsetAsync(s->bad, 0xff, sizeof(unsigned long long), p.s_k);
        (void)hipStreamSynchronize(p.s_k);
        return result;
    }
    if (first_bad) *first_bad = -1;
    return BSK_OK;
}

// BSK_HOST batches are processed in chunks so the staging buffers stay bounded.
static long long host_chunk_points()
{
    // BSK_HOST_CHUNK lowers the chunk so that tests can run the chunk loops on small batches
    static const long long v = [] {
        const char *e = getenv("BSK_HOST_CHUNK");
        const long long x = e ? atoll(e) : 0;
        return x > 0 ? x : 1ll << 24;
    }();
    return v;
}

template <typename T>
static bsk_status run_points(bsk_spline s, bool jac, const int *wrt, const void *const *uvw, long long n, bsk_mem mem,
                             void *out, hipStream_t st, int64_t *first_bad)
{
    Wrt w;
    for (int iv = 0; iv < MAXI; ++iv) w.w[iv] = (wrt && iv < s->nInd) ? wrt[iv] : 0;
    for (int iv = 0; iv < s->nInd; ++iv)
        if (w.w[iv] < 0) return fail(BSK_ERR_INVALID, "negative derivative order");
    const int outs = jac ? s->nDep * s->nInd : s->nDep;   // output rows per point
    if (first_bad) *first_bad = -1;
    if (n == 0) return BSK_OK;

    if (mem == BSK_DEVICE) {
        Params<T> prm;
        for (int iv = 0; iv < MAXI; ++iv) prm.p[iv] = iv < s->nInd ? static_cast<const T *>(uvw[iv]) : nullptr;
        return jac ? dispatch_jac<T>(s, prm, n, static_cast<T *>(out), st)
                   : dispatch_eval<T>(s, prm, n, static_cast<T *>(out), n, w, st);
    }

    if (n <= small_call_points())
        return run_small<T>(s, uvw, n, outs, out, st, first_bad, [&](const Params<T> &prm, T *dout) {
            return jac ? dispatch_jac<T>(s, prm, n, dout, st) : dispatch_eval<T>(s, prm, n, dout, n, w, st);
        });

    if (n >= PIPE_MIN_POINTS) {
        if (CopyPool *pool = copy_pool()) {
            HIPCHK(hipStreamSynchronize(st));                 // the call is blocking: order it after the caller's stream
            return run_piped<T>(s, pool, uvw, n, outs, out, first_bad, [&](const Params<T> &prm, long long m, T *dout, hipStream_t ks) {
                return jac ? dispatch_jac<T>(s, prm, m, dout, ks) : dispatch_eval<T>(s, prm, m, dout, m, w, ks);
            });
        }
    }

    // host buffers: stage chunk by chunk
    const long long chunk = std::min(n, host_chunk_points());
    HIPCHK(s->in_ws.reserve(sizeof(T) * (size_t)chunk * s->nInd));
    HIPCHK(s->out_ws.reserve(sizeof(T) * (size_t)chunk * outs));
    T *din = static_cast<T *>(s->in_ws.p);
    T *dout = static_cast<T *>(s->out_ws.p);
    for (long long start = 0; start < n; start += chunk) {
        const long long m = std::min(chunk, n - start);
        Params<T> prm;
        for (int iv = 0; iv < MAXI; ++iv) prm.p[iv] = nullptr;
        for (int iv = 0; iv < s->nInd; ++iv) {
            HIPCHK(hipMemcpyAsync(din + (size_t)iv * m, static_cast<const T *>(uvw[iv]) + start, sizeof(T) * (size_t)m,
                                  hipMemcpyHostToDevice, st));
            prm.p[iv] = din + (size_t)iv * m;
        }
        bsk_status r = jac ? dispatch_jac<T>(s, prm, m, dout, st) : dispatch_eval<T>(s, prm, m, dout, m, w, st);
        if (r != BSK_OK) return r;
        for (int row = 0; row < outs; ++row)
            HIPCHK(hipMemcpyAsync(static_cast<T *>(out) + (size_t)row * n + start, dout + (size_t)row * m,
                                  sizeof(T) * (size_t)m, hipMemcpyDeviceToHost, st));
        int64_t bad = -1;
        r = read_bad(s, st, &bad);     // also synchronises the chunk
        if (r == BSK_ERR_DOMAIN) {
            if (first_bad) *first_bad = start + bad;
            return r;
        }
        if (r != BSK_OK) return r;
    }
    return BSK_OK;
}

static bsk_status check_call(bsk_spline s, const void *const *uvw, int64_t n, void *out)
{
    if (!s) return fail(BSK_ERR_INVALID, "NULL handle");
    if (n < 0) return fail(BSK_ERR_INVALID, "negative point count");
    if (n > 0) {
        if (!uvw || !out) return fail(BSK_ERR_INVALID, "NULL buffer");
        for (int iv = 0; iv < s->nInd; ++iv)
            if (!uvw[iv]) return fail(BSK_ERR_INVALID, "NULL parameter pointer");
    }
    return BSK_OK;
}

extern "C" bsk_status bsk_evaluate(bsk_spline s, const int *wrt, const void *const *uvw, int64_t n, bsk_mem mem,
                                   void *out, void *stream, int64_t *first_bad)
{
    bsk_status r = check_call(s, uvw, n, out);
    if (r != BSK_OK) return r;
    HIPCHK(hipSetDevice(s->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    return s->dtype == BSK_F32 ? run_points<float>(s, false, wrt, uvw, n, mem, out, st, first_bad)
                               : run_points<double>(s, false, wrt, uvw, n, mem, out, st, first_bad);
}

extern "C" bsk_status bsk_jacobian(bsk_spline s, const void *const *uvw, int64_t n, bsk_mem mem, void *out,
                                   void *stream, int64_t *first_bad)
{
    bsk_status r = check_call(s, uvw, n, out);
    if (r != BSK_OK) return r;
    HIPCHK(hipSetDevice(s->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    return s->dtype == BSK_F32 ? run_points<float>(s, true, nullptr, uvw, n, mem, out, st, first_bad)
                               : run_points<double>(s, true, nullptr, uvw, n, mem, out, st, first_bad);
}

// ------------------------------------------------------------------------------------
// normal
// ------------------------------------------------------------------------------------
template <typename T>
static bsk_status run_normal(bsk_spline s, const void *const *uvw, long long n, bsk_mem mem, int normalize, int negate,
                             void *out, hipStream_t st, int64_t *first_bad)
{
    const int big = std::max(s->nInd, s->nDep);
    if (first_bad) *first_bad = -1;
    if (n == 0) return BSK_OK;
    const long long chunk = mem == BSK_HOST ? std::min(n, host_chunk_points()) : n;
    // jacobian workspace (device) for one chunk, unless the normal is fused into the jacobian kernel
    const bool fused = s->nDep == 3 && rowrot_applies<T>(s);
    // one chunk of m points on stream ks (jacobian workspace in aux_ws when the normal is not fused)
    auto normal_chunk = [&](const Params<T> &prm, long long m, T *dout, hipStream_t ks) -> bsk_status {
        if (fused) return launch_jac_rowrot<T, true>(s, prm, m, dout, normalize, negate, ks);
        T *dj = static_cast<T *>(s->aux_ws.p);
        const bsk_status r = dispatch_jac<T>(s, prm, m, dj, ks);
        if (r != BSK_OK) return r;
        const int grid = (int)std::max<long long>(1, std::min<long long>((m + 255) / 256, (long long)s->num_cu * 8));
        hipLaunchKernelGGL((normal_epilogue<T>), dim3(grid), dim3(256), 0, ks, dj, s->nInd, s->nDep, m, normalize, negate, dout);
        HIPCHK(hipGetLastError());
        return BSK_OK;
    };
    if (mem == BSK_HOST && n <= small_call_points()) {
        if (!fused) if (bsk_status r_ = ws_reserve(s->aux_ws, sizeof(T) * (size_t)n * s->nDep * s->nInd, st); r_ != BSK_OK) return r_;
        return run_small<T>(s, uvw, n, big, out, st, first_bad,
                            [&](const Params<T> &prm, T *dout) { return normal_chunk(prm, n, dout, st); });
    }
    if (mem == BSK_HOST && n >= PIPE_MIN_POINTS) {
        if (CopyPool *pool = copy_pool()) {
            if (!fused) if (bsk_status r_ = ws_reserve(s->aux_ws, sizeof(T) * (size_t)PIPE_CHUNK * s->nDep * s->nInd, st); r_ != BSK_OK) return r_;
            HIPCHK(hipStreamSynchronize(st));
            return run_piped<T>(s, pool, uvw, n, big, out, first_bad, normal_chunk);
        }
    }
    if (!fused) if (bsk_status r_ = ws_reserve(s->aux_ws, sizeof(T) * (size_t)chunk * s->nDep * s->nInd, st); r_ != BSK_OK) return r_;
    T *djac = static_cast<T *>(s->aux_ws.p);
    T *din = nullptr, *dout = static_cast<T *>(out);
    if (mem == BSK_HOST) {
        HIPCHK(s->in_ws.reserve(sizeof(T) * (size_t)chunk * s->nInd));
        HIPCHK(s->out_ws.reserve(sizeof(T) * (size_t)chunk * big));
        din = static_cast<T *>(s->in_ws.p);
        dout = static_cast<T *>(s->out_ws.p);
    }
    for (long long start = 0; start < n; start += chunk) {
        const long long m = std::min(chunk, n - start);
        Params<T> prm;
        for (int iv = 0; iv < MAXI; ++iv) prm.p[iv] = nullptr;
        for (int iv = 0; iv < s->nInd; ++iv) {
            if (mem == BSK_HOST) {
                HIPCHK(hipMemcpyAsync(din + (size_t)iv * m, static_cast<const T *>(uvw[iv]) + start, sizeof(T) * (size_t)m,
                                      hipMemcpyHostToDevice, st));
                prm.p[iv] = din + (size_t)iv * m;
            } else {
                prm.p[iv] = static_cast<const T *>(uvw[iv]);
            }
        }
        bsk_status r;
        if (fused) {
            // surface in 3-D on the LDS image: tangents never leave the registers
            r = launch_jac_rowrot<T, true>(s, prm, m, dout, normalize, negate, st);
            if (r != BSK_OK) return r;
        } else {
            r = dispatch_jac<T>(s, prm, m, djac, st);
            if (r != BSK_OK) return r;
            const int block = 256;
            const int grid = (int)std::max<long long>(1, std::min<long long>((m + block - 1) / block, (long long)s->num_cu * 8));
            hipLaunchKernelGGL((normal_epilogue<T>), dim3(grid), dim3(block), 0, st, djac, s->nInd, s->nDep, m, normalize,
                               negate, dout);
            HIPCHK(hipGetLastError());
        }
        if (mem == BSK_HOST) {
            for (int row = 0; row < big; ++row)
                HIPCHK(hipMemcpyAsync(static_cast<T *>(out) + (size_t)row * n + start, dout + (size_t)row * m,
                                      sizeof(T) * (size_t)m, hipMemcpyDeviceToHost, st));
            int64_t bad = -1;
            r = read_bad(s, st, &bad);
            if (r == BSK_ERR_DOMAIN) {
                if (first_bad) *first_bad = start + bad;
                return r;
            }
            if (r != BSK_OK) return r;
        }
    }
    return BSK_OK;
}

extern "C" bsk_status bsk_normal(bsk_spline s, const void *const *uvw, int64_t n, bsk_mem mem, int normalize, int negate,
                                 void *out, void *stream, int64_t *first_bad)
{
    bsk_status r = check_call(s, uvw, n, out);
    if (r != BSK_OK) return r;
    if (std::abs(s->nInd - s->nDep) != 1)
        return fail(BSK_ERR_INVALID, "The number of independent variables must be one different than the number of dependent variables.");
    if (std::max(s->nInd, s->nDep) > 4) return fail(BSK_ERR_UNSUPPORTED, "normal supports max(nInd, nDep) <= 4");
    HIPCHK(hipSetDevice(s->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    return s->dtype == BSK_F32 ? run_normal<float>(s, uvw, n, mem, normalize, negate, out, st, first_bad)
                               : run_normal<double>(s, uvw, n, mem, normalize, negate, out, st, first_bad);
}

// ------------------------------------------------------------------------------------
// curvature
// ------------------------------------------------------------------------------------
template <typename T>
static bsk_status run_curvature(bsk_spline s, const void *const *uvw, long long n, bsk_mem mem, void *out, hipStream_t st,
                                int64_t *first_bad)
{
    if (first_bad) *first_bad = -1;
    if (n == 0) return BSK_OK;
    const bool surface = s->nInd == 2;
    const bool fused = surface && s->nDep == 3 && rowrot_applies<T>(s);   // curv_rowrot: no intermediates at all
    const int nbuf = fused ? 0 : (surface ? 6 : 2);      // derivative buffers (+ normal) of nDep rows each
    const long long chunk = std::min<long long>(n, mem == BSK_HOST ? host_chunk_points() : (1ll << 22));
    // derivative workspace: persistent on the handle (no allocation, free or synchronisation per call)
    if (bsk_status r_ = ws_reserve(s->curv_ws, std::max<size_t>(16, sizeof(T) * (size_t)chunk * s->nDep * nbuf), st); r_ != BSK_OK) return r_;
    T *w = static_cast<T *>(s->curv_ws.p);
    T *din = nullptr, *dout = static_cast<T *>(out);
    if (mem == BSK_HOST) {
        HIPCHK(s->in_ws.reserve(sizeof(T) * (size_t)chunk * s->nInd));
        HIPCHK(s->out_ws.reserve(sizeof(T) * (size_t)chunk));
        din = static_cast<T *>(s->in_ws.p);
    }
    const size_t one = (size_t)chunk * s->nDep;
    for (long long start = 0; start < n; start += chunk) {
        const long long m = std::min(chunk, n - start);
        Params<T> prm;
        for (int iv = 0; iv < MAXI; ++iv) prm.p[iv] = nullptr;
        for (int iv = 0; iv < s->nInd; ++iv) {
            if (mem == BSK_HOST) {
                HIPCHK(hipMemcpyAsync(din + (size_t)iv * m, static_cast<const T *>(uvw[iv]) + start, sizeof(T) * (size_t)m,
                                      hipMemcpyHostToDevice, st));
                prm.p[iv] = din + (size_t)iv * m;
            } else {
                prm.p[iv] = static_cast<const T *>(uvw[iv]) + start;
            }
        }
        T *o = mem == BSK_HOST ? static_cast<T *>(s->out_ws.p) : dout + start;
        auto deriv = [&](int w0, int w1, T *dst) -> bsk_status {
            Wrt wr;
            for (int iv = 0; iv < MAXI; ++iv) wr.w[iv] = 0;
            wr.w[0] = w0;
            wr.w[1] = w1;
            return dispatch_eval<T>(s, prm, m, dst, m, wr, st);
        };
        const int block = 256;
        const int grid = (int)std::max<long long>(1, std::min<long long>((m + block - 1) / block, (long long)s->num_cu * 8));
        bsk_status r;
        if (!surface) {
            if ((r = deriv(1, 0, w)) != BSK_OK) return r;
            if ((r = deriv(2, 0, w + one)) != BSK_OK) return r;
            hipLaunchKernelGGL((curvature_curve<T>), dim3(grid), dim3(block), 0, st, w, w + one, s->nDep, m, o);
        } else if (fused) {
            const Desc<T> &d = desc_of<T>(s);
            const TileDesc<T> &tdr = tile_of<T>(s);
            const size_t lds_rr = rowrot_lds_bytes<T>(s);
            const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(2, s->lds_max / lds_rr));
            const long long cmax = rr_chunk_points();
            for (long long c0 = 0; c0 < m; c0 += cmax) {
                const long long mm = std::min<long long>(m - c0, cmax);
                const int g = (int)std::max<long long>(1, std::min<long long>((mm + TILE - 1) / TILE, (long long)s->num_cu * per_cu));
                Params<T> cp = prm;
                for (int iv = 0; iv < s->nInd; ++iv) cp.p[iv] = prm.p[iv] + c0;
                if (s->order[0] == 4) {
                    HIPCHK(allow_lds(curv_rowrot<T, 4>, lds_rr));
                    s->last_kernel = "curv_rowrot";
                    hipLaunchKernelGGL((curv_rowrot<T, 4>), dim3(g), dim3(TILE), lds_rr, st, d, tdr, static_cast<const T *>(s->tab),
                                       s->lut, static_cast<const T *>(s->coef), cp, (unsigned)mm, c0, o + c0, s->bad);
                } else {
                    HIPCHK(allow_lds(curv_rowrot<T, 2>, lds_rr));
                    s->last_kernel = "curv_rowrot";
                    hipLaunchKernelGGL((curv_rowrot<T, 2>), dim3(g), dim3(TILE), lds_rr, st, d, tdr, static_cast<const T *>(s->tab),
                                       s->lut, static_cast<const T *>(s->coef), cp, (unsigned)mm, c0, o + c0, s->bad);
                }
            }
        } else {
            if ((r = deriv(1, 0, w)) != BSK_OK) return r;
            if ((r = deriv(0, 1, w + one)) != BSK_OK) return r;
            if ((r = deriv(2, 0, w + 2 * one)) != BSK_OK) return r;
            if ((r = deriv(1, 1, w + 3 * one)) != BSK_OK) return r;
            if ((r = deriv(0, 2, w + 4 * one)) != BSK_OK) return r;
            // unit normal (fused into the jacobian kernel when the image fits LDS)
            if (rowrot_applies<T>(s)) {
                if ((r = launch_jac_rowrot<T, true>(s, prm, m, w + 5 * one, 1, 0, st)) != BSK_OK) return r;
            } else {
                if (bsk_status r_ = ws_reserve(s->aux_ws, sizeof(T) * (size_t)m * s->nDep * s->nInd, st); r_ != BSK_OK) return r_;
                T *djac = static_cast<T *>(s->aux_ws.p);
                if ((r = dispatch_jac<T>(s, prm, m, djac, st)) != BSK_OK) return r;
                hipLaunchKernelGGL((normal_epilogue<T>), dim3(grid), dim3(block), 0, st, djac, 2, 3, m, 1, 0, w + 5 * one);
            }
            hipLaunchKernelGGL((curvature_surface<T>), dim3(grid), dim3(block), 0, st, w, w + one, w + 2 * one, w + 3 * one,
                               w + 4 * one, w + 5 * one, m, o);
        }
        HIPCHK(hipGetLastError());
        if (mem == BSK_HOST) {
            HIPCHK(hipMemcpyAsync(static_cast<T *>(out) + start, o, sizeof(T) * (size_t)m, hipMemcpyDeviceToHost, st));
            int64_t bad = -1;
            r = read_bad(s, st, &bad);
            if (r == BSK_ERR_DOMAIN) {
                if (first_bad) *first_bad = start + bad;
                return r;
            }
            if (r != BSK_OK) return r;
        }
    }
    return BSK_OK;
}

extern "C" bsk_status bsk_curvature(bsk_spline s, const void *const *uvw, int64_t n, bsk_mem mem, void *out, void *stream,
                                    int64_t *first_bad)
{
    bsk_status r = check_call(s, uvw, n, out);
    if (r != BSK_OK) return r;
    if (!((s->nInd == 1 && s->nDep >= 2) || (s->nInd == 2 && s->nDep == 3)))
        return fail(BSK_ERR_UNSUPPORTED, "curvature needs a curve with nDep >= 2 or a surface with nDep == 3");
    HIPCHK(hipSetDevice(s->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    return s->dtype == BSK_F32 ? run_curvature<float>(s, uvw, n, mem, out, st, first_bad)
                               : run_curvature<double>(s, uvw, n, mem, out, st, first_bad);
}

// ------------------------------------------------------------------------------------
// tensor-product grid
// ------------------------------------------------------------------------------------
template <typename T>
static bsk_status run_grid(bsk_spline s, const int *wrt, const void *const *grid, const int64_t *ngrid, bsk_mem mem,
                           void *out, hipStream_t st, int64_t *first_bad)
{
    const Desc<T> &d = desc_of<T>(s);
    if (first_bad) *first_bad = -1;
    GridDims g;
    long long total = 1, npar = 0, nrow = 0;
    for (int iv = 0; iv < MAXI; ++iv) { g.n[iv] = 1; g.goff[iv] = 0; g.roff[iv] = 0; }
    for (int iv = 0; iv < s->nInd; ++iv) {
        if (ngrid[iv] < 0) return fail(BSK_ERR_INVALID, "negative grid size");
        if (wrt && wrt[iv] < 0) return fail(BSK_ERR_INVALID, "negative derivative order");
        g.n[iv] = ngrid[iv];
        g.goff[iv] = npar;
        g.roff[iv] = nrow;
        npar += ngrid[iv];
        nrow += ngrid[iv] * s->order[iv];
        total *= ngrid[iv];
    }
    if (total == 0) return BSK_OK;

    // aux workspace: [params T x npar (host mode only)] [rows T x nrow] [ix int x npar] [outside u8 x npar]
    const size_t par_b = mem == BSK_HOST ? ((sizeof(T) * (size_t)npar + 15) & ~(size_t)15) : 0;
    const size_t row_b = (sizeof(T) * (size_t)nrow + 15) & ~(size_t)15;
    const size_t ix_b = (sizeof(int) * (size_t)npar + 15) & ~(size_t)15;
    if (bsk_status r_ = ws_reserve(s->aux_ws, par_b + row_b + ix_b + (size_t)npar + 16, st); r_ != BSK_OK) return r_;
    char *base = static_cast<char *>(s->aux_ws.p);
    T *dpar = reinterpret_cast<T *>(base);
    T *rows = reinterpret_cast<T *>(base + par_b);
    int *ixs = reinterpret_cast<int *>(base + par_b + row_b);
    unsigned char *outside = reinterpret_cast<unsigned char *>(base + par_b + row_b + ix_b);

    GridAxes<T> ax;
    long long nmax = 1;
    for (int iv = 0; iv < MAXI; ++iv) { ax.u[iv] = nullptr; ax.wrt[iv] = 0; }
    for (int iv = 0; iv < s->nInd; ++iv) {
        const T *u = static_cast<const T *>(grid[iv]);
        if (mem == BSK_HOST) {
            HIPCHK(hipMemcpyAsync(dpar + g.goff[iv], u, sizeof(T) * (size_t)ngrid[iv], hipMemcpyHostToDevice, st));
            u = dpar + g.goff[iv];
        }
        ax.u[iv] = u;
        ax.wrt[iv] = wrt ? wrt[iv] : 0;
        nmax = std::max<long long>(nmax, ngrid[iv]);
    }
    hipLaunchKernelGGL((basis_rows_grid<T>), dim3((unsigned)((nmax + 255) / 256), (unsigned)s->nInd), dim3(256), 0, st, d,
                       static_cast<const T *>(s->tab), ax, g, ixs, rows, outside);
    HIPCHK(hipGetLastError());

    T *dout = static_cast<T *>(out);
    if (mem == BSK_HOST) {
        HIPCHK(s->out_ws.reserve(sizeof(T) * (size_t)total * s->nDep));
        dout = static_cast<T *>(s->out_ws.p);
    }
    const T *coef = static_cast<const T *>(s->coef);
    bool launched = false;
    const size_t rowc_bytes = sizeof(T) * (size_t)s->nDep * (size_t)(s->nInd == 2 ? s->ncoef[1] : 0);
    const int omax2 = s->nInd == 2 ? std::max(s->order[0], s->order[1]) : 0;
    if (s->nInd == 2 && omax2 <= 6 && rowc_bytes <= 48 * 1024 && g.n[1] >= 64 && s->variant != 1) {
        // row-factored surface grid
        constexpr long long VEC = 16 / (long long)sizeof(T);
        const int vec_ok = (g.n[1] % VEC == 0) && ((reinterpret_cast<uintptr_t>(dout) & 15) == 0) ? 1 : 0;
        const int gridx = (int)std::max<long long>(1, std::min<long long>(g.n[0], (long long)s->num_cu * 8));
#define GRID_ROWS(O)                                                                                           \
    case O:                                                                                                    \
        if (s->same_order)                                                                                     \
            hipLaunchKernelGGL((grid_rows<T, O, false>), dim3(gridx), dim3(256), rowc_bytes, st, d, coef, g, ixs, rows, \
                               outside, dout, s->bad, vec_ok);                                                 \
        else                                                                                                   \
            hipLaunchKernelGGL((grid_rows<T, O, true>), dim3(gridx), dim3(256), rowc_bytes, st, d, coef, g, ixs, rows,  \
                               outside, dout, s->bad, vec_ok);                                                 \
        launched = true;                                                                                       \
        break;
        switch (omax2) {
            GRID_ROWS(1) GRID_ROWS(2) GRID_ROWS(3) GRID_ROWS(4) GRID_ROWS(5) GRID_ROWS(6)
            default: break;
        }
#undef GRID_ROWS
    }
    if (!launched && s->nInd == 2 && s->same_order && g.n[0] <= 65535) {
        const int block = 256;
        const int gx = (int)std::max<long long>(1, std::min<long long>((g.n[1] + block - 1) / block, 64));
#define GRID_SURF(O)                                                                                           \
    case O:                                                                                                    \
        hipLaunchKernelGGL((grid_surface<T, O>), dim3(gx, (unsigned)g.n[0]), dim3(block), 0, st, d, coef, g,   \
                           ixs, rows, outside, dout, s->bad);                                                  \
        launched = true;                                                                                       \
        break;
        switch (s->order[0]) {
            GRID_SURF(1) GRID_SURF(2) GRID_SURF(3) GRID_SURF(4) GRID_SURF(5) GRID_SURF(6)
            default: break;
        }
#undef GRID_SURF
    }
    if (!launched) {
        const int block = 256;
        const long long blocks = (total + block - 1) / block;
        const int gridx = (int)std::max<long long>(1, std::min<long long>(blocks, (long long)s->num_cu * 8));
        hipLaunchKernelGGL((grid_generic<T>), dim3(gridx), dim3(block), 0, st, d, coef, g, ixs, rows, outside, total,
                           dout, s->bad);
    }
    HIPCHK(hipGetLastError());
    if (mem == BSK_HOST) {
        HIPCHK(hipMemcpyAsync(out, dout, sizeof(T) * (size_t)total * s->nDep, hipMemcpyDeviceToHost, st));
        return read_bad(s, st, first_bad);
    }
    return BSK_OK;
}

extern "C" bsk_status bsk_evaluate_grid(bsk_spline s, const int *wrt, const void *const *grid, const int64_t *ngrid,
                                        bsk_mem mem, void *out, void *stream, int64_t *first_bad)
{
    if (!s || !grid || !ngrid || !out) return fail(BSK_ERR_INVALID, "NULL argument");
    for (int iv = 0; iv < s->nInd; ++iv)
        if (!grid[iv] && ngrid[iv] > 0) return fail(BSK_ERR_INVALID, "NULL grid pointer");
    HIPCHK(hipSetDevice(s->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    return s->dtype == BSK_F32 ? run_grid<float>(s, wrt, grid, ngrid, mem, out, st, first_bad)
                               : run_grid<double>(s, wrt, grid, ngrid, mem, out, st, first_bad);
}

// ------------------------------------------------------------------------------------
// tessellation of a batch of patches (positions + normals), SURVEY 8f-2
// ------------------------------------------------------------------------------------
template <typename T>
static bsk_status run_tessellate(const bsk_spline *sp, int count, const void *const *grid, const int64_t *ngrid,
                                 bsk_mem mem, int normalize, int negate, void *positions, void *normals, hipStream_t st,
                                 int64_t *first_bad)
{
    bsk_spline s = sp[0];
    const Desc<T> &d = desc_of<T>(s);
    if (first_bad) *first_bad = -1;
    GridDims g;
    long long npar = 0, nrow = 0;
    for (int iv = 0; iv < MAXI; ++iv) { g.n[iv] = 1; g.goff[iv] = 0; g.roff[iv] = 0; }
    for (int iv = 0; iv < 2; ++iv) {
        if (ngrid[iv] < 0) return fail(BSK_ERR_INVALID, "negative grid size");
        g.n[iv] = ngrid[iv];
        g.goff[iv] = npar;
        g.roff[iv] = nrow;
        npar += ngrid[iv];
        nrow += ngrid[iv] * s->order[iv];
    }
    const long long total = g.n[0] * g.n[1];
    if (total == 0 || count == 0) return BSK_OK;

    // aux workspace of the first patch: [params (host mode)] [value rows] [derivative rows] [ix] [outside]
    const size_t par_b = mem == BSK_HOST ? ((sizeof(T) * (size_t)npar + 15) & ~(size_t)15) : 0;
    const size_t row_b = (sizeof(T) * (size_t)nrow + 15) & ~(size_t)15;
    const size_t ix_b = (sizeof(int) * (size_t)npar + 15) & ~(size_t)15;
    if (bsk_status r_ = ws_reserve(s->aux_ws, par_b + 2 * row_b + 2 * ix_b + 2 * (size_t)npar + 32, st); r_ != BSK_OK) return r_;
    char *base = static_cast<char *>(s->aux_ws.p);
    T *dpar = reinterpret_cast<T *>(base);
    T *rows = reinterpret_cast<T *>(base + par_b);
    T *drows = reinterpret_cast<T *>(base + par_b + row_b);
    int *ixs = reinterpret_cast<int *>(base + par_b + 2 * row_b);
    int *ixs2 = reinterpret_cast<int *>(base + par_b + 2 * row_b + ix_b);
    unsigned char *outside = reinterpret_cast<unsigned char *>(base + par_b + 2 * row_b + 2 * ix_b);
    unsigned char *outside2 = outside + ((npar + 15) & ~15ll);

    GridAxes<T> ax;
    long long nmax = 1;
    for (int iv = 0; iv < MAXI; ++iv) { ax.u[iv] = nullptr; ax.wrt[iv] = 0; }
    for (int iv = 0; iv < 2; ++iv) {
        const T *u = static_cast<const T *>(grid[iv]);
        if (mem == BSK_HOST) {
            HIPCHK(hipMemcpyAsync(dpar + g.goff[iv], u, sizeof(T) * (size_t)ngrid[iv], hipMemcpyHostToDevice, st));
            u = dpar + g.goff[iv];
        }
        ax.u[iv] = u;
        nmax = std::max<long long>(nmax, ngrid[iv]);
    }
    const T *tab = static_cast<const T *>(s->tab);
    hipLaunchKernelGGL((basis_rows_grid<T>), dim3((unsigned)((nmax + 255) / 256), 2u), dim3(256), 0, st, d, tab, ax, g, ixs,
                       rows, outside);
    if (normals) {
        ax.wrt[0] = ax.wrt[1] = 1;                            // first-derivative rows of both variables
        hipLaunchKernelGGL((basis_rows_grid<T>), dim3((unsigned)((nmax + 255) / 256), 2u), dim3(256), 0, st, d, tab, ax, g,
                           ixs2, drows, outside2);
    }
    HIPCHK(hipGetLastError());

    T *dpos = static_cast<T *>(positions), *dnrm = static_cast<T *>(normals);
    const size_t plane = sizeof(T) * (size_t)total * 3 * (size_t)count;
    if (mem == BSK_HOST) {
        HIPCHK(s->out_ws.reserve(plane * (normals ? 2 : 1)));
        dpos = static_cast<T *>(s->out_ws.p);
        dnrm = normals ? dpos + (size_t)total * 3 * (size_t)count : nullptr;
    }
    constexpr long long VEC = 16 / (long long)sizeof(T);
    const int vec_ok = (g.n[1] % VEC == 0) && ((reinterpret_cast<uintptr_t>(dpos) & 15) == 0) &&
                       ((reinterpret_cast<uintptr_t>(dnrm) & 15) == 0) ? 1 : 0;
    const int tess_r = getenv("BSK_TESS_R") ? std::max(1, std::min(64, atoi(getenv("BSK_TESS_R")))) : TESS_R;      // measurement knobs
    const int tess_t = getenv("BSK_TESS_T") ? atoi(getenv("BSK_TESS_T")) : 512;
    const int tess_w = getenv("BSK_TESS_W") ? std::max(1, atoi(getenv("BSK_TESS_W"))) : 0;                          // 256-lane workgroups per CU (0: 8, or 2 for the 512-lane form = one workgroup per CU: the fewer streams are written at a time, the better the memory side does - 0.287 / 0.255 / 0.239 ms at 8 / 4 / 2)
    const size_t lds = sizeof(T) * 3 * (size_t)s->ncoef[1] * std::max(2, tess_r);     // contracted rows: [2] with normals, [tess_r] rows without
    for (int p0 = 0; p0 < count; p0 += TESS_MAX_PATCHES) {
        const int np = std::min(TESS_MAX_PATCHES, count - p0);
        PatchCoefs<T> pc;
        for (int i = 0; i < TESS_MAX_PATCHES; ++i) pc.c[i] = i < np ? static_cast<const T *>(sp[p0 + i]->coef) : nullptr;
        // (the positions-only form with hoisted column bases takes TESS_R grid rows per workgroup iteration: same test as in the kernel)
        const bool hoist = !normals && vec_ok && s->same_order && g.n[1] <= 256 * VEC * 2;
        const bool wide = hoist && tess_t == 512 && g.n[1] <= 512 * VEC;       // 512 lanes, one hoisted step per lane
        const long long row_units = hoist ? (g.n[0] + tess_r - 1) / tess_r : g.n[0];
        const int gx = (int)std::max<long long>(1, std::min<long long>(row_units, std::max<long long>(1, (long long)s->num_cu * (tess_w ? tess_w : wide ? 2 : 8) / (wide ? 2 : 1) / np)));
        T *pp = dpos + (size_t)p0 * 3 * (size_t)total;
        T *pn = dnrm ? dnrm + (size_t)p0 * 3 * (size_t)total : nullptr;
#define TESS_LAUNCH(O, NRM, MIX)                                                                                       \
    hipLaunchKernelGGL((tess_rows<T, O, NRM, MIX>), dim3(gx, np), dim3(256), lds, st, d, pc, g, ixs, rows,             \
                       NRM ? drows : rows, outside, pp, pn, s->bad, vec_ok, normalize, negate, tess_r)
#define TESS_LAUNCH_WIDE(O)                                                                                            \
    hipLaunchKernelGGL((tess_rows<T, O, false, false, 1>), dim3(gx, np), dim3(512), lds, st, d, pc, g, ixs, rows,      \
                       rows, outside, pp, pn, s->bad, vec_ok, normalize, negate, tess_r)
#define TESS(O)                                                                                                        \
    case O:                                                                                                            \
        if (s->same_order) { if (normals) TESS_LAUNCH(O, true, false); else if (wide) TESS_LAUNCH_WIDE(O); else TESS_LAUNCH(O, false, false); } \
        else { if (normals) TESS_LAUNCH(O, true, true); else TESS_LAUNCH(O, false, true); }                            \
        break;
        switch (std::max(s->order[0], s->order[1])) {
            TESS(1) TESS(2) TESS(3) TESS(4) TESS(5) TESS(6)
            default: return fail(BSK_ERR_UNSUPPORTED, "bsk_tessellate: orders 1..6");
        }
#undef TESS
#undef TESS_LAUNCH
#undef TESS_LAUNCH_WIDE
    }
    HIPCHK(hipGetLastError());
    if (mem == BSK_HOST) {
        HIPCHK(hipMemcpyAsync(positions, dpos, plane, hipMemcpyDeviceToHost, st));
        if (normals) HIPCHK(hipMemcpyAsync(normals, dnrm, plane, hipMemcpyDeviceToHost, st));
        return read_bad(s, st, first_bad);
    }
    return BSK_OK;
}

extern "C" bsk_status bsk_tessellate(const bsk_spline *splines, int count, const void *const *grid, const int64_t *ngrid,
                                     bsk_mem mem, int normalize, int negate, void *positions, void *normals, void *stream,
                                     int64_t *first_bad)
{
    if (count < 0) return fail(BSK_ERR_INVALID, "negative patch count");
    if (count == 0) return BSK_OK;
    if (!splines || !grid || !ngrid || !positions) return fail(BSK_ERR_INVALID, "NULL argument");
    bsk_spline s = splines[0];
    if (!s) return fail(BSK_ERR_INVALID, "NULL handle");
    if (s->nInd != 2 || s->nDep != 3) return fail(BSK_ERR_UNSUPPORTED, "bsk_tessellate: surfaces in 3-D (nInd 2, nDep 3)");
    if (std::max(s->order[0], s->order[1]) > 6 || sizeof(double) * 6 * (size_t)s->ncoef[1] > 60 * 1024)
        return fail(BSK_ERR_UNSUPPORTED, "bsk_tessellate: orders <= 6, nCoef[1] <= 1280");
    for (int i = 1; i < count; ++i) {
        bsk_spline t = splines[i];
        if (!t) return fail(BSK_ERR_INVALID, "NULL handle");
        if (t->dtype != s->dtype || t->device != s->device || t->nInd != 2 || t->nDep != 3 || t->order[0] != s->order[0] ||
            t->order[1] != s->order[1] || t->ncoef[0] != s->ncoef[0] || t->ncoef[1] != s->ncoef[1] ||
            t->tab_host != s->tab_host)
            return fail(BSK_ERR_INVALID, "bsk_tessellate: the patches of a batch must share dtype, device, orders, nCoef and knots");
    }
    for (int iv = 0; iv < 2; ++iv)
        if (!grid[iv] && ngrid[iv] > 0) return fail(BSK_ERR_INVALID, "NULL grid pointer");
    HIPCHK(hipSetDevice(s->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    return s->dtype == BSK_F32
               ? run_tessellate<float>(splines, count, grid, ngrid, mem, normalize, negate, positions, normals, st, first_bad)
               : run_tessellate<double>(splines, count, grid, ngrid, mem, normalize, negate, positions, normals, st, first_bad);
}

// ------------------------------------------------------------------------------------
// batched bspline_values
// ------------------------------------------------------------------------------------
template <typename T>
static bsk_status run_basis(int device, const T *knots, int nknots, int order, const T *u, long long n, int deriv,
                            int taylor, const int32_t *knot_in, int32_t *ix_out, T *basis_out)
{
    HIPCHK(hipSetDevice(device));
    std::vector<T> tab;
    build_axis_table<T>(knots, order, nknots, tab);
    const int ncoef = nknots - order;
    if (n <= small_call_points()) {
        // small calls (the reference's static Spline.bspline_values is one point): everything travels
        // through one pinned, device-mapped buffer - no allocation, no staging copy, one synchronisation
        // per-thread scratch, returned when the thread ends (found by the AddressSanitizer run of the host
        // code: the bare thread_local pointers leaked one pinned and one device buffer per calling thread)
        struct Scratch {
            void *pin = nullptr;
            size_t pin_cap = 0;
            DevBuf dtab;
            ~Scratch()
            {
                if (pin) (void)hipHostFree(pin);
                dtab.release();
            }
        };
        static thread_local Scratch scratch;
        void *&pin = scratch.pin;
        size_t &pin_cap = scratch.pin_cap;
        DevBuf &dtab_small = scratch.dtab;
        auto up = [](size_t b) { return (b + 63) & ~(size_t)63; };
        const size_t o_tab = 0, o_u = o_tab + up(sizeof(T) * tab.size()), o_k = o_u + up(sizeof(T) * (size_t)n);
        const size_t o_ix = o_k + up(sizeof(int) * (size_t)n), o_b = o_ix + up(sizeof(int) * (size_t)n);
        const size_t total = o_b + up(sizeof(T) * (size_t)n * order);
        if (total > pin_cap) {
            if (pin) (void)hipHostFree(pin);
            pin = nullptr;
            pin_cap = 0;
            HIPCHK(hipHostMalloc(&pin, total, hipHostMallocMapped | hipHostMallocPortable));
            pin_cap = total;
        }
        char *hp = static_cast<char *>(pin);
        void *dpv = nullptr;
        HIPCHK(hipHostGetDevicePointer(&dpv, pin, 0));
        char *dp = static_cast<char *>(dpv);
        memcpy(hp + o_tab, tab.data(), sizeof(T) * tab.size());
        memcpy(hp + o_u, u, sizeof(T) * (size_t)n);
        if (knot_in) memcpy(hp + o_k, knot_in, sizeof(int) * (size_t)n);
        // the axis table is read many times with dependent accesses (span search, recursion): it goes to
        // device memory (one small DMA from the pinned buffer); parameters and results stay zero-copy
        HIPCHK(dtab_small.reserve(sizeof(T) * tab.size()));
        HIPCHK(hipMemcpyAsync(dtab_small.p, hp + o_tab, sizeof(T) * tab.size(), hipMemcpyHostToDevice, 0));
        const int blocks = (int)((n + 255) / 256);
        hipLaunchKernelGGL((basis_rows<T>), dim3(blocks), dim3(256), 0, 0, static_cast<const T *>(dtab_small.p), nknots,
                           order, ncoef, ceil_log2(ncoef - order + 1), reinterpret_cast<const T *>(dp + o_u), n, deriv, taylor,
                           knot_in ? reinterpret_cast<const int *>(dp + o_k) : nullptr, reinterpret_cast<int *>(dp + o_ix),
                           reinterpret_cast<T *>(dp + o_b), T(0), T(0), (unsigned char *)nullptr);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(0));
        memcpy(ix_out, hp + o_ix, sizeof(int) * (size_t)n);
        memcpy(basis_out, hp + o_b, sizeof(T) * (size_t)n * order);
        return BSK_OK;
    }
    DevBuf dtab, du, dk, dix, db;
    auto release = [&]() { dtab.release(); du.release(); dk.release(); dix.release(); db.release(); };
#define HIPCHK_R(expr)                                                                        \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) {                                                               \
            release();                                                                        \
            return fail(BSK_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));      \
        }                                                                                     \
    } while (0)
    HIPCHK_R(dtab.reserve(sizeof(T) * tab.size()));
    HIPCHK_R(du.reserve(sizeof(T) * (size_t)n));
    HIPCHK_R(dix.reserve(sizeof(int) * (size_t)n));
    HIPCHK_R(db.reserve(sizeof(T) * (size_t)n * order));
    HIPCHK_R(hipMemcpy(dtab.p, tab.data(), sizeof(T) * tab.size(), hipMemcpyHostToDevice));
    HIPCHK_R(hipMemcpy(du.p, u, sizeof(T) * (size_t)n, hipMemcpyHostToDevice));
    if (knot_in) {
        HIPCHK_R(dk.reserve(sizeof(int) * (size_t)n));
        HIPCHK_R(hipMemcpy(dk.p, knot_in, sizeof(int) * (size_t)n, hipMemcpyHostToDevice));
    }
    const int block = 256;
    const int blocks = (int)((n + block - 1) / block);
    hipLaunchKernelGGL((basis_rows<T>), dim3(blocks), dim3(block), 0, 0, static_cast<const T *>(dtab.p), nknots, order,
                       ncoef, ceil_log2(ncoef - order + 1), static_cast<const T *>(du.p), n, deriv, taylor,
                       static_cast<const int *>(dk.p), static_cast<int *>(dix.p), static_cast<T *>(db.p), T(0), T(0),
                       (unsigned char *)nullptr);
    HIPCHK_R(hipGetLastError());
    HIPCHK_R(hipMemcpy(ix_out, dix.p, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost));
    HIPCHK_R(hipMemcpy(basis_out, db.p, sizeof(T) * (size_t)n * order, hipMemcpyDeviceToHost));
#undef HIPCHK_R
    release();
    return BSK_OK;
}

extern "C" bsk_status bsk_bspline_values(bsk_dtype dtype, int device, const void *knots, int nknots, int order,
                                         const void *u, int64_t n, int derivative_order, int taylor_coefs,
                                         const int32_t *knot_in, int32_t *ix_out, void *basis_out)
{
    if (!knots || !u || !ix_out || !basis_out) return fail(BSK_ERR_INVALID, "NULL argument");
    if (order < 1 || order > MAXO) return fail(BSK_ERR_UNSUPPORTED, "order must be in [1, BSK_MAX_ORDER]");
    if (nknots < 2 * order) return fail(BSK_ERR_INVALID, "need at least 2 * order knots");
    if (derivative_order < 0) return fail(BSK_ERR_INVALID, "negative derivative order");
    if (n < 0) return fail(BSK_ERR_INVALID, "negative count");
    if (n == 0) return BSK_OK;
    if (knot_in)
        for (int64_t i = 0; i < n; ++i)
            if (knot_in[i] < order || knot_in[i] > nknots - order)
                return fail(BSK_ERR_INVALID, "explicit knot index outside [order, len(knots) - order]");
    if (dtype == BSK_F32)
        return run_basis<float>(device, static_cast<const float *>(knots), nknots, order, static_cast<const float *>(u),
                                n, derivative_order, taylor_coefs, knot_in, ix_out, static_cast<float *>(basis_out));
    if (dtype == BSK_F64)
        return run_basis<double>(device, static_cast<const double *>(knots), nknots, order,
                                 static_cast<const double *>(u), n, derivative_order, taylor_coefs, knot_in, ix_out,
                                 static_cast<double *>(basis_out));
    return fail(BSK_ERR_INVALID, "dtype must be BSK_F32 or BSK_F64");
}

// ------------------------------------------------------------------------------------
// diagnostics
// ------------------------------------------------------------------------------------
// Streams u, v -> out (3 rows) with the evaluation kernels' launch geometry; mode 0/1 = 8/16
// bytes per lane, blocks_per_cu workgroups per CU, lds_bytes of dynamic LDS each.  Device
// pointers, fp64.  Used by tools/ to measure the memory-side floor; not an evaluation call.
extern "C" const char *bsk_last_kernel(bsk_spline s) { return s ? s->last_kernel : ""; }

// Per-kernel times of the most recent multi-kernel pipeline call on this handle (the cell-order pipeline): `enable`
// switches the recording of an event behind every kernel on or off for the FOLLOWING calls; with ms / names it
// returns the durations of the last recorded call (after synchronising its last event).
extern "C" bsk_status bsk_debug_stage_times(bsk_spline s, int enable, float *ms, const char **names, int cap, int *count)
{
    if (!s) return fail(BSK_ERR_INVALID, "spline is NULL");
    int n = 0;
    if (ms && names && count && s->stage_count > 1) {
        HIPCHK(hipSetDevice(s->device));
        HIPCHK(hipEventSynchronize(s->stage_ev[s->stage_count - 1]));
        for (int i = 0; i + 1 < s->stage_count && n < cap; ++i, ++n) {
            HIPCHK(hipEventElapsedTime(&ms[n], s->stage_ev[i], s->stage_ev[i + 1]));
            names[n] = s->stage_name[i + 1];
        }
    }
    if (count) *count = n;
    s->stage_timing = enable != 0;
    if (!enable) s->stage_count = 0;
    return BSK_OK;
}

extern "C" bsk_status bsk_debug_probe(bsk_spline s, int mode, int blocks_per_cu, int threads, int64_t lds_bytes,
                                      const void *u, const void *v, int64_t n, void *out, void *stream)
{
    if (!s || !u || !v || !out) return fail(BSK_ERR_INVALID, "NULL argument");
    HIPCHK(hipSetDevice(s->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int grid = s->num_cu * std::max(1, blocks_per_cu);
    if (mode == 0) {
        HIPCHK(allow_lds(probe_stream<0>, (size_t)lds_bytes));
        hipLaunchKernelGGL((probe_stream<0>), dim3(grid), dim3(threads), (size_t)lds_bytes, st, static_cast<const double *>(u),
                           static_cast<const double *>(v), (long long)n, static_cast<double *>(out), (long long)n);
    } else {
        HIPCHK(allow_lds(probe_stream<1>, (size_t)lds_bytes));
        hipLaunchKernelGGL((probe_stream<1>), dim3(grid), dim3(threads), (size_t)lds_bytes, st, static_cast<const double *>(u),
                           static_cast<const double *>(v), (long long)n, static_cast<double *>(out), (long long)n);
    }
    HIPCHK(hipGetLastError());
    return BSK_OK;
}
