// Multi-device entry points of the C ABI (include/bspy_amd.h: bsk_multi_*): one process drives
// several GPUs of a node.  The point batch is cut into contiguous shards (ceil(n / ndev) points per
// device), every device evaluates its shard with its own replica of the spline tables on its own
// stream through the single-device entry points, and - only when the caller asks for the whole
// result on every device - the shards are exchanged with ONE grouped RCCL all-gather per call
// (ncclGroupStart / one ncclAllGather per device and output row / ncclGroupEnd: SoA rows land in
// place, no repacking pass).  Points are independent: there is no other collective.
//
// librccl.so is opened on first use (dlopen): single-GPU users never load it, and the library has
// no link-time dependency on it.  The reference has no multi-device code at all; this
// is the north_star's "sharding the evaluation-point batch with an RCCL all-gather of results".
#include "bsk_host.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>

namespace {

struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

Rccl *rccl()
{
    static Rccl r;
    static bool tried = false;
    if (!tried) {
        tried = true;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.lib) break;
        }
        if (r.lib) {
            r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(dlsym(r.lib, "ncclCommInitAll"));
            r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.lib, "ncclCommDestroy"));
            r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(r.lib, "ncclAllGather"));
            r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(dlsym(r.lib, "ncclGroupStart"));
            r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(dlsym(r.lib, "ncclGroupEnd"));
            r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.lib, "ncclGetErrorString"));
            if (!r.CommInitAll || !r.CommDestroy || !r.AllGather || !r.GroupStart || !r.GroupEnd || !r.GetErrorString) {
                dlclose(r.lib);
                r.lib = nullptr;
            }
        }
    }
    return r.lib ? &r : nullptr;
}

}  // namespace

struct bsk_multi_s {
    int ndev = 0;
    bsk_dtype dtype = BSK_F64;
    int nInd = 0, nDep = 0;
    size_t esize = 8;
    std::vector<int> dev;
    std::vector<bsk_spline> h;           // one replica of the tables per device
    std::vector<hipStream_t> st;         // one stream per device
    std::vector<DevBuf> in_ws, out_ws;   // BSK_HOST staging per device
    std::vector<ncclComm_t> comm;        // created by the first gathering call
};

#define NCCLCHK(expr)                                                                              \
    do {                                                                                           \
        ncclResult_t r_ = (expr);                                                                  \
        if (r_ != ncclSuccess) return fail(BSK_ERR_HIP, std::string(#expr) + ": " + R->GetErrorString(r_)); \
    } while (0)

extern "C" bsk_status bsk_multi_destroy(bsk_multi m)
{
    if (!m) return BSK_OK;
    for (int d = 0; d < (int)m->h.size(); ++d) {
        (void)hipSetDevice(m->dev[d]);
        if (d < (int)m->st.size() && m->st[d]) { (void)hipStreamSynchronize(m->st[d]); }
    }
    if (!m->comm.empty()) {
        if (Rccl *R = rccl()) for (ncclComm_t c : m->comm) if (c) (void)R->CommDestroy(c);
    }
    for (int d = 0; d < (int)m->h.size(); ++d) {
        (void)hipSetDevice(m->dev[d]);
        if (d < (int)m->in_ws.size()) { m->in_ws[d].release(); m->out_ws[d].release(); }
        if (d < (int)m->st.size() && m->st[d]) (void)hipStreamDestroy(m->st[d]);
        if (m->h[d]) (void)bsk_spline_destroy(m->h[d]);
    }
    delete m;
    return BSK_OK;
}

extern "C" bsk_status bsk_multi_create(bsk_dtype dtype, int ndev, const int *devices, int nInd, int nDep, const int *order,
                                       const int *nCoef, const void *const *knots, const void *coefs, bsk_multi *out)
{
    if (!out) return fail(BSK_ERR_INVALID, "out is NULL");
    *out = nullptr;
    int have = 0;
    if (hipGetDeviceCount(&have) != hipSuccess || have < 1) return fail(BSK_ERR_NO_DEVICE, "no HIP device");
    if (ndev < 1 || ndev > have) return fail(BSK_ERR_INVALID, "ndev must be in [1, device count]");
    bsk_multi m = new bsk_multi_s();
    m->ndev = ndev;
    m->dtype = dtype;
    m->nInd = nInd;
    m->nDep = nDep;
    m->esize = dtype == BSK_F32 ? 4 : 8;
    m->in_ws.resize(ndev);
    m->out_ws.resize(ndev);
    for (int d = 0; d < ndev; ++d) {
        const int id = devices ? devices[d] : d;
        if (id < 0 || id >= have) { bsk_multi_destroy(m); return fail(BSK_ERR_INVALID, "device index out of range"); }
        for (int e = 0; e < d; ++e)
            if (m->dev[e] == id) { bsk_multi_destroy(m); return fail(BSK_ERR_INVALID, "device listed twice"); }
        m->dev.push_back(id);
        m->h.push_back(nullptr);
        m->st.push_back(nullptr);
        bsk_status r = bsk_spline_create(dtype, id, nInd, nDep, order, nCoef, knots, coefs, &m->h[d]);
        if (r != BSK_OK) { const std::string msg = g_err; bsk_multi_destroy(m); return fail(r, msg); }
        if (hipSetDevice(id) != hipSuccess || hipStreamCreateWithFlags(&m->st[d], hipStreamNonBlocking) != hipSuccess) {
            bsk_multi_destroy(m);
            return fail(BSK_ERR_HIP, "hipStreamCreate failed");
        }
    }
    *out = m;
    return BSK_OK;
}

extern "C" bsk_status bsk_multi_shard_plan(bsk_multi m, int64_t n, int64_t *start)
{
    if (!m || !start || n < 0) return fail(BSK_ERR_INVALID, "bad argument");
    const int64_t chunk = n > 0 ? (n + m->ndev - 1) / m->ndev : 0;
    for (int d = 0; d <= m->ndev; ++d) start[d] = std::min<int64_t>((int64_t)d * chunk, n);
    return BSK_OK;
}

extern "C" bsk_status bsk_multi_stream(bsk_multi m, int d, void **stream)
{
    if (!m || !stream || d < 0 || d >= m->ndev) return fail(BSK_ERR_INVALID, "bad argument");
    *stream = m->st[d];
    return BSK_OK;
}

static bsk_status multi_comm(bsk_multi m, Rccl *&R)
{
    R = rccl();
    if (!R) return fail(BSK_ERR_UNSUPPORTED, "librccl.so could not be loaded: gather needs RCCL");
    if (m->comm.empty()) {
        m->comm.assign(m->ndev, nullptr);
        NCCLCHK(R->CommInitAll(m->comm.data(), m->ndev, m->dev.data()));
    }
    return BSK_OK;
}

// One sharded call: jacobian ? bsk_jacobian : bsk_evaluate(wrt) on every device's shard.
static bsk_status multi_run(bsk_multi m, bool jac, const int *wrt, const void *const *uvw, int64_t n, bsk_mem mem, void *const *out,
                            int gather, int64_t *first_bad)
{
    if (first_bad) *first_bad = -1;
    if (!m || !uvw || !out || n < 0) return fail(BSK_ERR_INVALID, "bad argument");
    const int rows = jac ? m->nDep * m->nInd : m->nDep;
    const size_t es = m->esize;
    const int64_t chunk = n > 0 ? (n + m->ndev - 1) / m->ndev : 0;
    Rccl *R = nullptr;
    if (mem == BSK_DEVICE && gather) {
        const bsk_status r = multi_comm(m, R);
        if (r != BSK_OK) return r;
    }
    std::vector<const void *> ptrs((size_t)std::max(1, m->nInd));
    // enqueue every device's shard (copies in, kernel, copies out) on its own stream
    for (int d = 0; d < m->ndev; ++d) {
        const int64_t s0 = std::min<int64_t>((int64_t)d * chunk, n), cnt = std::min<int64_t>(s0 + chunk, n) - s0;
        HIPCHK(hipSetDevice(m->dev[d]));
        char *dout;
        if (mem == BSK_HOST) {
            HIPCHK(m->in_ws[d].reserve(std::max<size_t>(16, es * (size_t)cnt * m->nInd)));
            HIPCHK(m->out_ws[d].reserve(std::max<size_t>(16, es * (size_t)cnt * rows)));
            char *din = static_cast<char *>(m->in_ws[d].p);
            dout = static_cast<char *>(m->out_ws[d].p);
            for (int iv = 0; iv < m->nInd; ++iv) {
                if (cnt)
                    HIPCHK(hipMemcpyAsync(din + es * (size_t)iv * cnt, static_cast<const char *>(uvw[iv]) + es * (size_t)s0,
                                          es * (size_t)cnt, hipMemcpyHostToDevice, m->st[d]));
                ptrs[iv] = din + es * (size_t)iv * cnt;
            }
        } else {
            // device buffers: uvw[d * nInd + iv] = device d's shard of variable iv; with gather the results
            // are written at this device's slot of its own full-size buffer out[d] ((rows, ndev * chunk))
            for (int iv = 0; iv < m->nInd; ++iv) ptrs[iv] = uvw[(size_t)d * m->nInd + iv];
            dout = static_cast<char *>(out[d]);
            if (!dout) return fail(BSK_ERR_INVALID, "NULL output pointer");
        }
        // gathered calls: the single-device entry points write a compact (rows, cnt) block, which goes to a
        // staging buffer; the exchange below places every row chunk into the (rows, ndev * chunk) buffers
        if (mem == BSK_DEVICE && gather) HIPCHK(m->out_ws[d].reserve(std::max<size_t>(16, es * (size_t)chunk * rows)));
        if (cnt == 0) continue;
        bsk_status r;
        if (mem == BSK_DEVICE && gather) {
            char *stage = static_cast<char *>(m->out_ws[d].p);
            r = jac ? bsk_jacobian(m->h[d], ptrs.data(), cnt, BSK_DEVICE, stage, m->st[d], nullptr)
                    : bsk_evaluate(m->h[d], wrt, ptrs.data(), cnt, BSK_DEVICE, stage, m->st[d], nullptr);
        } else {
            r = jac ? bsk_jacobian(m->h[d], ptrs.data(), cnt, BSK_DEVICE, dout, m->st[d], nullptr)
                    : bsk_evaluate(m->h[d], wrt, ptrs.data(), cnt, BSK_DEVICE, dout, m->st[d], nullptr);
        }
        if (r != BSK_OK) return r;
        if (mem == BSK_HOST) {
            char *const host = static_cast<char *>(out[0]);
            for (int row = 0; row < rows; ++row)
                HIPCHK(hipMemcpyAsync(host + es * ((size_t)row * n + s0), dout + es * (size_t)row * cnt, es * (size_t)cnt,
                                      hipMemcpyDeviceToHost, m->st[d]));
        }
    }
    if (mem == BSK_DEVICE && gather && n > 0) {
        // ONE grouped exchange: per device and row an all-gather of its chunk into row r of the full buffer
        // (a short tail shard sends its staging block's first chunk values; the receivers' extra columns are padding)
        const ncclDataType_t dt = m->dtype == BSK_F32 ? ncclFloat32 : ncclFloat64;
        NCCLCHK(R->GroupStart());
        for (int row = 0; row < rows; ++row)
            for (int d = 0; d < m->ndev; ++d) {
                const int64_t s0 = std::min<int64_t>((int64_t)d * chunk, n), cnt = std::min<int64_t>(s0 + chunk, n) - s0;
                const char *send = static_cast<const char *>(m->out_ws[d].p) + es * (size_t)row * std::max<int64_t>(cnt, 0);
                char *recv = static_cast<char *>(out[d]) + es * (size_t)row * (size_t)m->ndev * chunk;
                NCCLCHK(R->AllGather(send, recv, (size_t)chunk, dt, m->comm[d], m->st[d]));
            }
        NCCLCHK(R->GroupEnd());
    }
    // wait for every device and collect the first out-of-domain point (global index)
    int64_t bad_global = -1;
    for (int d = 0; d < m->ndev; ++d) {
        const int64_t s0 = std::min<int64_t>((int64_t)d * chunk, n);
        HIPCHK(hipSetDevice(m->dev[d]));
        int64_t bad = -1;
        const bsk_status r = bsk_domain_status(m->h[d], m->st[d], &bad);
        if (r == BSK_ERR_DOMAIN) {
            if (bad_global < 0 || s0 + bad < bad_global) bad_global = s0 + bad;
        } else if (r != BSK_OK) {
            return r;
        }
    }
    if (bad_global >= 0) {
        if (first_bad) *first_bad = bad_global;
        return fail(BSK_ERR_DOMAIN, "parameter outside the spline's domain");
    }
    return BSK_OK;
}

extern "C" bsk_status bsk_multi_evaluate(bsk_multi m, const int *wrt, const void *const *uvw, int64_t n, bsk_mem mem,
                                         void *const *out, int gather, int64_t *first_bad)
{
    return multi_run(m, false, wrt, uvw, n, mem, out, gather, first_bad);
}

extern "C" bsk_status bsk_multi_jacobian(bsk_multi m, const void *const *uvw, int64_t n, bsk_mem mem, void *const *out,
                                         int gather, int64_t *first_bad)
{
    return multi_run(m, true, nullptr, uvw, n, mem, out, gather, first_bad);
}
