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
#include <mutex>
#include <thread>
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
    static std::once_flag once;
    std::call_once(once, []() {
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
    });
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

// This thread's current HIP device is put back when a multi-device call returns (torch and other users of the
// process keep their own idea of the current device).
struct DeviceRestore {
    int dev = -1;
    DeviceRestore() { if (hipGetDevice(&dev) != hipSuccess) dev = -1; }
    ~DeviceRestore() { if (dev >= 0) (void)hipSetDevice(dev); }
};

extern "C" bsk_status bsk_multi_destroy(bsk_multi m)
{
    if (!m) return BSK_OK;
    DeviceRestore restore;
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
    DeviceRestore restore;
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
        const ncclResult_t e = R->CommInitAll(m->comm.data(), m->ndev, m->dev.data());
        if (e != ncclSuccess) {
            m->comm.clear();                                      // a later call tries again instead of using null communicators
            return fail(BSK_ERR_HIP, std::string("ncclCommInitAll: ") + R->GetErrorString(e));
        }
    }
    return BSK_OK;
}

// Shard d of a BSK_HOST call: copies in, kernel, copies out on device d's stream, from the calling thread of this
// function.  The caller's arrays are pageable, so these copies block their thread: with more than one device every
// shard runs on a thread of its own (multi_run) and the shards move over their own PCIe links at the same time.
static bsk_status host_shard(bsk_multi m, int d, bool jac, const int *wrt, const void *const *uvw, int64_t n, int64_t s0,
                             int64_t cnt, void *host_out)
{
    const int rows = jac ? m->nDep * m->nInd : m->nDep;
    const size_t es = m->esize;
    HIPCHK(hipSetDevice(m->dev[d]));
    if (cnt == 0) return BSK_OK;
    HIPCHK(m->in_ws[d].reserve(std::max<size_t>(16, es * (size_t)cnt * m->nInd)));
    HIPCHK(m->out_ws[d].reserve(std::max<size_t>(16, es * (size_t)cnt * rows)));
    char *din = static_cast<char *>(m->in_ws[d].p), *dout = static_cast<char *>(m->out_ws[d].p);
    std::vector<const void *> ptrs((size_t)std::max(1, m->nInd));
    for (int iv = 0; iv < m->nInd; ++iv) {
        HIPCHK(hipMemcpyAsync(din + es * (size_t)iv * cnt, static_cast<const char *>(uvw[iv]) + es * (size_t)s0, es * (size_t)cnt,
                              hipMemcpyHostToDevice, m->st[d]));
        ptrs[iv] = din + es * (size_t)iv * cnt;
    }
    const bsk_status r = jac ? bsk_jacobian(m->h[d], ptrs.data(), cnt, BSK_DEVICE, dout, m->st[d], nullptr)
                             : bsk_evaluate(m->h[d], wrt, ptrs.data(), cnt, BSK_DEVICE, dout, m->st[d], nullptr);
    if (r != BSK_OK) return r;
    char *const host = static_cast<char *>(host_out);
    for (int row = 0; row < rows; ++row)
        HIPCHK(hipMemcpyAsync(host + es * ((size_t)row * n + s0), dout + es * (size_t)row * cnt, es * (size_t)cnt,
                              hipMemcpyDeviceToHost, m->st[d]));
    return BSK_OK;
}

// Enqueues one sharded call on every device; returns the first failure (the caller drains the devices either way).
static bsk_status multi_enqueue(bsk_multi m, bool jac, const int *wrt, const void *const *uvw, int64_t n, bsk_mem mem,
                                void *const *out, int gather, Rccl *R)
{
    const int rows = jac ? m->nDep * m->nInd : m->nDep;
    const size_t es = m->esize;
    const int64_t chunk = n > 0 ? (n + m->ndev - 1) / m->ndev : 0;
    auto shard = [&](int d, int64_t &s0, int64_t &cnt) {
        s0 = std::min<int64_t>((int64_t)d * chunk, n);
        cnt = std::min<int64_t>(s0 + chunk, n) - s0;
    };
    if (mem == BSK_HOST) {
        if (m->ndev == 1) {
            int64_t s0, cnt;
            shard(0, s0, cnt);
            return host_shard(m, 0, jac, wrt, uvw, n, s0, cnt, out[0]);
        }
        std::vector<bsk_status> rc((size_t)m->ndev, BSK_OK);
        std::vector<std::string> msg((size_t)m->ndev);
        std::vector<std::thread> th;
        for (int d = 0; d < m->ndev; ++d)
            th.emplace_back([&, d]() {
                int64_t s0, cnt;
                shard(d, s0, cnt);
                rc[d] = host_shard(m, d, jac, wrt, uvw, n, s0, cnt, out[0]);
                if (rc[d] != BSK_OK) msg[d] = g_err;             // the message lives in the worker's thread-local
            });
        for (auto &t : th) t.join();
        for (int d = 0; d < m->ndev; ++d)
            if (rc[d] != BSK_OK) return fail(rc[d], msg[d]);
        return BSK_OK;
    }
    // device buffers: uvw[d * nInd + iv] = device d's shard of variable iv.  Without gather the (rows, cnt) block is
    // written to out[d]; with gather it goes to a staging buffer and the exchange below places every row chunk into
    // row r of every device's full-size buffer out[d] ((rows, ndev * chunk)).
    std::vector<const void *> ptrs((size_t)std::max(1, m->nInd));
    for (int d = 0; d < m->ndev; ++d) {
        int64_t s0, cnt;
        shard(d, s0, cnt);
        HIPCHK(hipSetDevice(m->dev[d]));
        for (int iv = 0; iv < m->nInd; ++iv) ptrs[iv] = uvw[(size_t)d * m->nInd + iv];
        char *dout = static_cast<char *>(out[d]);
        if (!dout) return fail(BSK_ERR_INVALID, "NULL output pointer");
        if (gather) {
            HIPCHK(m->out_ws[d].reserve(std::max<size_t>(16, es * (size_t)chunk * rows)));
            dout = static_cast<char *>(m->out_ws[d].p);
        }
        if (cnt == 0) continue;
        const bsk_status r = jac ? bsk_jacobian(m->h[d], ptrs.data(), cnt, BSK_DEVICE, dout, m->st[d], nullptr)
                                 : bsk_evaluate(m->h[d], wrt, ptrs.data(), cnt, BSK_DEVICE, dout, m->st[d], nullptr);
        if (r != BSK_OK) return r;
    }
    if (gather && n > 0) {
        // ONE grouped exchange: per device and row an all-gather of its chunk into row r of the full buffer
        // (a short tail shard sends `chunk` values from its staging block, which is sized for a full chunk; the
        // receivers' columns beyond n are padding)
        const ncclDataType_t dt = m->dtype == BSK_F32 ? ncclFloat32 : ncclFloat64;
        NCCLCHK(R->GroupStart());
        bsk_status rc = BSK_OK;
        for (int row = 0; row < rows && rc == BSK_OK; ++row)
            for (int d = 0; d < m->ndev; ++d) {
                int64_t s0, cnt;
                shard(d, s0, cnt);
                const char *send = static_cast<const char *>(m->out_ws[d].p) + es * (size_t)row * (size_t)cnt;
                char *recv = static_cast<char *>(out[d]) + es * (size_t)row * (size_t)m->ndev * chunk;
                const ncclResult_t e = R->AllGather(send, recv, (size_t)chunk, dt, m->comm[d], m->st[d]);
                if (e != ncclSuccess) { rc = fail(BSK_ERR_HIP, std::string("ncclAllGather: ") + R->GetErrorString(e)); break; }
            }
        const ncclResult_t e = R->GroupEnd();                     // the group is closed on the failure path too
        if (rc != BSK_OK) return rc;
        if (e != ncclSuccess) return fail(BSK_ERR_HIP, std::string("ncclGroupEnd: ") + R->GetErrorString(e));
    }
    return BSK_OK;
}

// One sharded call: jacobian ? bsk_jacobian : bsk_evaluate(wrt) on every device's shard.
static bsk_status multi_run(bsk_multi m, bool jac, const int *wrt, const void *const *uvw, int64_t n, bsk_mem mem, void *const *out,
                            int gather, int64_t *first_bad)
{
    if (first_bad) *first_bad = -1;
    if (!m || !uvw || !out || n < 0) return fail(BSK_ERR_INVALID, "bad argument");
    const int64_t chunk = n > 0 ? (n + m->ndev - 1) / m->ndev : 0;
    DeviceRestore restore;
    Rccl *R = nullptr;
    if (mem == BSK_DEVICE && gather) {
        const bsk_status r = multi_comm(m, R);
        if (r != BSK_OK) return r;
    }
    const bsk_status rc = multi_enqueue(m, jac, wrt, uvw, n, mem, out, mem == BSK_DEVICE && gather, R);
    const std::string rc_msg = rc != BSK_OK ? g_err : std::string();
    // Wait for EVERY device - also after a failure: work already enqueued elsewhere still writes the caller's buffers,
    // and an out-of-domain record left behind would be reported by the next call - and collect the first
    // out-of-domain point (global index).
    int64_t bad_global = -1;
    bsk_status drain = BSK_OK;
    std::string drain_msg;
    for (int d = 0; d < m->ndev; ++d) {
        const int64_t s0 = std::min<int64_t>((int64_t)d * chunk, n);
        if (hipSetDevice(m->dev[d]) != hipSuccess) { if (drain == BSK_OK) { drain = BSK_ERR_HIP; drain_msg = "hipSetDevice failed"; } continue; }
        int64_t bad = -1;
        const bsk_status r = bsk_domain_status(m->h[d], m->st[d], &bad);     // synchronises the stream, resets the record
        if (r == BSK_ERR_DOMAIN) {
            if (bad_global < 0 || s0 + bad < bad_global) bad_global = s0 + bad;
        } else if (r != BSK_OK && drain == BSK_OK) {
            drain = r;
            drain_msg = g_err;
        }
    }
    if (rc != BSK_OK) return fail(rc, rc_msg);
    if (drain != BSK_OK) return fail(drain, drain_msg);
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
