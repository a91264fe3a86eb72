// Drives the multi-device entry points (bspy_amd/csrc/bsk_multi.hip, compiled host-only and unchanged) on the stub
// runtime with HIPSTUB_DEVICES fake devices and the stub librccl.so.1 (rccl_stub.cpp).  TEST INFRASTRUCTURE.
//
// The single-device layer below bsk_multi.hip (bsk_spline_create / bsk_evaluate / bsk_jacobian / bsk_domain_status)
// is replaced HERE by a recognisable stand-in: result row r of point u is  1000 r + 100 wrt0 + u0 + u1 / 2,  a point
// with u0 > 1 is out of the domain and is recorded per handle until bsk_domain_status reads it (like the kernels'
// record), and `g_fail_device` makes one device's call fail.  Every value of every output buffer is therefore known:
// the test checks the shard plan, the staging offsets, the placement of the grouped all-gather (ragged tail, empty
// tail shards), the first offender's global index, that every device is drained after a failure, and that the
// calling thread's current device is restored - for 1, 2, 3 and 8 devices, fp64 and fp32, under ASAN / TSAN.
#include "../../bspy_amd/csrc/bsk_host.hpp"

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <vector>

extern "C" long rcclstub_allgathers();
extern "C" long rcclstub_groups();
extern "C" long hipstub_live_allocations();

static int g_fail_device = -1;
static std::atomic<long> g_calls{0};

struct FakeState { int64_t bad = -1; };
static std::vector<FakeState *> g_states;

template <typename T>
static void fake_rows(bsk_spline s, int rows, int w0, const void *const *uvw, int64_t n, void *out)
{
    FakeState *fs = static_cast<FakeState *>(s->pin);
    const T *u0 = static_cast<const T *>(uvw[0]);
    const T *u1 = s->nInd > 1 ? static_cast<const T *>(uvw[1]) : nullptr;
    T *o = static_cast<T *>(out);
    for (int64_t i = 0; i < n; ++i) {
        if (u0[i] > T(1) && (fs->bad < 0 || i < fs->bad)) fs->bad = i;
        for (int r = 0; r < rows; ++r) o[(size_t)r * n + i] = T(1000 * r + 100 * w0) + u0[i] + (u1 ? u1[i] / 2 : T(0));
    }
}
static bsk_status fake_call(bsk_spline s, int rows, int w0, const void *const *uvw, int64_t n, bsk_mem mem, void *out)
{
    ++g_calls;
    int cur = -1;
    (void)hipGetDevice(&cur);
    if (cur != s->device) return fail(BSK_ERR_INVALID, "fake: the current device is not the handle's device");
    if (mem != BSK_DEVICE) return fail(BSK_ERR_INVALID, "fake: bsk_multi must call the device form");
    if (s->device == g_fail_device) return fail(BSK_ERR_HIP, "fake: injected failure");
    if (s->dtype == BSK_F32) fake_rows<float>(s, rows, w0, uvw, n, out);
    else fake_rows<double>(s, rows, w0, uvw, n, out);
    return BSK_OK;
}
extern "C" {
const char *bsk_last_error(void) { return g_err.c_str(); }
bsk_status bsk_spline_create(bsk_dtype dtype, int device, int nInd, int nDep, const int *, const int *, const void *const *,
                             const void *, bsk_spline *out)
{
    bsk_spline s = new bsk_spline_s();
    s->dtype = dtype;
    s->device = device;
    s->nInd = nInd;
    s->nDep = nDep;
    s->esize = dtype == BSK_F32 ? 4 : 8;
    s->pin = new FakeState();
    *out = s;
    return BSK_OK;
}
bsk_status bsk_spline_destroy(bsk_spline s)
{
    if (s) { delete static_cast<FakeState *>(s->pin); delete s; }
    return BSK_OK;
}
bsk_status bsk_evaluate(bsk_spline s, const int *wrt, const void *const *uvw, int64_t n, bsk_mem mem, void *out, void *, int64_t *)
{
    return fake_call(s, s->nDep, wrt ? wrt[0] : 0, uvw, n, mem, out);
}
bsk_status bsk_jacobian(bsk_spline s, const void *const *uvw, int64_t n, bsk_mem mem, void *out, void *, int64_t *)
{
    return fake_call(s, s->nDep * s->nInd, 7, uvw, n, mem, out);
}
bsk_status bsk_domain_status(bsk_spline s, void *, int64_t *first_bad)
{
    FakeState *fs = static_cast<FakeState *>(s->pin);
    const int64_t b = fs->bad;
    fs->bad = -1;
    if (first_bad) *first_bad = b;
    return b >= 0 ? fail(BSK_ERR_DOMAIN, "fake: outside") : BSK_OK;
}
}

#define REQUIRE(c)                                                                                   \
    do {                                                                                             \
        if (!(c)) { fprintf(stderr, "multi_driver: %s:%d: %s failed (%s)\n", __FILE__, __LINE__, #c, bsk_last_error()); exit(2); } \
    } while (0)

template <typename T>
static T expect(int row, int w0, T u0, T u1) { return T(1000 * row + 100 * w0) + u0 + u1 / 2; }

template <typename T>
static void run_case(bsk_dtype dt, int ndev, int64_t n)
{
    const int order[2] = {4, 4}, ncoef[2] = {8, 8};
    std::vector<T> kn(12, T(0)), coefs(3 * 64, T(1));
    const void *knots[2] = {kn.data(), kn.data()};
    bsk_multi m = nullptr;
    (void)hipSetDevice(ndev - 1);                               // the caller's current device must survive every call
    REQUIRE(bsk_multi_create(dt, ndev, nullptr, 2, 3, order, ncoef, knots, coefs.data(), &m) == BSK_OK);
    int cur = -1;
    (void)hipGetDevice(&cur);
    REQUIRE(cur == ndev - 1);
    std::vector<int64_t> start((size_t)ndev + 1);
    REQUIRE(bsk_multi_shard_plan(m, n, start.data()) == BSK_OK);
    const int64_t chunk = n > 0 ? (n + ndev - 1) / ndev : 0;
    REQUIRE(start[0] == 0 && start[ndev] == n);
    for (int d = 0; d < ndev; ++d) REQUIRE(start[d + 1] - start[d] <= chunk && start[d + 1] >= start[d]);

    std::vector<T> u((size_t)n), v((size_t)n);
    for (int64_t i = 0; i < n; ++i) { u[i] = T(i % 997) / T(1024); v[i] = T(i % 13) / T(16); }
    const int wrt[2] = {2, 0};

    // ---- host buffers: one (rows, n) result
    for (int jac = 0; jac < 2; ++jac) {
        const int rows = jac ? 6 : 3, w0 = jac ? 7 : 2;
        std::vector<T> out((size_t)rows * n + 1, T(-1));
        const void *uv[2] = {u.data(), v.data()};
        void *outs[1] = {out.data()};
        int64_t bad = 5;
        REQUIRE((jac ? bsk_multi_jacobian(m, uv, n, BSK_HOST, outs, 0, &bad) : bsk_multi_evaluate(m, wrt, uv, n, BSK_HOST, outs, 0, &bad)) == BSK_OK);
        REQUIRE(bad == -1);
        for (int r = 0; r < rows; ++r)
            for (int64_t i = 0; i < n; ++i) REQUIRE(out[(size_t)r * n + i] == expect<T>(r, w0, u[i], v[i]));
        REQUIRE(out[(size_t)rows * n] == T(-1));
        (void)hipGetDevice(&cur);
        REQUIRE(cur == ndev - 1);
    }

    // ---- device buffers ("device" memory of the stub is host memory), sharded inputs
    std::vector<std::vector<T>> du((size_t)ndev), dv((size_t)ndev);
    std::vector<const void *> duv((size_t)ndev * 2);
    for (int d = 0; d < ndev; ++d) {
        du[d].assign(u.begin() + start[d], u.begin() + start[d + 1]);
        dv[d].assign(v.begin() + start[d], v.begin() + start[d + 1]);
        du[d].push_back(T(0));                                  // never NULL, also for an empty shard
        dv[d].push_back(T(0));
        duv[(size_t)d * 2] = du[d].data();
        duv[(size_t)d * 2 + 1] = dv[d].data();
    }
    for (int jac = 0; jac < 2; ++jac)
        for (int gather = 0; gather < 2; ++gather) {
            const int rows = jac ? 6 : 3, w0 = jac ? 7 : 2;
            const size_t width = gather ? (size_t)ndev * chunk : 0;
            std::vector<std::vector<T>> dout((size_t)ndev);
            std::vector<void *> outs((size_t)ndev);
            for (int d = 0; d < ndev; ++d) {
                const size_t cnt = (size_t)(start[d + 1] - start[d]);
                dout[d].assign(gather ? rows * width + 1 : rows * cnt + 1, T(-1));      // exact size + a sentinel: ASAN and the check below see overruns
                outs[d] = dout[d].data();
            }
            int64_t bad = 5;
            const long groups0 = rcclstub_groups(), ag0 = rcclstub_allgathers();
            REQUIRE((jac ? bsk_multi_jacobian(m, duv.data(), n, BSK_DEVICE, outs.data(), gather, &bad)
                         : bsk_multi_evaluate(m, wrt, duv.data(), n, BSK_DEVICE, outs.data(), gather, &bad)) == BSK_OK);
            REQUIRE(bad == -1);
            if (gather && n > 0) {
                REQUIRE(rcclstub_groups() == groups0 + 1);                              // ONE grouped exchange per call
                REQUIRE(rcclstub_allgathers() == ag0 + (long)rows * ndev);
            } else {
                REQUIRE(rcclstub_groups() == groups0);                                  // no collective without gather
            }
            for (int d = 0; d < ndev; ++d) {
                const size_t cnt = (size_t)(start[d + 1] - start[d]);
                if (gather) {
                    for (int r = 0; r < rows; ++r)
                        for (int64_t i = 0; i < n; ++i) REQUIRE(dout[d][(size_t)r * width + i] == expect<T>(r, w0, u[i], v[i]));
                    REQUIRE(dout[d][rows * width] == T(-1));
                } else {
                    for (int r = 0; r < rows; ++r)
                        for (size_t i = 0; i < cnt; ++i) REQUIRE(dout[d][(size_t)r * cnt + i] == expect<T>(r, w0, u[start[d] + i], v[start[d] + i]));
                    REQUIRE(dout[d][rows * cnt] == T(-1));
                }
            }
        }

    // ---- first offender: on the LAST non-empty device and on an earlier one; the records are cleared by the call
    if (n >= 2) {
        int last = ndev - 1;
        while (start[last + 1] == start[last]) --last;
        std::vector<T> ub = u;
        const int64_t b1 = n - 1, b0 = start[last] > 0 ? start[last] - 1 : 0;
        ub[b1] = T(3);
        std::vector<T> out((size_t)3 * n);
        const void *uv[2] = {ub.data(), v.data()};
        void *outs[1] = {out.data()};
        int64_t bad = -7;
        REQUIRE(bsk_multi_evaluate(m, nullptr, uv, n, BSK_HOST, outs, 0, &bad) == BSK_ERR_DOMAIN && bad == b1);
        ub[b0] = T(2);
        REQUIRE(bsk_multi_evaluate(m, nullptr, uv, n, BSK_HOST, outs, 0, &bad) == BSK_ERR_DOMAIN && bad == b0);
        uv[0] = u.data();
        REQUIRE(bsk_multi_evaluate(m, nullptr, uv, n, BSK_HOST, outs, 0, &bad) == BSK_OK && bad == -1);   // nothing left behind

        // a failing device in the middle: the call fails, every OTHER device is drained (its offender is not
        // reported by the next call) and the current device is restored
        if (ndev >= 2 && start[1] > start[0] && start[2] > start[1]) {
            std::vector<T> uf = u;
            uf[0] = T(9);                                       // an offender on device 0, which runs before the failure
            uv[0] = uf.data();
            g_fail_device = 1;
            REQUIRE(bsk_multi_evaluate(m, nullptr, uv, n, BSK_HOST, outs, 0, &bad) == BSK_ERR_HIP);
            g_fail_device = -1;
            uv[0] = u.data();
            REQUIRE(bsk_multi_evaluate(m, nullptr, uv, n, BSK_HOST, outs, 0, &bad) == BSK_OK && bad == -1);
            (void)hipGetDevice(&cur);
            REQUIRE(cur == ndev - 1);
        }
    }
    REQUIRE(bsk_multi_destroy(m) == BSK_OK);
    (void)hipGetDevice(&cur);
    REQUIRE(cur == ndev - 1);
}

int main()
{
    int have = 0;
    (void)hipGetDeviceCount(&have);
    int cases = 0;
    for (int ndev : {1, 2, 3, 8}) {
        if (ndev > have) continue;
        for (int64_t n : {(int64_t)0, (int64_t)1, (int64_t)5, (int64_t)9, (int64_t)1001, (int64_t)40000}) {
            run_case<double>(BSK_F64, ndev, n);
            run_case<float>(BSK_F32, ndev, n);
            cases += 2;
        }
    }
    if (hipstub_live_allocations() != 0) { fprintf(stderr, "multi_driver: %ld allocations leaked\n", hipstub_live_allocations()); return 3; }
    printf("multi driver ok: %d devices, %d cases, %ld single-device calls, %ld grouped exchanges, %ld all-gathers\n", have, cases, g_calls.load(),
           rcclstub_groups(), rcclstub_allgathers());
    return 0;
}
