// Drives the host paths of the C ABI on the stub runtime: handle life cycle and update, the small-call path
// (pinned mapped buffer), the staged path, the pipelined path (copy-thread pool, double-buffered pinned
// staging, three streams) for evaluate / jacobian / normal / curvature / grid / tessellate, from two threads
// on two handles at once (handles are independent).  Exit code 0 and no sanitizer report = pass.
#include "../../include/bspy_amd.h"

#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

extern "C" long hipstub_live_allocations();
extern "C" long hipstub_kernel_launches();

#define CHECK(x)                                                                                  \
    do {                                                                                          \
        bsk_status s_ = (x);                                                                      \
        if (s_ != BSK_OK) { fprintf(stderr, "%s -> %d: %s\n", #x, (int)s_, bsk_last_error()); exit(2); } \
    } while (0)

static long long g_big = 4500000;      // largest batch: above the pipelined path's threshold (2^21 points)

static void work(int seed)
{
    const int order[2] = {4, 4}, ncoef[2] = {16, 12};
    std::vector<double> ku(order[0] + ncoef[0]), kv(order[1] + ncoef[1]), coefs(3 * ncoef[0] * ncoef[1]);
    for (size_t i = 0; i < ku.size(); ++i) ku[i] = i < 4 ? 0.0 : (i >= 16 ? 1.0 : (i - 3) / 13.0);
    for (size_t i = 0; i < kv.size(); ++i) kv[i] = i < 4 ? 0.0 : (i >= 12 ? 1.0 : (i - 3) / 9.0);
    for (size_t i = 0; i < coefs.size(); ++i) coefs[i] = (double)((i * 7 + seed) % 13) - 6.0;
    const void *knots[2] = {ku.data(), kv.data()};
    bsk_spline s = nullptr;
    CHECK(bsk_spline_create(BSK_F64, 0, 2, 3, order, ncoef, knots, coefs.data(), &s));
    for (long long n : {1LL, 100LL, 70000LL, 300000LL, g_big}) {          // small, staged, chunked, pipelined
        std::vector<double> u(n), v(n), out((size_t)6 * n);
        for (long long i = 0; i < n; ++i) { u[i] = (double)((i * 31 + seed) % 1000) / 1000.0; v[i] = (double)((i * 17) % 1000) / 1000.0; }
        const void *uv[2] = {u.data(), v.data()};
        int64_t bad = 0;
        const int wrt[2] = {1, 0};
        CHECK(bsk_evaluate(s, nullptr, uv, n, BSK_HOST, out.data(), nullptr, &bad));
        CHECK(bsk_evaluate(s, wrt, uv, n, BSK_HOST, out.data(), nullptr, &bad));
        CHECK(bsk_jacobian(s, uv, n, BSK_HOST, out.data(), nullptr, &bad));
        CHECK(bsk_normal(s, uv, n, BSK_HOST, 1, 0, out.data(), nullptr, &bad));
        CHECK(bsk_curvature(s, uv, n, BSK_HOST, out.data(), nullptr, &bad));
        if (bad != -1) { fprintf(stderr, "unexpected offender %lld\n", (long long)bad); exit(3); }
    }
    coefs[5] += 1.0;
    CHECK(bsk_spline_update(s, knots, coefs.data()));
    {
        std::vector<double> gu(300), gv(200), out((size_t)3 * 300 * 200 * 2);
        for (int i = 0; i < 300; ++i) gu[i] = i / 299.0;
        for (int i = 0; i < 200; ++i) gv[i] = i / 199.0;
        const void *grid[2] = {gu.data(), gv.data()};
        const int64_t ng[2] = {300, 200};
        int64_t bad = 0;
        CHECK(bsk_evaluate_grid(s, nullptr, grid, ng, BSK_HOST, out.data(), nullptr, &bad));
        CHECK(bsk_tessellate(&s, 1, grid, ng, BSK_HOST, 1, 0, out.data(), out.data() + (size_t)3 * 300 * 200, nullptr, &bad));
    }
    {
        std::vector<double> u(50, 0.5), basis(50 * 4);
        std::vector<int32_t> ix(50);
        CHECK(bsk_bspline_values(BSK_F64, 0, ku.data(), (int)ku.size(), 4, u.data(), 50, 1, 0, nullptr, ix.data(), basis.data()));
    }
    CHECK(bsk_spline_destroy(s));
}

// The REAL single-device host code under the multi-device entry points, on every fake device (HIPSTUB_DEVICES):
// tables replicated per device, BSK_HOST shards on a thread per device, device buffers with the grouped all-gather
// of the stub librccl.  Kernels are no-ops here (results stay zero); the sanitizer checks every copy's bounds.
static void multi_work()
{
    int ndev = 0;
    CHECK(bsk_device_count(&ndev));
    const int order[2] = {4, 4}, ncoef[2] = {16, 12};
    std::vector<double> ku(order[0] + ncoef[0]), kv(order[1] + ncoef[1]), coefs(3 * ncoef[0] * ncoef[1], 1.0);
    for (size_t i = 0; i < ku.size(); ++i) ku[i] = i < 4 ? 0.0 : (i >= 16 ? 1.0 : (i - 3) / 13.0);
    for (size_t i = 0; i < kv.size(); ++i) kv[i] = i < 4 ? 0.0 : (i >= 12 ? 1.0 : (i - 3) / 9.0);
    const void *knots[2] = {ku.data(), kv.data()};
    bsk_multi m = nullptr;
    CHECK(bsk_multi_create(BSK_F64, ndev, nullptr, 2, 3, order, ncoef, knots, coefs.data(), &m));
    for (long long n : {0LL, 7LL, 100003LL}) {
        std::vector<double> u(n, 0.25), v(n, 0.5), out((size_t)6 * n);
        const void *uv[2] = {u.data(), v.data()};
        void *outs[1] = {out.data()};
        int64_t bad = 0;
        const int wrt[2] = {0, 1};
        CHECK(bsk_multi_evaluate(m, wrt, uv, n, BSK_HOST, outs, 0, &bad));
        CHECK(bsk_multi_jacobian(m, uv, n, BSK_HOST, outs, 0, &bad));
        std::vector<int64_t> start(ndev + 1);
        CHECK(bsk_multi_shard_plan(m, n, start.data()));
        const long long chunk = n > 0 ? (n + ndev - 1) / ndev : 0;
        std::vector<std::vector<double>> du(ndev), dv(ndev), dout(ndev);
        std::vector<const void *> duv(2 * ndev);
        std::vector<void *> douts(ndev);
        for (int d = 0; d < ndev; ++d) {
            du[d].assign(start[d + 1] - start[d] + 1, 0.25);
            dv[d].assign(start[d + 1] - start[d] + 1, 0.5);
            dout[d].assign((size_t)6 * ndev * chunk + 1, 0.0);
            duv[2 * d] = du[d].data();
            duv[2 * d + 1] = dv[d].data();
            douts[d] = dout[d].data();
        }
        for (int gather = 0; gather < 2; ++gather) {
            CHECK(bsk_multi_evaluate(m, nullptr, duv.data(), n, BSK_DEVICE, douts.data(), gather, &bad));
            CHECK(bsk_multi_jacobian(m, duv.data(), n, BSK_DEVICE, douts.data(), gather, &bad));
        }
    }
    CHECK(bsk_multi_destroy(m));
}

int main(int argc, char **argv)
{
    if (argc > 1) g_big = atoll(argv[1]);
    multi_work();
    std::thread a(work, 1), b(work, 2);
    a.join();
    b.join();
    work(3);
    printf("hipstub driver ok: %ld kernel launches, %ld live allocations\n", hipstub_kernel_launches(), hipstub_live_allocations());
    return 0;
}
