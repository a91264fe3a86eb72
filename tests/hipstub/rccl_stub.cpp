// Stand-in for librccl.so.1 on the CPU box (TEST INFRASTRUCTURE, found by bsk_multi.hip's dlopen through
// LD_LIBRARY_PATH): single-process communicators from ncclCommInitAll, all-gather as memcpy between the fake
// devices' host memory.  Calls inside ncclGroupStart / ncclGroupEnd are queued per communicator and executed at
// ncclGroupEnd, matched across the ranks of a clique by their order - the semantics the grouped exchange of
// bsk_multi_evaluate / bsk_multi_jacobian relies on (one ncclAllGather per device and output row in ONE group).
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <mutex>
#include <vector>

namespace {
struct Op { const void *send; void *recv; size_t count; ncclDataType_t dt; };
struct Clique;
struct Comm { Clique *clique; int rank; std::vector<Op> ops; };
struct Clique { std::vector<Comm *> ranks; int alive; };
std::mutex g_mu;
int g_group = 0;
std::vector<Clique *> g_pending;
long g_allgathers = 0, g_groups = 0;

size_t dsize(ncclDataType_t dt) { return dt == ncclFloat64 || dt == ncclInt64 || dt == ncclUint64 ? 8 : dt == ncclFloat32 || dt == ncclInt32 || dt == ncclUint32 ? 4 : dt == ncclFloat16 || dt == ncclBfloat16 ? 2 : 1; }

ncclResult_t run(Clique *q)
{
    const size_t nops = q->ranks[0]->ops.size();
    for (Comm *c : q->ranks)
        if (c->ops.size() != nops) { fprintf(stderr, "rccl stub: ranks of a clique queued different numbers of operations\n"); return ncclInvalidUsage; }
    for (size_t k = 0; k < nops; ++k) {
        const Op &o0 = q->ranks[0]->ops[k];
        for (Comm *c : q->ranks)
            if (c->ops[k].count != o0.count || c->ops[k].dt != o0.dt) { fprintf(stderr, "rccl stub: mismatched all-gather\n"); return ncclInvalidUsage; }
        const size_t bytes = o0.count * dsize(o0.dt);
        for (Comm *r : q->ranks)
            for (Comm *s : q->ranks) {
                char *dst = static_cast<char *>(r->ops[k].recv) + (size_t)s->rank * bytes;
                if (dst != s->ops[k].send) memmove(dst, s->ops[k].send, bytes);
            }
    }
    for (Comm *c : q->ranks) c->ops.clear();
    return ncclSuccess;
}
}  // namespace

extern "C" {
long rcclstub_allgathers() { return g_allgathers; }
long rcclstub_groups() { return g_groups; }

ncclResult_t ncclCommInitAll(ncclComm_t *comms, int ndev, const int *)
{
    if (!comms || ndev < 1) return ncclInvalidArgument;
    Clique *q = new Clique();
    q->alive = ndev;
    for (int r = 0; r < ndev; ++r) {
        Comm *c = new Comm{q, r, {}};
        q->ranks.push_back(c);
        comms[r] = reinterpret_cast<ncclComm_t>(c);
    }
    return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    Comm *c = reinterpret_cast<Comm *>(comm);
    if (!c) return ncclInvalidArgument;
    Clique *q = c->clique;
    delete c;
    if (--q->alive == 0) delete q;
    return ncclSuccess;
}
ncclResult_t ncclGroupStart() { std::lock_guard<std::mutex> l(g_mu); ++g_group; return ncclSuccess; }
ncclResult_t ncclGroupEnd()
{
    std::lock_guard<std::mutex> l(g_mu);
    if (g_group <= 0) return ncclInvalidUsage;
    if (--g_group > 0) return ncclSuccess;
    ++g_groups;
    ncclResult_t rc = ncclSuccess;
    for (Clique *q : g_pending) { const ncclResult_t e = run(q); if (e != ncclSuccess) rc = e; }
    g_pending.clear();
    return rc;
}
ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, ncclDataType_t dt, ncclComm_t comm, hipStream_t)
{
    std::lock_guard<std::mutex> l(g_mu);
    Comm *c = reinterpret_cast<Comm *>(comm);
    if (!c || !send || !recv) return ncclInvalidArgument;
    ++g_allgathers;
    c->ops.push_back({send, recv, count, dt});
    if (g_group == 0) {                      // ungrouped: legal only for a one-rank clique in a single thread
        if (c->clique->ranks.size() != 1) return ncclInvalidUsage;
        return run(c->clique);
    }
    bool seen = false;
    for (Clique *q : g_pending) seen |= q == c->clique;
    if (!seen) g_pending.push_back(c->clique);
    return ncclSuccess;
}
const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "rccl stub error"; }
}
