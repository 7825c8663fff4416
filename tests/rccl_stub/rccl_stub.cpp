// rccl_stub.cpp -- TEST SCAFFOLDING, not part of the product: a stand-in for the six RCCL entry points cds_multi.hip binds
// (ncclCommInitAll, ncclCommDestroy, ncclGetErrorString, ncclGroupStart, ncclGroupEnd, ncclGather; /opt/rocm/include/rccl/rccl.h),
// selected with CDS_RCCL_LIBRARY=<this .so> + CDS_COMM_FORCE_RCCL=1.  It ACCEPTS duplicate devices (real RCCL refuses them), so the
// n > 1 branch of cds_search_sparse_sharded -- one grouped gather, one call per rank, NULL receive buffers on the non-roots, the root's
// read-back ordered behind the root's stream only, the peers' streams synchronised afterwards -- runs on a box with ONE GPU.
//
// Semantics kept from the real thing, so that an ordering bug in the caller shows as wrong scores or as an error:
//   * a gather is STREAM-ORDERED: rank r's bytes are read only after everything enqueued on rank r's stream before its ncclGather
//     call (an event recorded there, waited for by the root's stream), and land in the root's buffer in the root's stream order;
//   * every peer's stream is busy until the transfer is done (it waits for an event behind the root's copies), as a peer's send
//     kernel would be;
//   * nothing moves before ncclGroupEnd; calls outside a group, a non-NULL receive buffer on a non-root, a NULL one on the root,
//     unequal counts or roots, a rank called twice or not at all, and a call made while another device than the rank's is current
//     are ERRORS (ncclInvalidUsage / ncclInvalidArgument), reported through ncclGetErrorString.
// Built by tests/test_dist_gpu.py with hipcc; host code only.
#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <mutex>
#include <string>
#include <vector>

namespace {
enum { kSuccess = 0, kUnhandledHipError = 1, kInvalidArgument = 4, kInvalidUsage = 5 };

struct Call {
    bool made = false;
    const void *send = nullptr;
    void *recv = nullptr;
    size_t count = 0;
    int dtype = 0, root = 0;
    hipStream_t stream = nullptr;
};
struct Clique {
    int n = 0;
    std::vector<int> devices;
    std::vector<Call> calls;
    int live = 0;   // ranks not destroyed yet
    unsigned long gathers = 0;
};
struct Comm {
    Clique *clique;
    int rank;
};
std::mutex g_mu;
thread_local int g_group_depth = 0;
thread_local std::vector<Clique *> g_touched;
std::string g_why = "no error";
unsigned long g_gathers_total = 0, g_init_calls = 0;

int bad(int code, const std::string &why) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_why = "rccl_stub: " + why;
    return code;
}
size_t dtype_bytes(int t) {   // ncclDataType_t: int8 0, uint8 1, int32 2, uint32 3, int64 4, uint64 5, half 6, float 7, double 8, bf16 9
    switch (t) {
        case 0: case 1: return 1;
        case 2: case 3: case 7: return 4;
        case 4: case 5: case 8: return 8;
        case 6: case 9: return 2;
        default: return 0;
    }
}

int run_gather(Clique *q) {
    for (int r = 0; r < q->n; r++)
        if (!q->calls[(size_t)r].made) return bad(kInvalidUsage, "ncclGroupEnd: rank " + std::to_string(r) + " did not call ncclGather (a real gather would hang)");
    const Call &c0 = q->calls[0];
    const int root = c0.root;
    if (root < 0 || root >= q->n) return bad(kInvalidArgument, "root out of range");
    const size_t bytes = c0.count * dtype_bytes(c0.dtype);
    if (!dtype_bytes(c0.dtype)) return bad(kInvalidArgument, "unknown data type");
    for (int r = 0; r < q->n; r++) {
        const Call &c = q->calls[(size_t)r];
        if (c.count != c0.count || c.dtype != c0.dtype || c.root != root) return bad(kInvalidArgument, "ranks disagree on count / type / root");
        if (r == root && !c.recv) return bad(kInvalidArgument, "the root's receive buffer is NULL");
        if (r != root && c.recv) return bad(kInvalidArgument, "a non-root passed a receive buffer (cds_multi.hip passes NULL there)");
        if (!c.send && bytes) return bad(kInvalidArgument, "NULL send buffer");
    }
    const Call &cr = q->calls[(size_t)root];
    int prev = -1;
    hipGetDevice(&prev);
    hipError_t e = hipSuccess;
    std::vector<hipEvent_t> evs;
    auto ev_on = [&](int dev, hipStream_t st, hipEvent_t *out) {
        e = hipSetDevice(dev);
        if (e == hipSuccess) e = hipEventCreateWithFlags(out, hipEventDisableTiming);
        if (e == hipSuccess) { evs.push_back(*out); e = hipEventRecord(*out, st); }
    };
    // rank r's data is ready at this point of rank r's stream; the root's stream waits for it, then copies
    for (int r = 0; r < q->n && e == hipSuccess; r++) {
        const Call &c = q->calls[(size_t)r];
        hipEvent_t ready = nullptr;
        if (r != root) ev_on(q->devices[(size_t)r], c.stream, &ready);
        if (e == hipSuccess) e = hipSetDevice(q->devices[(size_t)root]);
        if (e == hipSuccess && ready) e = hipStreamWaitEvent(cr.stream, ready, 0);
        if (e == hipSuccess && bytes)
            e = hipMemcpyAsync(static_cast<char *>(cr.recv) + (size_t)r * bytes, c.send, bytes, hipMemcpyDeviceToDevice, cr.stream);
    }
    // every peer's stream stays busy until the root has its bytes
    hipEvent_t done = nullptr;
    if (e == hipSuccess) ev_on(q->devices[(size_t)root], cr.stream, &done);
    for (int r = 0; r < q->n && e == hipSuccess; r++) {
        if (r == root) continue;
        e = hipSetDevice(q->devices[(size_t)r]);
        if (e == hipSuccess) e = hipStreamWaitEvent(q->calls[(size_t)r].stream, done, 0);
    }
    for (hipEvent_t ev : evs) hipEventDestroy(ev);   // (destruction is deferred by the runtime until the recorded work is done)
    if (prev >= 0) hipSetDevice(prev);
    for (Call &c : q->calls) c = Call();
    if (e != hipSuccess) return bad(kUnhandledHipError, std::string("HIP: ") + hipGetErrorString(e));
    std::lock_guard<std::mutex> lk(g_mu);
    q->gathers++;
    g_gathers_total++;
    return kSuccess;
}
}  // namespace

extern "C" {

int ncclCommInitAll(void **comms, int ndev, const int *devlist) {
    if (!comms || ndev < 1) return bad(kInvalidArgument, "ncclCommInitAll: bad arguments");
    int have = 0;
    if (hipGetDeviceCount(&have) != hipSuccess) return bad(kUnhandledHipError, "hipGetDeviceCount");
    Clique *q = new Clique();
    q->n = ndev;
    q->live = ndev;
    q->calls.resize((size_t)ndev);
    for (int i = 0; i < ndev; i++) {
        const int d = devlist ? devlist[i] : i;
        if (d < 0 || d >= have) { delete q; return bad(kInvalidArgument, "ncclCommInitAll: no such device"); }
        q->devices.push_back(d);   // (duplicates accepted: the point of this stand-in)
    }
    for (int i = 0; i < ndev; i++) comms[i] = new Comm{q, i};
    std::lock_guard<std::mutex> lk(g_mu);
    g_init_calls++;
    return kSuccess;
}

int ncclCommDestroy(void *comm) {
    if (!comm) return kSuccess;
    Comm *c = static_cast<Comm *>(comm);
    bool last;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        last = --c->clique->live == 0;
    }
    if (last) delete c->clique;
    delete c;
    return kSuccess;
}

const char *ncclGetErrorString(int code) {
    static thread_local std::string s;
    std::lock_guard<std::mutex> lk(g_mu);
    s = code == kSuccess ? "no error" : g_why;
    return s.c_str();
}

int ncclGroupStart() {
    g_group_depth++;
    return kSuccess;
}

int ncclGather(const void *sendbuff, void *recvbuff, size_t sendcount, int datatype, int root, void *comm, hipStream_t stream) {
    if (!comm) return bad(kInvalidArgument, "ncclGather: NULL communicator");
    if (g_group_depth == 0) return bad(kInvalidUsage, "ncclGather outside ncclGroupStart / ncclGroupEnd: one thread driving several ranks must group them");
    Comm *c = static_cast<Comm *>(comm);
    int cur = -1;
    hipGetDevice(&cur);
    if (cur != c->clique->devices[(size_t)c->rank])
        return bad(kInvalidUsage, "ncclGather of rank " + std::to_string(c->rank) + " while device " + std::to_string(cur) + " is current (its device is " +
                                      std::to_string(c->clique->devices[(size_t)c->rank]) + ")");
    Call &k = c->clique->calls[(size_t)c->rank];
    if (k.made) return bad(kInvalidUsage, "rank " + std::to_string(c->rank) + " called ncclGather twice in one group");
    k.made = true, k.send = sendbuff, k.recv = recvbuff, k.count = sendcount, k.dtype = datatype, k.root = root, k.stream = stream;
    bool seen = false;
    for (Clique *q : g_touched) seen = seen || q == c->clique;
    if (!seen) g_touched.push_back(c->clique);
    return kSuccess;
}

int ncclGroupEnd() {
    if (g_group_depth == 0) return bad(kInvalidUsage, "ncclGroupEnd without ncclGroupStart");
    if (--g_group_depth > 0) return kSuccess;
    int rc = kSuccess;
    for (Clique *q : g_touched) {
        const int r = run_gather(q);
        if (rc == kSuccess) rc = r;
    }
    g_touched.clear();
    return rc;
}

// (for the test: how often the stand-in was really used)
unsigned long rccl_stub_gathers(void) { std::lock_guard<std::mutex> lk(g_mu); return g_gathers_total; }
unsigned long rccl_stub_inits(void) { std::lock_guard<std::mutex> lk(g_mu); return g_init_calls; }

}  // extern "C"
