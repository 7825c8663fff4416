// cds_multi.hip -- the target-sharded search over the GPUs of one node, behind the C ABI (one host process, n devices).
//
// Reference: the only parallel strategy of the reference is data parallelism over target partitions -- host threads in
// tools cmd/cdsprocess/LocalColorMIPSearchProcessor.java:93-116, Spark partitions whose results are collect()-ed in
// SparkColorMIPSearchProcessor.java:50-73.  Every (mask, target) score depends on that pair alone, so here the library
// is cut into n shards, each resident in its own GPU's HBM (a cds_sparse_library created through that GPU's context);
// the mask set is tiny and replicated (one cds_maskset per GPU).  A search = every GPU scans its shard (asynchronous
// launches from one host thread), then ONE RCCL gather of the per-target score arrays (int32 score | mirrored << 31) to
// the first GPU over xGMI, then one copy to the host -- no reduction, no ring, no other collective on the path.
//
// RCCL is bound at run time (dlopen of librccl.so.1: the copy already loaded by a torch host, or ROCm's): a single-GPU host
// does not need it.  A communicator over contexts that share a device (a one-GPU rehearsal of the sharding logic; RCCL
// refuses duplicate devices) moves the shards' scores with device-to-device copies instead -- same layout, same result.
#include <dlfcn.h>

#include <algorithm>
#include <cstring>
#include <map>
#include <set>
#include <thread>

#include "cds_internal.hpp"

namespace cds {

// the handful of RCCL entry points this file uses (rccl/rccl.h: ncclCommInitAll :236, ncclCommDestroy :260,
// ncclGetErrorString :339, ncclGroupStart / ncclGroupEnd, ncclGather :745)
typedef void *rccl_comm_t;
struct RcclApi {
    void *handle = nullptr;
    int (*CommInitAll)(rccl_comm_t *, int, const int *) = nullptr;
    int (*CommDestroy)(rccl_comm_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Gather)(const void *, void *, size_t, int /*ncclDataType_t*/, int /*root*/, rccl_comm_t, hipStream_t) = nullptr;
};
constexpr int kNcclInt32 = 2;  // ncclInt32 (rccl.h ncclDataType_t)

// CDS_RCCL_LIBRARY names the library to bind instead of the default search (a site's own RCCL build; tests/rccl_stub: a stand-in
// that moves the bytes with hipMemcpyAsync, so that the n > 1 control flow of this file runs on a one-GPU box).  One binding per path
// and process.
static RcclApi *load_rccl(std::string &why) {
    static std::map<std::string, RcclApi> bound;
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    const char *forced = getenv("CDS_RCCL_LIBRARY");
    const std::string key = forced && *forced ? forced : "";
    auto it = bound.find(key);
    if (it != bound.end()) return &it->second;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    if (!key.empty()) {
        h = dlopen(key.c_str(), RTLD_NOW | RTLD_LOCAL);
    } else {
        for (const char *n : names) {
            h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (h) break;
        }
    }
    if (!h) {
        const char *de = dlerror();
        why = (key.empty() ? std::string("librccl.so.1") : key) + " cannot be loaded: " + (de ? de : "?");
        return nullptr;
    }
    RcclApi a;
    a.handle = h;
    a.CommInitAll = (decltype(a.CommInitAll))dlsym(h, "ncclCommInitAll");
    a.CommDestroy = (decltype(a.CommDestroy))dlsym(h, "ncclCommDestroy");
    a.GetErrorString = (decltype(a.GetErrorString))dlsym(h, "ncclGetErrorString");
    a.GroupStart = (decltype(a.GroupStart))dlsym(h, "ncclGroupStart");
    a.GroupEnd = (decltype(a.GroupEnd))dlsym(h, "ncclGroupEnd");
    a.Gather = (decltype(a.Gather))dlsym(h, "ncclGather");
    if (!a.CommInitAll || !a.CommDestroy || !a.GetErrorString || !a.GroupStart || !a.GroupEnd || !a.Gather) {
        why = (key.empty() ? std::string("librccl.so.1") : key) + " lacks an entry point (ncclCommInitAll / ncclGather / ...)";
        return nullptr;
    }
    return &(bound[key] = a);
}

}  // namespace cds

struct cds_comm {
    std::vector<cds_context *> ctxs;
    bool rccl = false;
    cds::RcclApi *api = nullptr;
    std::vector<cds::rccl_comm_t> comms;
    // per-shard score buffers of the last search (grown on demand) and the root's gather buffer
    std::vector<int32_t *> d_send;
    std::vector<size_t> send_elems;
    int32_t *d_recv = nullptr;
    size_t recv_elems = 0;
    std::mutex mu;
};

using namespace cds;

extern "C" {

int cds_comm_create(cds_context *const *contexts, int32_t n, cds_comm **out) {
    if (!contexts || !out) return fail(CDS_ERR_INVALID_ARGUMENT, "NULL argument");
    *out = nullptr;
    if (n <= 0 || n > 64) return fail(CDS_ERR_INVALID_ARGUMENT, "a communicator has 1..64 contexts");
    std::set<int> devs;
    for (int i = 0; i < n; i++) {
        if (!contexts[i]) return fail(CDS_ERR_INVALID_ARGUMENT, "NULL context");
        devs.insert(contexts[i]->device);
    }
    cds_comm *c = new cds_comm();
    c->ctxs.assign(contexts, contexts + n);
    c->d_send.assign((size_t)n, nullptr);
    c->send_elems.assign((size_t)n, 0);
    const bool distinct = (int)devs.size() == n;
    // CDS_COMM_FORCE_RCCL: the RCCL branch even over contexts that share a device.  Real RCCL answers duplicate devices with an error
    // (reported below); the stand-in of tests/rccl_stub accepts them -- the one-GPU rehearsal of the grouped gather with NULL receive
    // buffers on the non-roots and of the ordering of the peers' streams behind it.
    if ((distinct || getenv("CDS_COMM_FORCE_RCCL")) && !getenv("CDS_COMM_NO_RCCL")) {
        std::string why;
        c->api = load_rccl(why);
        if (!c->api) {
            delete c;
            return fail(CDS_ERR_UNSUPPORTED, why);
        }
        std::vector<int> devlist;
        for (int i = 0; i < n; i++) devlist.push_back(contexts[i]->device);
        c->comms.assign((size_t)n, nullptr);
        const int rc = c->api->CommInitAll(c->comms.data(), n, devlist.data());
        if (rc != 0) {
            const std::string msg = std::string("ncclCommInitAll: ") + c->api->GetErrorString(rc);
            delete c;
            return fail(CDS_ERR_HIP, msg);
        }
        c->rccl = true;
    }
    *out = c;
    return CDS_OK;
}

void cds_comm_destroy(cds_comm *c) {
    if (!c) return;
    for (size_t i = 0; i < c->ctxs.size(); i++) {
        hipSetDevice(c->ctxs[i]->device);
        hipStreamSynchronize(c->ctxs[i]->stream);
        if (c->d_send[i]) hipFree(c->d_send[i]);
    }
    if (c->d_recv) {
        hipSetDevice(c->ctxs[0]->device);
        hipFree(c->d_recv);
    }
    if (c->rccl)
        for (rccl_comm_t cm : c->comms)
            if (cm) c->api->CommDestroy(cm);
    delete c;
}

int32_t cds_comm_size(const cds_comm *c) { return c ? (int32_t)c->ctxs.size() : 0; }
int32_t cds_comm_uses_rccl(const cds_comm *c) { return c && c->rccl ? 1 : 0; }

int cds_search_sparse_sharded(cds_comm *c, const cds_maskset *const *masksets, const cds_sparse_library *const *slibs, int64_t *shard_offsets,
                              int32_t *scores, uint8_t *mirrored) {
    if (!c || !masksets || !slibs) return fail(CDS_ERR_INVALID_ARGUMENT, "NULL argument");
    const int n = (int)c->ctxs.size();
    int n_masks = -1;
    std::vector<int64_t> off((size_t)n + 1, 0);
    int64_t pad = 0;
    for (int r = 0; r < n; r++) {
        if (!masksets[r] || !slibs[r]) return fail(CDS_ERR_INVALID_ARGUMENT, "NULL shard");
        if (masksets[r]->ctx != c->ctxs[(size_t)r] || slibs[r]->ctx != c->ctxs[(size_t)r])
            return fail(CDS_ERR_INVALID_ARGUMENT, "shard r must have been created through context r of the communicator");
        if (n_masks < 0) n_masks = masksets[r]->n_masks;
        if (masksets[r]->n_masks != n_masks) return fail(CDS_ERR_INVALID_ARGUMENT, "the shards' mask sets differ in size (the mask set is replicated)");
        const int64_t nl = cds_sparse_library_size(slibs[r]);
        off[(size_t)r + 1] = off[(size_t)r] + nl;
        pad = std::max(pad, nl);
    }
    if (shard_offsets) memcpy(shard_offsets, off.data(), off.size() * sizeof(int64_t));
    const int64_t n_total = off[(size_t)n];
    if (n_total == 0) return CDS_OK;
    if (!scores && !mirrored) return fail(CDS_ERR_INVALID_ARGUMENT, "no output buffer");
    std::lock_guard<std::mutex> lk(c->mu);
    const size_t elems = (size_t)n_masks * (size_t)pad;  // every shard sends [n_masks][pad] (the gather wants equal counts)
    for (int r = 0; r < n; r++) {
        CDS_HIP(hipSetDevice(c->ctxs[(size_t)r]->device));
        if (c->send_elems[(size_t)r] < elems) {
            if (c->d_send[(size_t)r]) hipFree(c->d_send[(size_t)r]);
            c->d_send[(size_t)r] = nullptr;
            c->send_elems[(size_t)r] = 0;
            CDS_HIP(dev_malloc(&c->d_send[(size_t)r], elems * sizeof(int32_t)));
            c->send_elems[(size_t)r] = elems;
        }
    }
    CDS_HIP(hipSetDevice(c->ctxs[0]->device));
    if (c->recv_elems < elems * (size_t)n) {
        if (c->d_recv) hipFree(c->d_recv);
        c->d_recv = nullptr;
        c->recv_elems = 0;
        CDS_HIP(dev_malloc(&c->d_recv, elems * (size_t)n * sizeof(int32_t)));
        c->recv_elems = elems * (size_t)n;
    }
    // Once a scan has been launched, no error path may return while kernels on other GPUs still write into the communicator's
    // buffers (the caller may destroy them next): every failure below first waits for all the shards' streams.
    auto quiesce = [&]() {
        for (int r = 0; r < n; r++) {
            hipSetDevice(c->ctxs[(size_t)r]->device);
            hipStreamSynchronize(c->ctxs[(size_t)r]->stream);
        }
    };
#define MG_HIP(call)                                 \
    do {                                             \
        hipError_t _e = (call);                      \
        if (_e != hipSuccess) {                      \
            quiesce();                               \
            return cds::hip_fail(_e, #call);         \
        }                                            \
    } while (0)
    // 1. every GPU scans its own shard: asynchronous launches, all devices busy at once.  Shard r writes mask m's scores
    //    at d_send[r] + m * n_local (its own row pitch); the padding is never read back
    for (int r = 0; r < n; r++) {
        const int64_t nl = off[(size_t)r + 1] - off[(size_t)r];
        if (nl == 0) continue;
        const int rc = cds_search_sparse_device(masksets[r], slibs[r], 0, nl, c->d_send[(size_t)r]);
        if (rc) {
            const std::string msg = "shard " + std::to_string(r) + ": " + cds_last_error();
            quiesce();
            return fail(rc, msg);
        }
    }
    // 2. ONE gather to the first GPU: RCCL over xGMI (each peer -> root transfer rides its own link), stream-ordered after
    //    the scans; contexts sharing a device (rehearsal) copy device-to-device instead
    if (c->rccl) {
        int rc = c->api->GroupStart();
        for (int r = 0; r < n && rc == 0; r++) {
            hipSetDevice(c->ctxs[(size_t)r]->device);
            rc = c->api->Gather(c->d_send[(size_t)r], r == 0 ? c->d_recv : nullptr, elems, kNcclInt32, 0, c->comms[(size_t)r], c->ctxs[(size_t)r]->stream);
        }
        const int rc2 = c->api->GroupEnd();
        if (rc == 0) rc = rc2;
        if (rc != 0) {
            quiesce();
            return fail(CDS_ERR_HIP, std::string("ncclGather: ") + c->api->GetErrorString(rc));
        }
    } else {
        for (int r = 0; r < n; r++) {
            MG_HIP(hipSetDevice(c->ctxs[(size_t)r]->device));
            MG_HIP(hipStreamSynchronize(c->ctxs[(size_t)r]->stream));
            MG_HIP(hipMemcpyAsync(c->d_recv + (size_t)r * elems, c->d_send[(size_t)r], elems * sizeof(int32_t), hipMemcpyDeviceToDevice, c->ctxs[0]->stream));
        }
    }
    // 3. root -> host, then into library order: scores[m * n_total + offset_r + t]
    MG_HIP(hipSetDevice(c->ctxs[0]->device));
    std::vector<int32_t> packed(elems * (size_t)n);
    MG_HIP(hipMemcpyAsync(packed.data(), c->d_recv, packed.size() * sizeof(int32_t), hipMemcpyDeviceToHost, c->ctxs[0]->stream));
    MG_HIP(hipStreamSynchronize(c->ctxs[0]->stream));
    for (int r = 1; r < n; r++) {  // (the peers' streams: their send buffers are free again)
        MG_HIP(hipSetDevice(c->ctxs[(size_t)r]->device));
        MG_HIP(hipStreamSynchronize(c->ctxs[(size_t)r]->stream));
    }
#undef MG_HIP
    for (int r = 0; r < n; r++) {
        const int64_t nl = off[(size_t)r + 1] - off[(size_t)r];
        for (int m = 0; m < n_masks; m++) {
            const int32_t *src = packed.data() + (size_t)r * elems + (size_t)m * (size_t)nl;
            const size_t dst = (size_t)m * (size_t)n_total + (size_t)off[(size_t)r];
            for (int64_t t = 0; t < nl; t++) {
                if (scores) scores[dst + (size_t)t] = src[t] & 0x7fffffff;
                if (mirrored) mirrored[dst + (size_t)t] = (uint8_t)((uint32_t)src[t] >> 31);
            }
        }
    }
    return CDS_OK;
}

// The match lists of the shards (cds_search_sparse_match_list per GPU: scan + isMatch + compaction on the device) in library order.
// No collective: a list's length is only known to its own GPU, and n PCIe links move n lists faster than one link moves their
// concatenation -- every GPU's list goes to the host directly; the host interleaves them by mask.  One host thread per shard, because
// building a list reads its length back between its kernels.
int cds_search_sparse_sharded_matches(cds_comm *c, const cds_maskset *const *masksets, const cds_sparse_library *const *slibs,
                                      double pct_positive_pixels, int64_t *shard_offsets, cds_match *matches, int64_t capacity,
                                      int64_t *n_matches) {
    if (!c || !masksets || !slibs || !n_matches) return fail(CDS_ERR_INVALID_ARGUMENT, "NULL argument");
    *n_matches = 0;
    if (capacity < 0 || (capacity > 0 && !matches)) return fail(CDS_ERR_INVALID_ARGUMENT, "bad match buffer");
    const int n = (int)c->ctxs.size();
    int n_masks = -1;
    std::vector<int64_t> off((size_t)n + 1, 0);
    for (int r = 0; r < n; r++) {
        if (!masksets[r] || !slibs[r]) return fail(CDS_ERR_INVALID_ARGUMENT, "NULL shard");
        if (masksets[r]->ctx != c->ctxs[(size_t)r] || slibs[r]->ctx != c->ctxs[(size_t)r])
            return fail(CDS_ERR_INVALID_ARGUMENT, "shard r must have been created through context r of the communicator");
        if (n_masks < 0) n_masks = masksets[r]->n_masks;
        if (masksets[r]->n_masks != n_masks) return fail(CDS_ERR_INVALID_ARGUMENT, "the shards' mask sets differ in size (the mask set is replicated)");
        off[(size_t)r + 1] = off[(size_t)r] + cds_sparse_library_size(slibs[r]);
    }
    if (shard_offsets) memcpy(shard_offsets, off.data(), off.size() * sizeof(int64_t));
    if (off[(size_t)n] >= ((int64_t)1 << 31)) return fail(CDS_ERR_UNSUPPORTED, "more than 2^31 targets over the shards (a match holds its target in 32 bits)");
    std::lock_guard<std::mutex> lk(c->mu);
    std::vector<std::vector<cds_match>> part((size_t)n);
    std::vector<int> rcs((size_t)n, CDS_OK);
    std::vector<std::string> errs((size_t)n);
    auto work = [&](int r) {
        const int64_t nl = off[(size_t)r + 1] - off[(size_t)r];
        if (nl == 0) return;
        cds_match_list *ml = nullptr;
        int64_t cnt = 0;
        int rc = cds_search_sparse_match_list(masksets[r], slibs[r], 0, nl, pct_positive_pixels, &ml, &cnt);
        if (rc == CDS_OK) {
            try {
                part[(size_t)r].resize((size_t)cnt);
                if (cnt > 0) rc = cds_match_list_fetch(ml, 0, cnt, part[(size_t)r].data());
            } catch (const std::bad_alloc &) {
                rc = fail(CDS_ERR_OUT_OF_MEMORY, "out of host memory for a shard's match list");
            }
        }
        if (ml) cds_match_list_destroy(ml);
        rcs[(size_t)r] = rc;
        if (rc) errs[(size_t)r] = cds_last_error();   // (the message lives in this thread)
    };
    {
        // one host thread per shard; a thread the process cannot have (container limit) is replaced by the calling thread -- no
        // exception leaves through the C ABI or out of a worker
        std::vector<std::thread> pool;
        int started = 1;
        try {
            pool.reserve((size_t)n);
            for (; started < n; started++) pool.emplace_back(work, started);
        } catch (...) {
        }
        work(0);
        for (int r = started; r < n; r++) work(r);
        for (auto &t : pool) t.join();
    }
    for (int r = 0; r < n; r++)
        if (rcs[(size_t)r]) return fail(rcs[(size_t)r], "shard " + std::to_string(r) + ": " + errs[(size_t)r]);
    int64_t total = 0;
    for (auto &v : part) total += (int64_t)v.size();
    *n_matches = total;
    if (total > capacity) {
        char buf[200];
        snprintf(buf, sizeof buf, "match buffer too small: %lld matches, capacity %lld (n_matches holds the count)", (long long)total, (long long)capacity);
        return fail(CDS_ERR_OUT_OF_MEMORY, buf);
    }
    // every shard's list is ordered by mask, then target: interleave by mask, shard after shard (= by mask, then library index)
    std::vector<size_t> at((size_t)n, 0);
    int64_t o = 0;
    for (int m = 0; m < n_masks; m++)
        for (int r = 0; r < n; r++) {
            const std::vector<cds_match> &v = part[(size_t)r];
            size_t &i = at[(size_t)r];
            for (; i < v.size() && v[i].mask == m; i++) {
                matches[o] = v[i];
                matches[o].target += (int32_t)off[(size_t)r];
                o++;
            }
        }
    return CDS_OK;
}

// Per mask the k best-scoring targets over all shards: every GPU selects its own k (cds_search_sparse_topk, all GPUs at once), the host
// merges the n * k candidates per mask (score descending, equal scores by ascending library index -- the order one library would give).
int cds_search_sparse_sharded_topk(cds_comm *c, const cds_maskset *const *masksets, const cds_sparse_library *const *slibs, int32_t k,
                                   int64_t *shard_offsets, int32_t *targets, int32_t *scores) {
    if (!c || !masksets || !slibs || !targets || !scores) return fail(CDS_ERR_INVALID_ARGUMENT, "NULL argument");
    if (k < 1 || k > 1024) return fail(CDS_ERR_INVALID_ARGUMENT, "k must be 1..1024");
    const int n = (int)c->ctxs.size();
    int n_masks = -1;
    std::vector<int64_t> off((size_t)n + 1, 0);
    for (int r = 0; r < n; r++) {
        if (!masksets[r] || !slibs[r]) return fail(CDS_ERR_INVALID_ARGUMENT, "NULL shard");
        if (masksets[r]->ctx != c->ctxs[(size_t)r] || slibs[r]->ctx != c->ctxs[(size_t)r])
            return fail(CDS_ERR_INVALID_ARGUMENT, "shard r must have been created through context r of the communicator");
        if (n_masks < 0) n_masks = masksets[r]->n_masks;
        if (masksets[r]->n_masks != n_masks) return fail(CDS_ERR_INVALID_ARGUMENT, "the shards' mask sets differ in size (the mask set is replicated)");
        off[(size_t)r + 1] = off[(size_t)r] + cds_sparse_library_size(slibs[r]);
    }
    if (shard_offsets) memcpy(shard_offsets, off.data(), off.size() * sizeof(int64_t));
    if (off[(size_t)n] >= ((int64_t)1 << 31)) return fail(CDS_ERR_UNSUPPORTED, "more than 2^31 targets over the shards");
    std::lock_guard<std::mutex> lk(c->mu);
    const size_t per = (size_t)n_masks * (size_t)k;
    std::vector<std::vector<int32_t>> tg((size_t)n), sc((size_t)n);
    std::vector<int> rcs((size_t)n, CDS_OK);
    std::vector<std::string> errs((size_t)n);
    auto work = [&](int r) {
        const int64_t nl = off[(size_t)r + 1] - off[(size_t)r];
        int rc = CDS_OK;
        try {
            tg[(size_t)r].assign(per, -1);
            sc[(size_t)r].assign(per, 0);
            if (nl == 0) return;
            rc = cds_search_sparse_topk(masksets[r], slibs[r], 0, nl, k, tg[(size_t)r].data(), sc[(size_t)r].data());
        } catch (const std::bad_alloc &) {
            rc = fail(CDS_ERR_OUT_OF_MEMORY, "out of host memory for a shard's candidates");
        }
        rcs[(size_t)r] = rc;
        if (rc) errs[(size_t)r] = cds_last_error();
    };
    {
        // one host thread per shard; a thread the process cannot have (container limit) is replaced by the calling thread -- no
        // exception leaves through the C ABI or out of a worker
        std::vector<std::thread> pool;
        int started = 1;
        try {
            pool.reserve((size_t)n);
            for (; started < n; started++) pool.emplace_back(work, started);
        } catch (...) {
        }
        work(0);
        for (int r = started; r < n; r++) work(r);
        for (auto &t : pool) t.join();
    }
    for (int r = 0; r < n; r++)
        if (rcs[(size_t)r]) return fail(rcs[(size_t)r], "shard " + std::to_string(r) + ": " + errs[(size_t)r]);
    struct Cand {
        int32_t score, target, packed;
    };
    std::vector<Cand> cand;
    for (int m = 0; m < n_masks; m++) {
        cand.clear();
        for (int r = 0; r < n; r++)
            for (int i = 0; i < k; i++) {
                const int32_t t = tg[(size_t)r][(size_t)m * k + i];
                if (t < 0) break;   // (behind a shard's last candidate)
                const int32_t p = sc[(size_t)r][(size_t)m * k + i];
                cand.push_back({p & 0x7fffffff, (int32_t)(t + off[(size_t)r]), p});
            }
        std::sort(cand.begin(), cand.end(), [](const Cand &a, const Cand &b) { return a.score > b.score || (a.score == b.score && a.target < b.target); });
        for (int i = 0; i < k; i++) {
            const bool have = (size_t)i < cand.size();
            targets[(size_t)m * k + i] = have ? cand[(size_t)i].target : -1;
            scores[(size_t)m * k + i] = have ? cand[(size_t)i].packed : 0;
        }
    }
    return CDS_OK;
}

// GradientAreaGap over the shards (kernel 2, BASELINE configs[4] on several GPUs).  The reference scores every selected (mask, target)
// pair on whichever worker holds the target's partition (Spark: SparkColorMIPSearchProcessor.java:50-73; per pair
// CalculateGradientScoresCmd.java:541-586 -> Shape2DMatchColorDepthSearchAlgorithm.calculateMatchingScore).  Here shard r keeps library
// indices [shard_offsets[r], shard_offsets[r + 1]) in slots 0 .. size_r - 1 of its dense library (targets + gradient / zgap variants),
// the shape set is replicated (one per GPU), and a pair goes to the GPU that owns its target: ONE host thread sorts the pairs by owner,
// starts every GPU's scoring call (pair upload, kernel, read-back: all asynchronous on that GPU's stream), then collects them in shard
// order and puts the results back in pair order.  No collective: a result is 17 bytes and every GPU hands its own to the host.
int cds_gradscore_sharded(cds_comm *c, const cds_shapeset *const *shapesets, const cds_library *const *libs, const int64_t *shard_sizes,
                          int64_t n_pairs, const int32_t *pair_mask, const int64_t *pair_target, int64_t *shard_offsets,
                          int64_t *gradient_area_gap, int64_t *high_expression_area, uint8_t *mirrored) {
    if (!c || !shapesets || !libs) return fail(CDS_ERR_INVALID_ARGUMENT, "NULL argument");
    if (n_pairs < 0) return fail(CDS_ERR_INVALID_ARGUMENT, "n_pairs < 0");
    const int n = (int)c->ctxs.size();
    int n_masks = -1;
    std::vector<int64_t> off((size_t)n + 1, 0);
    for (int r = 0; r < n; r++) {
        if (!shapesets[r] || !libs[r]) return fail(CDS_ERR_INVALID_ARGUMENT, "NULL shard");
        if (shapeset_context(shapesets[r]) != c->ctxs[(size_t)r] || libs[r]->ctx != c->ctxs[(size_t)r])
            return fail(CDS_ERR_INVALID_ARGUMENT, "shard r must have been created through context r of the communicator");
        if (n_masks < 0) n_masks = shapeset_n_masks(shapesets[r]);
        if (shapeset_n_masks(shapesets[r]) != n_masks) return fail(CDS_ERR_INVALID_ARGUMENT, "the shards' shape sets differ in size (the mask set is replicated)");
        const int64_t nl = shard_sizes ? shard_sizes[r] : libs[r]->capacity;
        if (nl < 0 || nl > libs[r]->capacity) return fail(CDS_ERR_INVALID_ARGUMENT, "a shard size exceeds its library's capacity");
        off[(size_t)r + 1] = off[(size_t)r] + nl;
    }
    if (shard_offsets) memcpy(shard_offsets, off.data(), off.size() * sizeof(int64_t));
    if (n_pairs == 0) return CDS_OK;
    if (!pair_mask || !pair_target || !gradient_area_gap || !high_expression_area || !mirrored) return fail(CDS_ERR_INVALID_ARGUMENT, "NULL argument");
    // 1. route: pair i -> the shard whose index range holds its target (empty shards own nothing)
    std::vector<std::vector<int32_t>> pm((size_t)n);
    std::vector<std::vector<int64_t>> ps((size_t)n), at((size_t)n);
    for (int64_t i = 0; i < n_pairs; i++) {
        const int64_t t = pair_target[i];
        if (t < 0 || t >= off[(size_t)n]) return fail(CDS_ERR_INVALID_ARGUMENT, "pair target outside the sharded library");
        if (pair_mask[i] < 0 || pair_mask[i] >= n_masks) return fail(CDS_ERR_INVALID_ARGUMENT, "pair mask index out of range");
        const int r = (int)(std::upper_bound(off.begin() + 1, off.end(), t) - (off.begin() + 1));
        pm[(size_t)r].push_back(pair_mask[i]);
        ps[(size_t)r].push_back(t - off[(size_t)r]);
        at[(size_t)r].push_back(i);
    }
    std::lock_guard<std::mutex> lk(c->mu);
    // 2. every GPU scores its own pairs: all calls are started before the first one is waited for
    std::vector<GradscoreJob *> jobs((size_t)n, nullptr);
    int rc = CDS_OK;
    std::string err;
    for (int r = 0; r < n && rc == CDS_OK; r++) {
        if (pm[(size_t)r].empty()) continue;
        rc = gradscore_begin(shapesets[r], libs[r], (int64_t)pm[(size_t)r].size(), pm[(size_t)r].data(), ps[(size_t)r].data(), nullptr, &jobs[(size_t)r]);
        if (rc) err = "shard " + std::to_string(r) + ": " + cds_last_error();
    }
    // 3. collect in shard order, results back into pair order.  After a failure the other GPUs' calls are still waited for (their
    //    kernels write into buffers the jobs own), only their results are dropped.
    std::vector<int64_t> g, h;
    std::vector<uint8_t> mr;
    for (int r = 0; r < n; r++) {
        GradscoreJob *j = jobs[(size_t)r];
        if (!j) continue;
        const size_t k = pm[(size_t)r].size();
        if (rc != CDS_OK) {
            gradscore_end(j, nullptr, nullptr, nullptr);
            continue;
        }
        g.resize(k), h.resize(k), mr.resize(k);
        const int rc2 = gradscore_end(j, g.data(), h.data(), mr.data());
        if (rc2) {
            rc = rc2;
            err = "shard " + std::to_string(r) + ": " + cds_last_error();
            continue;
        }
        for (size_t q = 0; q < k; q++) {
            const int64_t i = at[(size_t)r][q];
            gradient_area_gap[i] = g[q], high_expression_area[i] = h[q], mirrored[i] = mr[q];
        }
    }
    if (rc) return fail(rc, err);
    return CDS_OK;
}

}  // extern "C"
