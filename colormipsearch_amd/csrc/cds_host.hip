// cds_host.hip -- host side of libcds_hip.so: context, resident library, pixel-match mask preparation
// (the exact integer restatement of the reference's double predicate) and the search entry points.
//
// Reference citations: API/ = colormipsearch-api/src/main/java/org/janelia/colormipsearch/.
#include <algorithm>
#include <atomic>
#include <thread>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <unordered_map>

#include "cds_internal.hpp"

namespace cds {

static thread_local std::string g_err;
void set_error(const std::string &msg) { g_err = msg; }
int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}
int hip_fail(hipError_t e, const char *what) {
    g_err = std::string("HIP error ") + hipGetErrorName(e) + " (" + hipGetErrorString(e) + ") in " + what;
    return e == hipErrorOutOfMemory ? CDS_ERR_OUT_OF_MEMORY : CDS_ERR_HIP;
}

// ------------------------------------------------------------------------------------------------
// Ratio table.  A target pixel's hue ratio is second/max = n/d with 0 <= n < d <= 255; the reference
// evaluates it as (double)n/(double)d (0.0 when n == 0: AbstractColorDepthSearchAlgorithm.java:195-256).
// The device encodes it as bits(fl32(n) * rcp[d]) >> 1 with rcp[] = this GPU's v_rcp_f32.  The table
// lists every distinct double value, sorted, with the code range of its representatives, and proves
// that the code order equals the value order (so value intervals are code intervals).
// ------------------------------------------------------------------------------------------------
static inline uint32_t fbits(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}
uint32_t ratio_code(const float *rcp, int n, int d) {
    volatile float x = (float)n * rcp[d];  // one IEEE single multiply, as v_mul_f32 / v_pk_mul_f32
    return fbits(x) >> 1;
}

int build_ratio_table(const float *rcp, std::vector<RatioEntry> &out) {
    struct Item {
        double v;
        uint32_t c;
    };
    std::vector<Item> items;
    items.reserve(32640);
    for (int d = 1; d <= 255; d++)
        for (int n = 0; n < d; n++) {
            double v = n == 0 ? 0.0 : (double)n / (double)d;
            items.push_back({v, ratio_code(rcp, n, d)});
        }
    std::sort(items.begin(), items.end(), [](const Item &a, const Item &b) { return a.v < b.v || (a.v == b.v && a.c < b.c); });
    out.clear();
    for (const Item &it : items) {
        if (!out.empty() && out.back().value == it.v) {
            out.back().code_hi = std::max(out.back().code_hi, it.c);
            out.back().code_lo = std::min(out.back().code_lo, it.c);
        } else {
            out.push_back({it.v, it.c, it.c});
        }
    }
    for (size_t i = 1; i < out.size(); i++)
        if (!(out[i - 1].code_hi < out[i].code_lo))
            // (the message carries what a report needs: the first offending pair and the head of the table -- round 4's one refusal of
            // this kind, `s16u` in profiles/README.md, could not be told from a table that was never written)
            return failf(CDS_ERR_UNSUPPORTED,
                         "device reciprocal table does not preserve the order of hue ratios: ratio %.9g has codes [%u, %u] but the next ratio %.9g "
                         "has [%u, %u]; rcp[1..4] = %.9g %.9g %.9g %.9g",
                         out[i - 1].value, out[i - 1].code_lo, out[i - 1].code_hi, out[i].value, out[i].code_lo, out[i].code_hi,
                         (double)rcp[1], (double)rcp[2], (double)rcp[3], (double)rcp[4]);
    if (out.back().code_hi >= (1u << 29)) return fail(CDS_ERR_UNSUPPORTED, "hue ratio code overflow");
    return CDS_OK;
}

// Hue class + ratio of a colour exactly as the reference (AbstractColorDepthSearchAlgorithm.java:195-225).
enum { C_NONE = 0, C_BR, C_BG, C_GB, C_GR, C_RG, C_RB };
static int hue_class(int r, int g, int b, double *ratio) {
    *ratio = 0.0;
    if (b > r && b > g) {
        if (r > g) { *ratio = (double)r / (double)b; return C_BR; }
        if (g != 0) *ratio = (double)g / (double)b;
        return C_BG;
    }
    if (g > b && g > r) {
        if (b > r) { *ratio = (double)b / (double)g; return C_GB; }
        if (r != 0) *ratio = (double)r / (double)g;
        return C_GR;
    }
    if (r > b && r > g) {
        if (g > b) { *ratio = (double)g / (double)r; return C_RG; }
        if (b != 0) *ratio = (double)b / (double)r;
        return C_RB;
    }
    return C_NONE;
}

struct Intervals {
    uint32_t lo1, len1, lo2, len2;
};
static const Intervals kEmptyIv = {1u, 0u, 1u, 0u};  // code 1 never occurs

// The reference's adjacent-class rules (AbstractColorDepthSearchAlgorithm.java:269-381), from the point
// of view of a mask pixel of class c1 with ratio a: which neighbour class can match, and how.
struct Neighbour {
    int cls;      // target class
    bool low;     // true: (a-c)+(b-c) with a<la && b<lb;  false: (c-a)+(c-b) with a>la && b>lb
    double la, lb, c;
};
static int neighbours_of(int c1, Neighbour nb[2]) {
    const double BrBg = 0.354862745, BgGb = 0.996078431, GbGr = 0.505882353, GrRg = 0.996078431, RgRb = 0.505882353;
    switch (c1) {
        case C_BR: nb[0] = {C_BG, true, 0.44, 0.54, BrBg}; return 1;
        case C_BG: nb[0] = {C_BR, true, 0.54, 0.44, BrBg}; nb[1] = {C_GB, false, 0.8, 0.8, BgGb}; return 2;
        case C_GB: nb[0] = {C_BG, false, 0.8, 0.8, BgGb}; nb[1] = {C_GR, true, 0.7, 0.7, GbGr}; return 2;
        case C_GR: nb[0] = {C_GB, true, 0.7, 0.7, GbGr}; nb[1] = {C_RG, false, 0.8, 0.8, GrRg}; return 2;
        case C_RG: nb[0] = {C_GR, false, 0.8, 0.8, GrRg}; nb[1] = {C_RB, true, 0.7, 0.7, RgRb}; return 2;
        case C_RB: nb[0] = {C_RG, true, 0.7, 0.7, RgRb}; return 1;
    }
    return 0;
}

static inline bool same_class_ok(double a, double b, double ztol) {
    if (!(a > 0 && b > 0)) return false;
    double gap = (a != b) ? std::fabs(b - a) : 0.0;
    return gap <= ztol;
}
static inline bool neighbour_ok(const Neighbour &n, double a, double b, double ztol) {
    if (n.low) {
        if (!(a < n.la && b < n.lb)) return false;
        volatile double ga = a - n.c, gb = b - n.c;  // no contraction / re-association
        volatile double gap = ga + gb;
        return gap <= ztol;
    }
    if (!(a > n.la && b > n.lb)) return false;
    volatile double ga = n.c - a, gb = n.c - b;
    volatile double gap = ga + gb;
    return gap <= ztol;
}

// Acceptance intervals (in device-code space) of a mask colour of class c1 / ratio a.
// Every predicate below is monotone along the sorted ratio table (IEEE subtraction, addition and
// rounding are monotone), so the accepted set is one index range; binary search finds its ends with
// the reference's own double expressions as the oracle of each probe.
Intervals intervals_for(const std::vector<RatioEntry> &tab, int c1, double a, double ztol) {
    Intervals iv = kEmptyIv;
    if (c1 == C_NONE) return iv;
    const int n = (int)tab.size();
    // --- same class: a > 0, b > 0, |b - a| <= ztol
    if (a > 0) {
        int ia = (int)(std::lower_bound(tab.begin(), tab.end(), a, [](const RatioEntry &e, double v) { return e.value < v; }) - tab.begin());
        if (ia < n && tab[ia].value == a && same_class_ok(a, a, ztol)) {
            int lo = 1, hi = ia;  // index 0 is the value 0.0
            while (lo < hi) {     // smallest j in [1, ia] accepted
                int mid = (lo + hi) / 2;
                if (same_class_ok(a, tab[mid].value, ztol)) hi = mid; else lo = mid + 1;
            }
            int jlo = lo;
            lo = ia; hi = n - 1;
            while (lo < hi) {     // largest j in [ia, n-1] accepted
                int mid = (lo + hi + 1) / 2;
                if (same_class_ok(a, tab[mid].value, ztol)) lo = mid; else hi = mid - 1;
            }
            int jhi = lo;
            iv.lo1 = ((uint32_t)c1 << 29) | tab[jlo].code_lo;
            iv.len1 = (((uint32_t)c1 << 29) | tab[jhi].code_hi) - iv.lo1;
        }
    }
    // --- at most one adjacent class (the two conditions on a exclude each other)
    Neighbour nb[2];
    int k = neighbours_of(c1, nb);
    for (int i = 0; i < k; i++) {
        const Neighbour &q = nb[i];
        if (q.low) {
            if (!neighbour_ok(q, a, tab[0].value, ztol)) continue;
            int lo = 0, hi = n - 1;
            while (lo < hi) {  // largest accepted index (accepted set is down-closed)
                int mid = (lo + hi + 1) / 2;
                if (neighbour_ok(q, a, tab[mid].value, ztol)) lo = mid; else hi = mid - 1;
            }
            iv.lo2 = ((uint32_t)q.cls << 29) | tab[0].code_lo;
            iv.len2 = (((uint32_t)q.cls << 29) | tab[lo].code_hi) - iv.lo2;
            break;
        } else {
            if (!neighbour_ok(q, a, tab[n - 1].value, ztol)) continue;
            int lo = 0, hi = n - 1;
            while (lo < hi) {  // smallest accepted index (accepted set is up-closed)
                int mid = (lo + hi) / 2;
                if (neighbour_ok(q, a, tab[mid].value, ztol)) hi = mid; else lo = mid + 1;
            }
            iv.lo2 = ((uint32_t)q.cls << 29) | tab[lo].code_lo;
            iv.len2 = (((uint32_t)q.cls << 29) | tab[n - 1].code_hi) - iv.lo2;
            break;
        }
    }
    return iv;
}

// Host model of the device's hue_code() (cds_scan.hip) -- used only by the diagnostics entry points.
uint32_t host_hue_code(const float *rcp, int r, int g, int b, int thr) {
    int mx = std::max(r, std::max(g, b));
    int mn = std::min(r, std::min(g, b));
    int md = r + g + b - mx - mn;
    uint32_t idx = (r > g ? 4u : 0u) | (g > b ? 8u : 0u) | (b > r ? 16u : 0u);
    uint32_t cls = (0x03125460u >> idx) & 7u;
    if (!(mx > thr && mx > md)) return 0;
    return (cls << 29) | ratio_code(rcp, md, mx);
}

cds_context *clone_context_for_slot(cds_context *parent) {
    hipSetDevice(parent->device);
    cds_context *c = new cds_context();
    c->layout_stamp = sizeof(cds_context);
    c->device = parent->device;
    c->n_cus = parent->n_cus;
    c->lds_per_block = parent->lds_per_block;
    memcpy(c->rcp, parent->rcp, sizeof c->rcp);
    c->ratios = parent->ratios;
    c->d_slice_lut = parent->d_slice_lut;
    c->owns_lut = false;
    if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return nullptr;
    }
    c->stream = c->own_stream;
    c->upload_stream = nullptr;   // (a pair-slot context prepares no mask sets)
    register_context(c);
    return c;
}

PairSlot::~PairSlot() {
    if (lib) cds_library_destroy(lib);
    if (d_scores) hipFree(d_scores);
    if (sub) cds_context_destroy(sub);
}

int PairSlotPool::acquire(cds_context *parent, int W, int H, size_t score_elems, PairSlot **out) {
    int max_slots = 8;
    if (const char *e = getenv("CDS_PAIR_SLOTS")) max_slots = std::max(1, std::min(64, atoi(e)));
    for (;;) {
        size_t n;
        {
            std::lock_guard<std::mutex> lk(grow_mu);
            n = slots.size();
            for (size_t i = 0; i < n; i++)
                if (slots[i]->mu.try_lock()) {
                    *out = slots[i].get();
                    return CDS_OK;
                }
            if ((int)n < max_slots) {  // every slot is busy: one more
                std::unique_ptr<PairSlot> s(new PairSlot());
                s->sub = clone_context_for_slot(parent);
                if (!s->sub) return fail(CDS_ERR_HIP, "hipStreamCreate (pair slot)");
                int rc = cds_library_create(s->sub, W, H, 1, &s->lib);
                if (rc) return rc;
                if (score_elems) {
                    hipError_t e = dev_malloc(&s->d_scores, score_elems * sizeof(int32_t));
                    if (e != hipSuccess) return hip_fail(e, "hipMalloc(pair scores)");
                }
                s->mu.lock();
                *out = s.get();
                slots.push_back(std::move(s));
                return CDS_OK;
            }
        }
        // the pool is full and busy: wait for one slot (round robin over the callers)
        PairSlot *s = slots[next.fetch_add(1) % n].get();
        s->mu.lock();
        *out = s;
        return CDS_OK;
    }
}

static bool in_rects(const cds_rect *r, int n, int x, int y) {
    for (int i = 0; i < n; i++)
        if (x >= r[i].x0 && x < r[i].x1 && y >= r[i].y0 && y < r[i].y1) return true;
    return false;
}

}  // namespace cds

using namespace cds;

// =================================================================================================
//                                            C ABI
// =================================================================================================
extern "C" {

const char *cds_last_error(void) { return g_err.c_str(); }
const char *cds_version(void) { return "cds-hip 0.1 (gfx950)"; }

int cds_context_create(int device_id, cds_context **out) {
    if (!out) return fail(CDS_ERR_INVALID_ARGUMENT, "out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return fail(CDS_ERR_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
    if (device_id < 0 || device_id >= n) return fail(CDS_ERR_INVALID_ARGUMENT, "device_id out of range");
    hipDeviceProp_t prop;
    CDS_HIP(hipGetDeviceProperties(&prop, device_id));
    if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos)
        return fail(CDS_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 (MI355X) only");
    CDS_HIP(hipSetDevice(device_id));
    cds_context *ctx = new cds_context();
    ctx->layout_stamp = sizeof(cds_context);
    ctx->device = device_id;
    if (const char *e = getenv("CDS_CACHE_LIMIT_MB")) ctx->cache_cap = (size_t)std::max(0ll, atoll(e)) << 20;
    ctx->n_cus = prop.multiProcessorCount;
    ctx->lds_per_block = prop.sharedMemPerBlock;
    if (ctx->lds_per_block < 160 * 1024) ctx->lds_per_block = 160 * 1024;  // gfx950: 160 KiB per workgroup
    hipError_t se = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking);
    if (se != hipSuccess) {
        delete ctx;
        return hip_fail(se, "hipStreamCreate");
    }
    ctx->stream = ctx->own_stream;
    se = hipStreamCreateWithFlags(&ctx->upload_stream, hipStreamNonBlocking);
    if (se != hipSuccess) {
        hipStreamDestroy(ctx->own_stream);
        delete ctx;
        return hip_fail(se, "hipStreamCreate");
    }
    int rc = init_rcp_table(ctx);
    if (rc == CDS_OK) rc = build_ratio_table(ctx->rcp, ctx->ratios);
    if (rc != CDS_OK) {
        hipStreamDestroy(ctx->own_stream);
        hipStreamDestroy(ctx->upload_stream);
        delete ctx;
        return rc;
    }
    cds::register_context(ctx);
    *out = ctx;
    return CDS_OK;
}

}  // extern "C"
namespace cds {
hipError_t cache_alloc(cds_context *ctx, void **p, size_t bytes) {
    bytes = std::max<size_t>(bytes, 256);
    {
        std::lock_guard<std::mutex> lk(ctx->cache_mu);
        size_t best = SIZE_MAX, at = 0;
        for (size_t i = 0; i < ctx->cache_idle.size(); i++) {
            const size_t b = ctx->cache_idle[i].second;
            if (b >= bytes && b <= 2 * bytes + ((size_t)1 << 20) && b < best) best = b, at = i;
        }
        if (best != SIZE_MAX) {
            *p = ctx->cache_idle[at].first;
            ctx->cache_idle.erase(ctx->cache_idle.begin() + (std::ptrdiff_t)at);
            ctx->cache_idle_bytes -= best;
            return hipSuccess;
        }
    }
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) {   // out of memory with idle blocks lying around: give them back and try once more
        cache_drop(ctx);
        (void)hipGetLastError();
        e = hipMalloc(p, bytes);
    }
    if (e == hipSuccess) {
        std::lock_guard<std::mutex> lk(ctx->cache_mu);
        ctx->cache_sizes[*p] = bytes;
    }
    return e;
}
void cache_free(cds_context *ctx, void *p) {
    if (!p) return;
    {
        std::lock_guard<std::mutex> lk(ctx->cache_mu);
        auto it = ctx->cache_sizes.find(p);
        if (it != ctx->cache_sizes.end() && ctx->cache_idle_bytes + it->second <= ctx->cache_cap) {
            ctx->cache_idle.push_back({p, it->second});
            ctx->cache_idle_bytes += it->second;
            return;
        }
        if (it != ctx->cache_sizes.end()) ctx->cache_sizes.erase(it);
    }
    hipFree(p);
}
static std::mutex g_ctx_mu;
static std::vector<cds_context *> g_contexts;
void register_context(cds_context *ctx) {
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    g_contexts.push_back(ctx);
}
void unregister_context(cds_context *ctx) {
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    g_contexts.erase(std::remove(g_contexts.begin(), g_contexts.end(), ctx), g_contexts.end());
}
hipError_t dev_malloc_raw(void **p, size_t bytes) {
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipErrorOutOfMemory) return e;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return e;
    size_t dropped = 0;
    {
        std::lock_guard<std::mutex> lk(g_ctx_mu);   // (a context is unregistered before it is deleted: the pointers are live while we hold this)
        for (cds_context *c : g_contexts)
            if (c->device == dev) {
                dropped += c->cache_idle_bytes;
                cache_drop(c);
            }
    }
    if (!dropped) return e;
    (void)hipGetLastError();
    return hipMalloc(p, bytes);
}
void cache_drop(cds_context *ctx) {
    std::vector<std::pair<void *, size_t>> idle;
    {
        std::lock_guard<std::mutex> lk(ctx->cache_mu);
        idle.swap(ctx->cache_idle);
        ctx->cache_idle_bytes = 0;
        for (auto &b : idle) ctx->cache_sizes.erase(b.first);
    }
    for (auto &b : idle) hipFree(b.first);
}
}  // namespace cds
extern "C" {

int cds_context_trim(cds_context *ctx, uint64_t *freed_bytes) {
    if (!ctx) return fail(CDS_ERR_INVALID_ARGUMENT, "context is NULL");
    CDS_HIP(hipSetDevice(ctx->device));
    CDS_HIP(hipDeviceSynchronize());   // (a released block may still be read by work enqueued before its release)
    size_t idle;
    {
        std::lock_guard<std::mutex> lk(ctx->cache_mu);
        idle = ctx->cache_idle_bytes;
    }
    cds::cache_drop(ctx);
    if (freed_bytes) *freed_bytes = idle;
    return CDS_OK;
}

int cds_context_set_cache_limit(cds_context *ctx, uint64_t bytes) {
    if (!ctx) return fail(CDS_ERR_INVALID_ARGUMENT, "context is NULL");
    bool over;
    {
        std::lock_guard<std::mutex> lk(ctx->cache_mu);
        ctx->cache_cap = (size_t)bytes;
        over = ctx->cache_idle_bytes > ctx->cache_cap;
    }
    return over ? cds_context_trim(ctx, nullptr) : CDS_OK;
}

void cds_context_destroy(cds_context *ctx) {
    if (!ctx) return;
    cds::unregister_context(ctx);
    hipSetDevice(ctx->device);
    cds::cache_drop(ctx);
    if (ctx->d_slice_lut && ctx->owns_lut) hipFree(ctx->d_slice_lut);
    if (ctx->own_stream) hipStreamDestroy(ctx->own_stream);
    if (ctx->upload_stream) hipStreamDestroy(ctx->upload_stream);
    delete ctx;
}

int cds_context_set_stream(cds_context *ctx, void *hip_stream) {
    if (!ctx) return fail(CDS_ERR_INVALID_ARGUMENT, "ctx is NULL");
    ctx->stream = (hipStream_t)hip_stream;
    return CDS_OK;
}

int cds_context_reset_stream(cds_context *ctx) {
    if (!ctx) return fail(CDS_ERR_INVALID_ARGUMENT, "ctx is NULL");
    ctx->stream = ctx->own_stream;
    return CDS_OK;
}

int cds_context_synchronize(cds_context *ctx) {
    if (!ctx) return fail(CDS_ERR_INVALID_ARGUMENT, "ctx is NULL");
    CDS_HIP(hipSetDevice(ctx->device));
    CDS_HIP(hipStreamSynchronize(ctx->stream));
    return CDS_OK;
}

// ---- library ------------------------------------------------------------------------------------
size_t cds_library_bytes(int32_t W, int32_t H, int64_t capacity) {
    if (W <= 0 || H <= 0 || capacity <= 0) return 0;
    return (size_t)capacity * (size_t)H * (size_t)rgb_pitch(W);
}

static int library_init(cds_context *ctx, int W, int H, int64_t capacity, cds_library **out) {
    if (!ctx || !out) return fail(CDS_ERR_INVALID_ARGUMENT, "NULL argument");
    if (!image_size_ok(W, H)) return fail(CDS_ERR_UNSUPPORTED, kImageSizeMessage);
    if (capacity <= 0) return fail(CDS_ERR_INVALID_ARGUMENT, "capacity must be positive");
    cds_library *lib = new cds_library();
    lib->ctx = ctx; lib->W = W; lib->H = H; lib->capacity = capacity;
    lib->pitch = rgb_pitch(W);
    lib->img_stride = (size_t)H * lib->pitch;
    lib->gpitch = grad_pitch(W);
    lib->grad_stride = (size_t)H * lib->gpitch;
    lib->has_grad.assign((size_t)capacity, 0);
    lib->has_zgap.assign((size_t)capacity, 0);
    lib->zslice_ok.assign((size_t)capacity, 0);
    *out = lib;
    return CDS_OK;
}

int cds_library_create(cds_context *ctx, int32_t W, int32_t H, int64_t capacity, cds_library **out) {
    cds_library *lib = nullptr;
    int rc = library_init(ctx, W, H, capacity, &lib);
    if (rc) return rc;
    hipSetDevice(ctx->device);
    size_t bytes = cds_library_bytes(W, H, capacity);
    hipError_t e = dev_malloc(&lib->d_rgb, bytes);
    if (e == hipSuccess) e = hipMemsetAsync(lib->d_rgb, 0, bytes, ctx->stream);
    if (e != hipSuccess) {
        if (lib->d_rgb) hipFree(lib->d_rgb);
        delete lib;
        return hip_fail(e, "hipMalloc(library)");
    }
    lib->owns_rgb = true;
    *out = lib;
    return CDS_OK;
}

int cds_library_wrap(cds_context *ctx, int32_t W, int32_t H, int64_t capacity, void *device_ptr, size_t device_bytes,
                     cds_library **out) {
    if (!device_ptr) return fail(CDS_ERR_INVALID_ARGUMENT, "device_ptr is NULL");
    if (((uintptr_t)device_ptr & 15) != 0) return fail(CDS_ERR_INVALID_ARGUMENT, "device_ptr must be 16-byte aligned");
    if (device_bytes < cds_library_bytes(W, H, capacity)) return fail(CDS_ERR_INVALID_ARGUMENT, "device buffer too small");
    cds_library *lib = nullptr;
    int rc = library_init(ctx, W, H, capacity, &lib);
    if (rc) return rc;
    lib->d_rgb = (uint8_t *)device_ptr;
    lib->owns_rgb = false;
    *out = lib;
    return CDS_OK;
}

int cds_library_dims(const cds_library *lib, int32_t *W, int32_t *H, int64_t *capacity) {
    if (!lib) return fail(CDS_ERR_INVALID_ARGUMENT, "NULL argument");
    if (W) *W = lib->W;
    if (H) *H = lib->H;
    if (capacity) *capacity = lib->capacity;
    return CDS_OK;
}

void cds_library_destroy(cds_library *lib) {
    if (!lib) return;
    hipSetDevice(lib->ctx->device);
    hipStreamSynchronize(lib->ctx->stream);
    if (lib->owns_rgb && lib->d_rgb) hipFree(lib->d_rgb);
    if (lib->d_grad) hipFree(lib->d_grad);
    if (lib->d_zgap) hipFree(lib->d_zgap);
    if (lib->d_zslice) hipFree(lib->d_zslice);
    if (lib->d_stage) hipFree(lib->d_stage);
    delete lib;
}

static int check_range(const cds_library *lib, int64_t first, int64_t count) {
    if (!lib) return fail(CDS_ERR_INVALID_ARGUMENT, "library is NULL");
    if (first < 0 || count < 0 || first + count > lib->capacity) return fail(CDS_ERR_INVALID_ARGUMENT, "slot range outside the library");
    return CDS_OK;
}

// Host <-> pitched device rows through a contiguous staging buffer kept with the library: ONE transfer per piece (<= 64 MB)
// plus a device-side 2-D copy.  (A host-side 2-D copy of pageable memory degenerates into one small transfer per image
// row: 4 ms for a single 1210x566 image instead of 0.2 ms.)
static int stage_reserve(cds_library *lib, size_t bytes) {
    if (lib->stage_bytes >= bytes) return CDS_OK;
    if (lib->d_stage) hipFree(lib->d_stage);
    lib->d_stage = nullptr; lib->stage_bytes = 0;
    CDS_HIP(dev_malloc(&lib->d_stage, bytes));
    lib->stage_bytes = bytes;
    return CDS_OK;
}

static int rows_h2d(cds_library *lib, uint8_t *dst, size_t dpitch, const uint8_t *src, size_t row_bytes, size_t rows) {
    const size_t piece_rows = std::max<size_t>(1, ((size_t)64 << 20) / row_bytes);
    std::lock_guard<std::mutex> lk(lib->stage_mu);
    int rc = stage_reserve(lib, std::min(piece_rows, rows) * row_bytes);
    if (rc) return rc;
    for (size_t done = 0; done < rows; done += piece_rows) {
        const size_t n = std::min(piece_rows, rows - done);
        CDS_HIP(hipMemcpyAsync(lib->d_stage, src + done * row_bytes, n * row_bytes, hipMemcpyHostToDevice, lib->ctx->stream));
        CDS_HIP(hipMemcpy2DAsync(dst + done * dpitch, dpitch, lib->d_stage, row_bytes, row_bytes, n, hipMemcpyDeviceToDevice, lib->ctx->stream));
        CDS_HIP(hipStreamSynchronize(lib->ctx->stream));  // the staging buffer is reused by the next piece
    }
    return CDS_OK;
}

static int rows_d2h(cds_library *lib, uint8_t *dst, const uint8_t *src, size_t spitch, size_t row_bytes, size_t rows) {
    const size_t piece_rows = std::max<size_t>(1, ((size_t)64 << 20) / row_bytes);
    std::lock_guard<std::mutex> lk(lib->stage_mu);
    int rc = stage_reserve(lib, std::min(piece_rows, rows) * row_bytes);
    if (rc) return rc;
    for (size_t done = 0; done < rows; done += piece_rows) {
        const size_t n = std::min(piece_rows, rows - done);
        CDS_HIP(hipMemcpy2DAsync(lib->d_stage, row_bytes, src + done * spitch, spitch, row_bytes, n, hipMemcpyDeviceToDevice, lib->ctx->stream));
        CDS_HIP(hipMemcpyAsync(dst + done * row_bytes, lib->d_stage, n * row_bytes, hipMemcpyDeviceToHost, lib->ctx->stream));
        CDS_HIP(hipStreamSynchronize(lib->ctx->stream));
    }
    return CDS_OK;
}

int cds_library_upload(cds_library *lib, int64_t first, int64_t count, const uint8_t *rgb) {
    int rc = check_range(lib, first, count);
    if (rc) return rc;
    if (count == 0) return CDS_OK;
    if (!rgb) return fail(CDS_ERR_INVALID_ARGUMENT, "rgb is NULL");
    CDS_HIP(hipSetDevice(lib->ctx->device));
    // rows of consecutive images are consecutive rows of one tall pitched image; the padding columns of a row must be
    // black (the kernels treat them as off-image pixels), also in caller-owned memory (cds_library_wrap)
    if (lib->pitch > 3 * lib->W)
        CDS_HIP(hipMemset2DAsync(lib->d_rgb + (size_t)first * lib->img_stride + (size_t)3 * lib->W, lib->pitch, 0,
                                 (size_t)(lib->pitch - 3 * lib->W), (size_t)lib->H * count, lib->ctx->stream));
    return rows_h2d(lib, lib->d_rgb + (size_t)first * lib->img_stride, lib->pitch, rgb, (size_t)3 * lib->W, (size_t)lib->H * count);
}

int cds_library_download(cds_library *lib, int64_t first, int64_t count, uint8_t *rgb) {
    int rc = check_range(lib, first, count);
    if (rc) return rc;
    if (count == 0) return CDS_OK;
    if (!rgb) return fail(CDS_ERR_INVALID_ARGUMENT, "rgb is NULL");
    CDS_HIP(hipSetDevice(lib->ctx->device));
    return rows_d2h(lib, rgb, lib->d_rgb + (size_t)first * lib->img_stride, lib->pitch, (size_t)3 * lib->W, (size_t)lib->H * count);
}

int cds_library_ensure_variants(cds_library *lib) {
    std::lock_guard<std::mutex> lk(lib->mu);
    CDS_HIP(hipSetDevice(lib->ctx->device));
    if (!lib->d_grad) {
        size_t bytes = (size_t)lib->capacity * lib->grad_stride;
        CDS_HIP(dev_malloc(&lib->d_grad, bytes));
        CDS_HIP(hipMemsetAsync(lib->d_grad, 0, bytes, lib->ctx->stream));
    }
    if (!lib->d_zgap) {
        size_t bytes = (size_t)lib->capacity * lib->img_stride;
        CDS_HIP(dev_malloc(&lib->d_zgap, bytes));
        CDS_HIP(hipMemsetAsync(lib->d_zgap, 0, bytes, lib->ctx->stream));
    }
    // (the derived zslice plane is allocated by the first scoring call that wants it: cds_shape.hip, ensure_zslice)
    return CDS_OK;
}

int cds_library_upload_variants(cds_library *lib, int64_t first, int64_t count, const uint16_t *grad,
                                const uint8_t *zgap_rgb) {
    int rc = check_range(lib, first, count);
    if (rc) return rc;
    if (count == 0) return CDS_OK;
    rc = cds_library_ensure_variants(lib);
    if (rc) return rc;
    if (grad) {
        rc = rows_h2d(lib, (uint8_t *)lib->d_grad + (size_t)first * lib->grad_stride, lib->gpitch, (const uint8_t *)grad,
                      (size_t)2 * lib->W, (size_t)lib->H * count);
        if (rc) return rc;
        for (int64_t i = 0; i < count; i++) lib->has_grad[(size_t)(first + i)] = 1;
    }
    if (zgap_rgb) {
        rc = rows_h2d(lib, lib->d_zgap + (size_t)first * lib->img_stride, lib->pitch, zgap_rgb, (size_t)3 * lib->W,
                      (size_t)lib->H * count);
        if (rc) return rc;
        zgap_written(lib, first, count);
    }
    CDS_HIP(hipStreamSynchronize(lib->ctx->stream));
    return CDS_OK;
}

int cds_library_upload_gradient_u8(cds_library *lib, int64_t first, int64_t count, const uint8_t *grad8) {
    int rc = check_range(lib, first, count);
    if (rc) return rc;
    if (count == 0) return CDS_OK;
    if (!grad8) return fail(CDS_ERR_INVALID_ARGUMENT, "grad8 is NULL");
    // widened on the host, a batch of images at a time (the upload path of the 16-bit plane is reused; a GRAY8 gradient is the rare case)
    const size_t px = (size_t)lib->W * lib->H;
    const int64_t batch = std::max<int64_t>(1, (int64_t)(((size_t)64 << 20) / (2 * px)));
    try {
        std::vector<uint16_t> wide;
        for (int64_t done = 0; done < count; done += batch) {
            const int64_t n = std::min(batch, count - done);
            wide.resize((size_t)n * px);
            const uint8_t *src = grad8 + (size_t)done * px;
            for (size_t i = 0; i < (size_t)n * px; i++) wide[i] = src[i];
            rc = cds_library_upload_variants(lib, first + done, n, wide.data(), nullptr);
            if (rc) return rc;
        }
    } catch (const std::bad_alloc &) {
        return fail(CDS_ERR_OUT_OF_MEMORY, "out of host memory widening the gradient images");
    }
    return CDS_OK;
}

int cds_library_download_variants(cds_library *lib, int64_t first, int64_t count, uint16_t *grad, uint8_t *zgap_rgb) {
    int rc = check_range(lib, first, count);
    if (rc) return rc;
    if (count == 0) return CDS_OK;
    if (!lib->d_grad || !lib->d_zgap) return fail(CDS_ERR_INVALID_ARGUMENT, "library holds no variants");
    CDS_HIP(hipSetDevice(lib->ctx->device));
    if (grad) {
        rc = rows_d2h(lib, (uint8_t *)grad, (const uint8_t *)lib->d_grad + (size_t)first * lib->grad_stride, lib->gpitch,
                      (size_t)2 * lib->W, (size_t)lib->H * count);
        if (rc) return rc;
    }
    if (zgap_rgb) {
        rc = rows_d2h(lib, zgap_rgb, lib->d_zgap + (size_t)first * lib->img_stride, lib->pitch, (size_t)3 * lib->W,
                      (size_t)lib->H * count);
        if (rc) return rc;
    }
    CDS_HIP(hipStreamSynchronize(lib->ctx->stream));
    return CDS_OK;
}

// ---- pixel-match mask set -----------------------------------------------------------------------
static int ensure_dense_layout(const cds_maskset *cms);   // (the dense kernel's records, laid out by the first dense search)

int cds_maskset_create(cds_context *ctx, int32_t n_masks, const uint8_t *masks_rgb, int32_t W, int32_t H,
                       const cds_pixmatch_params *params, const cds_rect *excluded, int32_t n_excluded,
                       cds_maskset **out) {
    if (!ctx || !out || !params) return fail(CDS_ERR_INVALID_ARGUMENT, "NULL argument");
    *out = nullptr;
    if (n_masks <= 0 || !masks_rgb) return fail(CDS_ERR_INVALID_ARGUMENT, "no masks");
    if (!image_size_ok(W, H)) return fail(CDS_ERR_UNSUPPORTED, kImageSizeMessage);
    if (params->xy_shift & 1) return fail(CDS_ERR_INVALID_ARGUMENT, "XY shift parameter must be an even number.");
    if (params->xy_shift < 0 || params->xy_shift > 2)
        return fail(CDS_ERR_UNSUPPORTED, "xy_shift must be 0 or 2 (the reference overruns its shift table for >= 4)");
    if (n_excluded < 0 || n_excluded > CDS_MAX_RECTS || (n_excluded > 0 && !excluded))
        return fail(CDS_ERR_INVALID_ARGUMENT, "bad excluded-region list");
    const int shift = params->xy_shift;
    const bool mirror = params->mirror != 0;
    const size_t img_bytes = (size_t)3 * W * H;

    PrepTimer pt;
    std::vector<std::vector<MaskPx>> per_mask((size_t)n_masks);
    std::vector<int32_t> sizes((size_t)n_masks, 0);
    int m0 = H, m1 = 0, xlo = W, xhi = 0;  // over all device pixels of all masks
    // the masks are independent: host threads take them in turn (each with its own interval cache)
    {
        const unsigned hc = std::thread::hardware_concurrency();
        const int n_threads = std::max(1, std::min<int>(n_masks, (int)std::min<unsigned>(hc ? hc : 1u, 16u)));
        std::atomic<int> next{0};
        std::mutex merge;
        auto worker = [&]() {
            std::unordered_map<uint64_t, Intervals> cache;
            int lm0 = H, lm1 = 0, lxlo = W, lxhi = 0;
            for (int m = next.fetch_add(1); m < n_masks; m = next.fetch_add(1)) {
                const uint8_t *img = masks_rgb + (size_t)m * img_bytes;
                std::vector<MaskPx> &list = per_mask[(size_t)m];
                int32_t size = 0;
                auto one_pixel = [&](int x, int y) {
                    const uint8_t *p = img + 3 * ((size_t)y * W + x);
                    int r = p[0], g = p[1], b = p[2];
                    if (!(r > params->mask_threshold || g > params->mask_threshold || b > params->mask_threshold)) return;
                    if (in_rects(excluded, n_excluded, x, y)) return;
                    size++;  // getQuerySize counts every position, class or not
                    double a;
                    int c1 = hue_class(r, g, b, &a);
                    if (c1 == C_NONE) return;  // pxGap is 10000 against everything
                    uint64_t key;
                    memcpy(&key, &a, 8);
                    key ^= (uint64_t)c1 << 61;   // a < 1 so the top bits of its encoding are free
                    auto it = cache.find(key);
                    Intervals iv;
                    if (it == cache.end()) {
                        iv = intervals_for(ctx->ratios, c1, a, params->z_tolerance);
                        cache.emplace(key, iv);
                    } else {
                        iv = it->second;
                    }
                    if (iv.len1 == 0 && iv.lo1 == 1 && iv.len2 == 0 && iv.lo2 == 1) return;  // can never match
                    list.push_back({x, y, iv.lo1, iv.len1, iv.lo2, iv.len2});
                    lm0 = std::min(lm0, y); lm1 = std::max(lm1, y + 1);
                    lxlo = std::min(lxlo, x); lxhi = std::max(lxhi, x + 1);
                };
                // a mask is a few percent signal on black and a black pixel is above no threshold (>= 0): whole runs of 8 black pixels
                // (24 zero bytes) are skipped with three 64-bit compares -- the scan of the frame was most of a single mask's preparation
                const bool skip_black = params->mask_threshold >= 0;
                for (int y = 0; y < H; y++) {
                    const uint8_t *row = img + (size_t)3 * y * W;
                    int x = 0;
                    while (x < W) {
                        if (skip_black && x + 8 <= W) {
                            uint64_t w0, w1, w2;
                            memcpy(&w0, row + 3 * x, 8);
                            memcpy(&w1, row + 3 * x + 8, 8);
                            memcpy(&w2, row + 3 * x + 16, 8);
                            if ((w0 | w1 | w2) == 0) {
                                x += 8;
                                continue;
                            }
                            for (int k = 0; k < 8; k++) one_pixel(x + k, y);
                            x += 8;
                        } else {
                            one_pixel(x, y);
                            x++;
                        }
                    }
                }
                sizes[(size_t)m] = size;
            }
            std::lock_guard<std::mutex> lk(merge);
            m0 = std::min(m0, lm0); m1 = std::max(m1, lm1); xlo = std::min(xlo, lxlo); xhi = std::max(xhi, lxhi);
        };
        std::vector<std::thread> pool;
        for (int i = 1; i < n_threads; i++) pool.emplace_back(worker);
        worker();
        for (auto &th : pool) th.join();
    }
    pt.lap("pixel lists + intervals");
    cds_maskset *ms = new cds_maskset();
    ms->ctx = ctx; ms->params = *params; ms->W = W; ms->H = H; ms->n_masks = n_masks; ms->sizes = sizes;
    size_t total = 0;
    for (auto &l : per_mask) total += l.size();
    ms->n_px = total;
    {   // the device pixels laid out for the sparse scan (cds_sparse.hip)
        int rc = build_sparse_plans(ms, per_mask);
        if (rc) {
            delete ms;
            return rc;
        }
    }
    pt.lap("sparse plans (all)");
    if (total == 0) {  // nothing can ever match: the search returns zeros without a launch
        ms->m0 = ms->m1 = 0;
        ms->dense_ready = true;
        *out = ms;
        return CDS_OK;
    }
    ms->m0 = m0; ms->m1 = m1; ms->px_xlo = xlo; ms->px_xhi = xhi;
    // column window of the dense scan: plain probes touch [xlo-shift, xhi+shift), mirrored ones [W-xhi-shift, W-xlo+shift)
    {
        int clo = xlo - shift, chi = xhi + shift;
        if (mirror) {
            clo = std::min(clo, W - xhi - shift);
            chi = std::max(chi, W - xlo + shift);
        }
        if (getenv("CDS_SCAN_FULLWINDOW")) {  // experiment: stage whole rows instead of the column window
            clo = std::min(clo, 0);
            chi = std::max(chi, W);
        }
        ms->cx0 = floor4(clo) & ~15;                       // floor to a multiple of 16 pixels (also for negatives)
        ms->wc = std::max(32, ((ceil4(chi) - ms->cx0) + 15) & ~15);   // >= 2 items per row (32-bit magic division)
        if (ms->wc > 2047) {
            cds_maskset_destroy(ms);
            return fail(CDS_ERR_UNSUPPORTED, "column window too wide");
        }
    }
    ms->lazy_px = std::move(per_mask);   // (the dense kernel's records are laid out by the first dense search)
    if (getenv("CDS_MASKSET_EAGER_DENSE")) {
        const int rc = ensure_dense_layout(ms);
        if (rc) {
            cds_maskset_destroy(ms);
            return rc;
        }
    }
    *out = ms;
    return CDS_OK;
}

// The dense kernel's record layout of a mask set (see cds_scan.hip), built once, by the first search that needs it.
static int build_dense_layout(cds_maskset *ms) {
    PrepTimer pt;
    cds_context *ctx = ms->ctx;
    std::vector<std::vector<MaskPx>> &per_mask = ms->lazy_px;
    const int n_masks = ms->n_masks, W = ms->W, H = ms->H;
    const int shift = ms->params.xy_shift;
    const bool mirror = ms->params.mirror != 0;
    const int m0 = ms->m0, m1 = ms->m1;
    size_t total = ms->n_px;
    typedef MaskPx Px;
    const int nrows = m1 - m0;
    const int nbands = (nrows + kScanBandRows - 1) / kScanBandRows;
    ms->nbands = nbands;
    // Pixels are regrouped per band of kScanBandRows mask rows, sorted by (x, y) inside a band so that 64
    // consecutive pixels form a spatially compact chunk; each chunk carries the staging blocks / rows its probes
    // can touch, which lets a wave skip an orientation whose neighbourhood is all black (cds_scan.hip).
    // Masks are scanned in launches ("groups") of ms->group masks sharing one pass over the library; the chunks of a
    // group are stored band-major (band, mask, chunk) so that one band's work of the whole group is one contiguous
    // chunk range: bandptr[group][band] .. bandptr[group][band + 1].
    int group = 128;
    if (const char *e = getenv("CDS_SCAN_MASKS_PER_LAUNCH")) group = std::max(1, std::min(256, atoi(e)));
    if (n_masks == 1) group = 1;
    group = std::min(group, n_masks);
    // the batch kernel keeps 4 more ring rows and the masks' counters in LDS: a window too wide for that is scanned
    // mask by mask with the single-mask kernel (same results)
    if (group > 1 && scan_lds_bytes(ms->wc, group, true, shift == 2, mirror) > ctx->lds_per_block) group = 1;
    ms->group = group;
    // Record layout: the batch kernel's (band-major chunks, LDS offsets) for groups of masks -- and for a single mask
    // with so many pixels that its probes outweigh the staging (the batch kernel deals any number of chunks per band to
    // 16 waves; the single-mask kernel prefetches 4 chunks per wave and band and crawls through the rest).
    bool batch = group > 1;
    if (!batch) {
        const char *fb = getenv("CDS_SCAN_SINGLE_AS_BATCH");  // "1" always, "0" never, unset: by size
        const size_t px_per_band = total / (size_t)std::max(1, (m1 - m0 + kScanBandRows - 1) / kScanBandRows);
        const bool big = px_per_band > (size_t)kBatchForSinglePxPerBand;
        if ((fb ? atoi(fb) != 0 : big) && scan_lds_bytes(ms->wc, 1, true, shift == 2, mirror) <= ctx->lds_per_block) batch = true;
    }
    if (!batch && nrows > 1023) {
        // the single-mask records keep a pixel's row in 10 bits (relative to the mask's first row): a taller mask -- a descending neuron in a
        // 573 x 1119 VNC MIP -- takes the batch layout, whose records address the LDS ring instead
        if (scan_lds_bytes(ms->wc, 1, true, shift == 2, mirror) > ctx->lds_per_block)
            return fail(CDS_ERR_UNSUPPORTED, "a mask spanning more than 1,023 rows whose window does not fit the batch kernel's LDS cannot be scanned over a dense "
                                             "library: use a sparse library");
        batch = true;
    }
    ms->batch = batch;
    const int ngroups = (n_masks + group - 1) / group;
    std::vector<uint32_t> bandptr((size_t)ngroups * (nbands + 1));
    std::vector<uint4> chunks;
    std::vector<uint32_t> pos;
    std::vector<uint4> ivs;
    pos.reserve(total + total / 4 + 64);
    ivs.reserve(total + total / 4 + 64);
    auto desc = [&](int cmin, int cmax) -> uint32_t {   // columns are window-relative pixels
        const int g0 = (cmin - shift) >> 2, g1 = (cmax + shift) >> 2;
        if ((g1 >> 5) - (g0 >> 5) > 1) return 0;   // spans more than two occupancy words: never skip
        return (uint32_t)g0 | ((uint32_t)g1 << 9) | (1u << 31);
    };
    std::vector<std::vector<std::vector<Px>>> bands((size_t)n_masks);
    for (int m = 0; m < n_masks; m++) {
        bands[(size_t)m].resize((size_t)nbands);
        for (const Px &p : per_mask[(size_t)m]) bands[(size_t)m][(size_t)((p.y - m0) / kScanBandRows)].push_back(p);
    }
    // (the pixel lists stay with the mask set until the upload below has succeeded: a failed build -- out of device memory with
    // several host threads in the per-pair path -- must leave the set as it was, so that the next dense search can try again)
    for (int q = 0; q < ngroups; q++) {
        for (int b = 0; b < nbands; b++) {
            bandptr[(size_t)q * (nbands + 1) + b] = (uint32_t)chunks.size();
            for (int m = q * group; m < std::min(n_masks, (q + 1) * group); m++) {
                std::vector<Px> &bl = bands[(size_t)m][(size_t)b];
                std::sort(bl.begin(), bl.end(), [](const Px &a, const Px &c) { return a.x < c.x || (a.x == c.x && a.y < c.y); });
                for (size_t s0 = 0; s0 < bl.size(); s0 += 64) {
                    const size_t s1 = std::min(bl.size(), s0 + 64);
                    int xmin = W, xmax = -1, ymin = H, ymax = -1;
                    bool two = false;
                    for (size_t i = s0; i < s1; i++) {
                        const Px &p = bl[i];
                        xmin = std::min(xmin, p.x); xmax = std::max(xmax, p.x);
                        ymin = std::min(ymin, p.y); ymax = std::max(ymax, p.y);
                        // the mirrored column lies inside the window only when the window was made symmetric (mirror on);
                        // without mirroring it is unused and must not spill into the neighbouring bit fields
                        const uint32_t cpl = (uint32_t)(p.x - ms->cx0), cmi = mirror ? (uint32_t)(W - 1 - p.x - ms->cx0) : cpl;
                        if (!batch) {
                            pos.push_back(cpl | (cmi << 11) | ((uint32_t)(p.y - m0) << 22));
                        } else {  // mask batches: dword offsets of the first probe inside the (non-wrapping) LDS ring
                            const uint32_t row0 = (uint32_t)((p.y - m0) & 15) * (uint32_t)ring_pitch(ms->wc);
                            pos.push_back((row0 + cpl - shift) | ((row0 + cmi - shift) << 16));
                        }
                        ivs.push_back(make_uint4(p.lo1, p.len1, p.lo2, p.len2));
                        two = two || !(p.lo2 == kEmptyIv.lo2 && p.len2 == kEmptyIv.len2);
                    }
                    for (size_t i = s1; i < s0 + 64; i++) {  // pad the chunk: a pixel that can never match, at a valid position
                        pos.push_back(pos[pos.size() - (i - s0)]);
                        ivs.push_back(make_uint4(kEmptyIv.lo1, kEmptyIv.len1, kEmptyIv.lo2, kEmptyIv.len2));
                    }
                    const int r0 = ymin - m0, nr = (ymax - ymin) + 1 + 2 * shift;  // ring rows r0 .. r0+nr-1 (relative to m0-halo)
                    uint4 ch;
                    ch.x = (uint32_t)(s1 - s0) | ((uint32_t)(m - q * group) << 8) | (two ? 1u << 16 : 0u);
                    ch.y = (uint32_t)r0 | ((uint32_t)nr << 10);
                    ch.z = desc(xmin - ms->cx0, xmax - ms->cx0);
                    ch.w = mirror ? desc(W - 1 - xmax - ms->cx0, W - 1 - xmin - ms->cx0) : 0;
                    chunks.push_back(ch);
                }
            }
        }
        bandptr[(size_t)q * (nbands + 1) + nbands] = (uint32_t)chunks.size();
    }
    // slack behind the arrays: the single-mask kernel prefetches the records of kScanPrefetchChunks chunks at a fixed stride
    // without clamping (a wave with fewer chunks in the band reads -- and ignores -- what follows)
    for (int i = 0; i < kScanPrefetchSlack; i++) {
        chunks.push_back(make_uint4(0, 0, 0, 0));
        for (int l = 0; l < 64; l++) {
            pos.push_back(0);
            ivs.push_back(make_uint4(kEmptyIv.lo1, kEmptyIv.len1, kEmptyIv.lo2, kEmptyIv.len2));
        }
    }
    total = pos.size();
    pt.lap("dense layout (host)");
    hipSetDevice(ctx->device);
    hipError_t e = dev_malloc(&ms->d_bandptr, bandptr.size() * 4);
    if (e == hipSuccess) e = dev_malloc(&ms->d_chunks, chunks.size() * sizeof(uint4));
    if (e == hipSuccess) e = dev_malloc(&ms->d_px_pos, total * 4);
    if (e == hipSuccess) e = dev_malloc(&ms->d_px_iv, total * sizeof(uint4));
    if (e == hipSuccess) e = hipMemcpy(ms->d_bandptr, bandptr.data(), bandptr.size() * 4, hipMemcpyHostToDevice);
    ms->h_bandptr = bandptr;
    if (e == hipSuccess) e = hipMemcpy(ms->d_chunks, chunks.data(), chunks.size() * sizeof(uint4), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(ms->d_px_pos, pos.data(), total * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(ms->d_px_iv, ivs.data(), total * sizeof(uint4), hipMemcpyHostToDevice);
    if (e != hipSuccess) {   // nothing half-built stays behind: the next call starts from the pixel lists again
        if (ms->d_bandptr) hipFree(ms->d_bandptr);
        if (ms->d_chunks) hipFree(ms->d_chunks);
        if (ms->d_px_pos) hipFree(ms->d_px_pos);
        if (ms->d_px_iv) hipFree(ms->d_px_iv);
        ms->d_bandptr = nullptr; ms->d_chunks = nullptr; ms->d_px_pos = nullptr; ms->d_px_iv = nullptr;
        ms->h_bandptr.clear();
        return hip_fail(e, "maskset upload");
    }
    std::vector<std::vector<MaskPx>>().swap(ms->lazy_px);   // the layout is on the device: the lists have served their purpose
    pt.lap("dense layout (upload)");
    return CDS_OK;
}

static int ensure_dense_layout(const cds_maskset *cms) {
    if (cms->dense_ready.load(std::memory_order_acquire)) return CDS_OK;
    cds_maskset *ms = const_cast<cds_maskset *>(cms);
    std::lock_guard<std::mutex> lk(ms->dense_mu);
    if (ms->dense_ready.load(std::memory_order_relaxed)) return CDS_OK;
    const int rc = build_dense_layout(ms);
    if (rc) return rc;   // (a failed build frees what it allocated and keeps the pixel lists: the next dense search builds again)
    ms->dense_ready.store(true, std::memory_order_release);
    return CDS_OK;
}

void cds_maskset_destroy(cds_maskset *ms) {
    if (!ms) return;
    hipSetDevice(ms->ctx->device);
    if (ms->d_bandptr) hipFree(ms->d_bandptr);
    if (ms->d_chunks) hipFree(ms->d_chunks);
    if (ms->d_px_pos) hipFree(ms->d_px_pos);
    if (ms->d_px_iv) hipFree(ms->d_px_iv);
    free_sparse_plans(ms);
    ms->pair_pool.slots.clear();
    delete ms;
}

int32_t cds_maskset_size(const cds_maskset *ms) { return ms ? ms->n_masks : 0; }

int cds_maskset_query_sizes(const cds_maskset *ms, int32_t *sizes) {
    if (!ms || !sizes) return fail(CDS_ERR_INVALID_ARGUMENT, "NULL argument");
    memcpy(sizes, ms->sizes.data(), ms->sizes.size() * sizeof(int32_t));
    return CDS_OK;
}

// `on_stream` (the pair-slot path): launch on that stream instead of the context's; the library then belongs to a slot
// context (same device) and the context check is skipped
static int search_impl(const cds_maskset *ms, const cds_library *lib, int64_t first, int64_t count, int32_t *d_scores,
                       int32_t *d_counts, bool timed_sync, hipStream_t on_stream = nullptr) {
    if (!ms || !lib) return fail(CDS_ERR_INVALID_ARGUMENT, "NULL argument");
    int rc = check_range(lib, first, count);
    if (rc) return rc;
    if (lib->W != ms->W || lib->H != ms->H) {
        char buf[200];
        snprintf(buf, sizeof buf, "Invalid image size - target's image size (%d, %d) must match query's image size: (%d, %d)",
                 ms->W, ms->H, lib->W, lib->H);  // PixelMatchColorDepthSearchAlgorithm.java:171-175
        return fail(CDS_ERR_INVALID_ARGUMENT, buf);
    }
    if (!on_stream && lib->ctx != ms->ctx) return fail(CDS_ERR_INVALID_ARGUMENT, "mask set and library belong to different contexts");
    if (count == 0) return CDS_OK;
    cds_context *ctx = ms->ctx;
    CDS_HIP(hipSetDevice(ctx->device));
    rc = ensure_dense_layout(ms);
    if (rc) return rc;
    hipStream_t st = on_stream ? on_stream : ctx->stream;
    if (ms->n_px == 0) {  // no mask pixel can match: all-zero result without a launch (the kernel writes every output otherwise)
        CDS_HIP(hipMemsetAsync(d_scores, 0, (size_t)ms->n_masks * count * sizeof(int32_t), st));
        if (d_counts) CDS_HIP(hipMemsetAsync(d_counts, 0, (size_t)ms->n_masks * count * CDS_N_TRANSFORMS * sizeof(int32_t), st));
    }
    double ms_sum = 0;
    int launches = 0;
    if (ms->n_px > 0) {
        const int chunk = ms->group;  // masks per launch (fixed when the mask set was laid out)
        hipEvent_t ev0 = nullptr, ev1 = nullptr;
        int32_t *scratch = nullptr;
        // every error path below leaves through this: the events and the stream-ordered scratch buffer are released
        auto cleanup = [&]() {
            if (ev0) hipEventDestroy(ev0);
            if (ev1) hipEventDestroy(ev1);
            if (scratch) hipFreeAsync(scratch, st);
            ev0 = ev1 = nullptr;
            scratch = nullptr;
        };
#define SI_HIP(call)                                   \
    do {                                               \
        hipError_t _e = (call);                        \
        if (_e != hipSuccess) {                        \
            cleanup();                                 \
            return cds::hip_fail(_e, #call);           \
        }                                              \
    } while (0)
        if (timed_sync) {
            SI_HIP(hipEventCreate(&ev0));
            SI_HIP(hipEventCreate(&ev1));
        }
        for (int g0 = 0; g0 < ms->n_masks; g0 += chunk) {
            ScanParams p{};
            p.lib = lib->d_rgb + (size_t)first * lib->img_stride;
            p.img_stride = lib->img_stride;
            p.pitch = lib->pitch; p.W = lib->W; p.H = lib->H; p.wpad = lib->pitch / 3;
            p.count = count;
            p.n_masks = std::min(chunk, ms->n_masks - g0);
            p.batch = ms->batch;
            p.nrows = ms->m1 - ms->m0;
            p.m0 = ms->m0;
            p.halo = ms->params.xy_shift;
            p.cx0 = ms->cx0; p.wc = ms->wc; p.gpr = ms->wc / 4;
            p.rp = ring_pitch(ms->wc);
            p.nseg = (p.gpr + 63) / 64;
            p.occw = 2 * p.nseg;
            p.nbands = ms->nbands;
            p.data_thr = (float)ms->params.data_threshold;
            p.bandptr = ms->d_bandptr + (size_t)(g0 / chunk) * (ms->nbands + 1);
            p.chunks = ms->d_chunks;
            p.px_pos = ms->d_px_pos;
            p.px_iv = ms->d_px_iv;
            p.out = d_scores + (size_t)g0 * count;
            p.counts = d_counts ? d_counts + (size_t)g0 * count * CDS_N_TRANSFORMS : nullptr;
            p.out_stride = count;
            p.out_offset = 0;
            if (const char *e = getenv("CDS_SCAN_DEBUG")) p.debug = atoi(e);
            // Band slices (mask batches): every probe of a (mask, band) re-reads that band's pixel records; a group's records
            // (~20 B per mask pixel: 38 MB for 128 benchmark masks) exceed an XCD's 4 MB L2, so one launch over all bands
            // streams them from the Infinity Cache for every target (239 GB per 16,384 targets).  Launching slice by slice
            // (all workgroups on the same few bands at the same time) keeps more of a slice's records in L2 (121 GB with
            // 3-band slices); the per-transform counts travel from slice to slice in a scratch buffer (or the caller's
            // counts array).  Measured: +6 % with 3..5 bands per slice; single-band slices lose it again to staging.
            const std::vector<uint32_t> &bp = ms->h_bandptr;
            const size_t gb = (size_t)(g0 / chunk) * (ms->nbands + 1);
            const size_t group_bytes = (size_t)(bp[gb + ms->nbands] - bp[gb]) * 64 * 20;
            int slice = ms->nbands;
            if (ms->batch && ms->nbands > 1) {
                const size_t budget = (size_t)10 << 20;  // ~10 MB of records per slice (4 bands of 128 benchmark masks)
                const size_t per_band = std::max<size_t>(1, group_bytes / ms->nbands);
                if (group_bytes > budget) slice = (int)std::max<size_t>(1, budget / per_band);
                if (const char *e = getenv("CDS_SCAN_BAND_SLICE")) slice = atoi(e) > 0 ? std::min(atoi(e), ms->nbands) : ms->nbands;
            }
            if (slice < ms->nbands && !p.counts) {
                SI_HIP(hipMallocAsync((void **)&scratch, (size_t)p.n_masks * count * CDS_N_TRANSFORMS * sizeof(int32_t), st));
                p.counts = scratch;
            }
            if (timed_sync) SI_HIP(hipEventRecord(ev0, st));
            for (int b0 = 0; b0 < ms->nbands; b0 += slice) {
                p.band_begin = b0;
                p.band_end = std::min(ms->nbands, b0 + slice);
                rc = launch_scan(ctx, p, ms->params.xy_shift == 2, ms->params.mirror != 0, st);
                if (rc) {
                    cleanup();
                    return rc;
                }
            }
            if (scratch) {
                int32_t *s = scratch;
                scratch = nullptr;
                SI_HIP(hipFreeAsync(s, st));
            }
            if (timed_sync) {
                SI_HIP(hipEventRecord(ev1, st));
                SI_HIP(hipEventSynchronize(ev1));
                float t = 0;
                SI_HIP(hipEventElapsedTime(&t, ev0, ev1));
                ms_sum += t;
            }
            launches++;
        }
        cleanup();
#undef SI_HIP
    }
    if (timed_sync) {
        std::lock_guard<std::mutex> lk(ms->mu);
        ms->last_ms = launches ? ms_sum / launches : 0;
        ms->last_launches = launches;
    }
    return CDS_OK;
}

int cds_search_device(const cds_maskset *ms, const cds_library *lib, int64_t first, int64_t count, int32_t *d_scores) {
    if (!d_scores && count > 0) return fail(CDS_ERR_INVALID_ARGUMENT, "d_scores is NULL");
    const bool timed = getenv("CDS_TIME_KERNELS") != nullptr;
    return search_impl(ms, lib, first, count, d_scores, nullptr, timed);
}

static int search_host(const cds_maskset *ms, const cds_library *lib, int64_t first, int64_t count, int32_t *scores,
                       uint8_t *mirrored, int32_t *counts) {
    if (!ms || !lib) return fail(CDS_ERR_INVALID_ARGUMENT, "NULL argument");
    int rc = check_range(lib, first, count);
    if (rc) return rc;
    if (count == 0) return CDS_OK;
    CDS_HIP(hipSetDevice(ms->ctx->device));
    const size_t n = (size_t)ms->n_masks * count;
    int32_t *d_scores = nullptr, *d_counts = nullptr;
    CDS_HIP(dev_malloc(&d_scores, n * sizeof(int32_t)));
    if (counts) {
        hipError_t e = dev_malloc(&d_counts, n * CDS_N_TRANSFORMS * sizeof(int32_t));
        if (e != hipSuccess) {
            hipFree(d_scores);
            return hip_fail(e, "hipMalloc(counts)");
        }
    }
    rc = search_impl(ms, lib, first, count, d_scores, d_counts, true);
    if (rc == CDS_OK) {
        std::vector<int32_t> packed(n);
        hipError_t e = hipMemcpyAsync(packed.data(), d_scores, n * sizeof(int32_t), hipMemcpyDeviceToHost, ms->ctx->stream);
        if (e == hipSuccess && counts)
            e = hipMemcpyAsync(counts, d_counts, n * CDS_N_TRANSFORMS * sizeof(int32_t), hipMemcpyDeviceToHost, ms->ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ms->ctx->stream);
        if (e != hipSuccess) rc = hip_fail(e, "copy scores");
        else
            for (size_t i = 0; i < n; i++) {
                if (scores) scores[i] = packed[i] & 0x7fffffff;
                if (mirrored) mirrored[i] = (uint8_t)((uint32_t)packed[i] >> 31);
            }
    }
    hipFree(d_scores);
    if (d_counts) hipFree(d_counts);
    return rc;
}

int cds_search(const cds_maskset *ms, const cds_library *lib, int64_t first, int64_t count, int32_t *scores,
               uint8_t *mirrored) {
    return search_host(ms, lib, first, count, scores, mirrored, nullptr);
}

int cds_search_counts(const cds_maskset *ms, const cds_library *lib, int64_t first, int64_t count, int32_t *counts) {
    if (!counts && count > 0) return fail(CDS_ERR_INVALID_ARGUMENT, "counts is NULL");
    return search_host(ms, lib, first, count, nullptr, nullptr, counts);
}

int cds_score_pair(const cds_maskset *ms, int32_t mask_index, const uint8_t *target_rgb, int32_t W, int32_t H,
                   int32_t *score, uint8_t *mirrored) {
    if (!ms || !target_rgb || !score || !mirrored) return fail(CDS_ERR_INVALID_ARGUMENT, "NULL argument");
    if (mask_index < 0 || mask_index >= ms->n_masks) return fail(CDS_ERR_INVALID_ARGUMENT, "mask_index out of range");
    if (ms->sizes[(size_t)mask_index] == 0) {  // PixelMatchColorDepthSearchAlgorithm.java:169-170
        *score = 0; *mirrored = 0;
        return CDS_OK;
    }
    if (W != ms->W || H != ms->H) {
        char buf[200];
        snprintf(buf, sizeof buf, "Invalid image size - target's image size (%d, %d) must match query's image size: (%d, %d)",
                 ms->W, ms->H, W, H);
        return fail(CDS_ERR_INVALID_ARGUMENT, buf);
    }
    // the per-pair call of the reference's interface: a pool of resident slots (each with a device score array and its own
    // stream) is kept with the mask set, so a call costs an upload, a one-workgroup launch and a 4-byte-per-mask read-back, and
    // calls from different host threads overlap
    PairSlot *slot = nullptr;
    int rc = ms->pair_pool.acquire(ms->ctx, W, H, (size_t)ms->n_masks, &slot);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(slot->mu, std::adopt_lock);
    rc = cds_library_upload(slot->lib, 0, 1, target_rgb);
    if (rc) return rc;
    hipStream_t st = slot->sub->stream;
    rc = search_impl(ms, slot->lib, 0, 1, slot->d_scores, nullptr, false, st);
    if (rc) return rc;
    int32_t packed = 0;
    CDS_HIP(hipMemcpyAsync(&packed, slot->d_scores + mask_index, sizeof packed, hipMemcpyDeviceToHost, st));
    CDS_HIP(hipStreamSynchronize(st));
    *score = packed & 0x7fffffff;
    *mirrored = (uint8_t)((uint32_t)packed >> 31);
    return CDS_OK;
}

float cds_pixel_match_normalized(int32_t score, int32_t query_size) {
    if (query_size == 0) return 0.0f;
    return (float)((double)score / (double)query_size);
}

int cds_is_match(int32_t score, int32_t query_size, double pct_positive_pixels) {
    double thr = pct_positive_pixels / 100;
    return score > 0 && (double)cds_pixel_match_normalized(score, query_size) > thr;
}

int cds_maskset_last_kernel_ms(const cds_maskset *ms, double *avg_ms, int32_t *launches) {
    if (!ms) return fail(CDS_ERR_INVALID_ARGUMENT, "NULL argument");
    std::lock_guard<std::mutex> lk(ms->mu);
    if (avg_ms) *avg_ms = ms->last_ms;
    if (launches) *launches = ms->last_launches;
    return CDS_OK;
}

// ---- diagnostics used by the CPU-side tests of the host logic (include/cds_hip_debug.h) ----------
int cds_debug_intervals(const float *rcp_table, int32_t r, int32_t g, int32_t b, double z_tolerance, uint32_t *out4) {
    if (!rcp_table || !out4) return fail(CDS_ERR_INVALID_ARGUMENT, "NULL argument");
    std::vector<RatioEntry> tab;
    static thread_local std::vector<RatioEntry> cached;
    static thread_local float cached_rcp[256];
    static thread_local bool have = false;
    if (!have || memcmp(cached_rcp, rcp_table, sizeof cached_rcp) != 0) {
        int rc = build_ratio_table(rcp_table, cached);
        if (rc) return rc;
        memcpy(cached_rcp, rcp_table, sizeof cached_rcp);
        have = true;
    }
    double a;
    int c1 = hue_class(r, g, b, &a);
    Intervals iv = intervals_for(cached, c1, a, z_tolerance);
    out4[0] = iv.lo1; out4[1] = iv.len1; out4[2] = iv.lo2; out4[3] = iv.len2;
    return CDS_OK;
}

uint32_t cds_debug_hue_code(const float *rcp_table, int32_t r, int32_t g, int32_t b, int32_t threshold) {
    return host_hue_code(rcp_table, r, g, b, threshold);
}

int cds_debug_device_rcp(const cds_context *ctx, float *out256) {
    if (!ctx || !out256) return fail(CDS_ERR_INVALID_ARGUMENT, "NULL argument");
    memcpy(out256, ctx->rcp, sizeof ctx->rcp);
    return CDS_OK;
}

}  // extern "C"
