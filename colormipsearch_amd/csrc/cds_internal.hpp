// cds_internal.hpp -- private structures of libcds_hip.so (MI355X / gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <memory>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/cds_hip.h"

namespace cds {

void set_error(const std::string &msg);
int fail(int code, const std::string &msg);
int hip_fail(hipError_t e, const char *what);
// fail() with a printf-style message
template <typename... A>
inline int failf(int code, const char *fmt, A... a) {
    char buf[768];
    snprintf(buf, sizeof buf, fmt, a...);
    return fail(code, std::string(buf));
}

#define CDS_HIP(call)                                      \
    do {                                                   \
        hipError_t _e = (call);                            \
        if (_e != hipSuccess) return cds::hip_fail(_e, #call); \
    } while (0)

// ---- scan kernel geometry shared by host preparation and device code (cds_scan.hip) ----
constexpr int kScanBandRows = 12;  // mask rows per band
constexpr int kScanPrefetchSlack = 16;  // chunks of slack behind the record arrays (4 waves x 4 prefetched chunks)
// a single mask with more pixels per band than this would use the batch kernel; measured (tools/big_mask.py): the
// single-mask kernel stays ahead up to at least 65 K mask pixels (3.2 vs 6.1 ms per 16 K targets), so: never by default
constexpr int kBatchForSinglePxPerBand = 1 << 30;

// ---- geometry limits (bit-packing of the mask pixel records, see cds_scan.hip) ----
// Width: 11-bit window columns, four 512-column strips of a sparse library.  Height: round 5 raised it from 1,000 so that the VNC alignment
// spaces' 573 x 1119 MIPs fit (the dense single-mask layout's 10-bit row field is the only thing that counted rows: a mask spanning more
// than 1,023 rows goes through the batch layout, cds_host.hip); the AREA stays where it was -- the scorer's per-lane sums and the
// multi-mask kernel's 20-bit plane offsets were sized for it.
constexpr int kMaxW = 2000;
constexpr int kMaxH = 1536;
constexpr int64_t kMaxPixels = 2000000;
inline bool image_size_ok(int W, int H) { return W >= 8 && H >= 1 && W <= kMaxW && H <= kMaxH && (int64_t)W * H <= kMaxPixels; }
constexpr const char *kImageSizeMessage = "image size outside 8..2000 x 1..1536 (at most 2,000,000 pixels)";

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }
// floor to a multiple of 4, also for negatives
inline int floor4(int v) { return v >= 0 ? v / 4 * 4 : -((-v + 3) / 4 * 4); }
inline int ceil4(int v) { return -floor4(-v); }

// Row pitch of a resident RGB image: multiple of 48 B so that rows are 16-B aligned AND a row is a
// whole number of 12-byte (4 pixel) load groups.
inline int rgb_pitch(int W) { return round_up(3 * W, 48); }
inline int grad_pitch(int W) { return round_up(2 * W, 16); }
// LDS ring row pitch (dwords) of the scan kernel for a column window of wc pixels (a multiple of 16): wc + 4 = 4 (mod 8),
// so the 12 rows of a band start 4 banks apart (a chunk of 64 x-sorted mask pixels is ~5 columns x 12 rows).
inline int ring_pitch(int wc) { return wc + 4; }

// One distinct value of second/max (a fraction n/d, 0 <= n < d <= 255) as the reference computes it in
// double, with the range of device codes its (n,d) representatives map to.
struct RatioEntry {
    double value;
    uint32_t code_lo, code_hi;  // (float_bits(fl32(n) * rcp[d]) >> 1), min/max over equal fractions
};

}  // namespace cds

struct cds_context {
    // sizeof(cds_context) as the translation unit that CREATED the context saw it.  The library's objects share this header; an object
    // compiled against another revision of it (a stale .o of an A/B build: round 4's `s16u` run, profiles/README.md) would read the
    // members at other offsets -- init_rcp_table (cds_scan.hip) checks the stamp before it touches the context.
    size_t layout_stamp = 0;
    int device = -1;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipStream_t upload_stream = nullptr;   // mask-set uploads (own non-blocking stream: a mask set is prepared while earlier ones are being
                                           // scanned on `stream`; its copies must neither queue behind those scans nor go through the
                                           // legacy default stream's implicit synchronisation)
    int n_cus = 0;
    size_t lds_per_block = 0;
    float rcp[256];                        // v_rcp_f32 of 0..255 as THIS device computes it
    std::vector<cds::RatioEntry> ratios;   // sorted by value, distinct
    uint8_t *d_slice_lut = nullptr;        // [6][256][256] slice number - 1 (shape scorer)
    bool owns_lut = true;                  // false in a pair-slot context (clone_context_for_slot): the table is the parent's
    std::atomic<int> sparse_load_only{0};  // cds_debug_sparse_load_only: THIS context's single-mask sparse searches launch the loads-only twin
    std::mutex mu;
    // Device blocks of mask sets, match lists and per-call temporaries are RECYCLED through the context (cds::cache_alloc / cache_free):
    // hipFree waits for every stream of the device, so a caller that pipelines mask sets -- one being scanned while the previous one's
    // results are fetched and its buffers released -- stalled behind the running scan after every set (measured: 60 ms of idle device
    // per two sets of 128 masks x 1 M targets).  A released block is kept (up to cache_cap bytes) and handed to the next request it
    // fits; the caller releases a block only when its work on the device is complete.
    std::atomic<long long> match_rows_hint{0};   // rows of the last device match list (cds_search_sparse_match_list sizes the next one by it)
    std::mutex cache_mu;
    std::vector<std::pair<void *, size_t>> cache_idle;       // (block, bytes)
    std::unordered_map<void *, size_t> cache_sizes;          // every block handed out by cache_alloc
    size_t cache_idle_bytes = 0, cache_cap = (size_t)6 << 30;
};

struct cds_library {
    cds_context *ctx = nullptr;
    int W = 0, H = 0;
    int pitch = 0;          // bytes per RGB row
    size_t img_stride = 0;  // bytes per RGB image = H * pitch
    int gpitch = 0;         // bytes per gradient row
    size_t grad_stride = 0;
    int64_t capacity = 0;
    uint8_t *d_rgb = nullptr;
    bool owns_rgb = false;
    uint16_t *d_grad = nullptr;  // lazily allocated
    uint8_t *d_zgap = nullptr;   // lazily allocated, same layout as d_rgb
    uint16_t *d_zslice = nullptr;  // derived from d_zgap, gradient layout: (slice number - 1) | max channel << 8; allocated and filled
                                   // LAZILY, per slot, by the first scoring call that prefers the converted plane (cds_shape.hip)
    std::vector<uint8_t> has_grad, has_zgap;  // per slot
    std::vector<uint8_t> zslice_ok;           // per slot: d_zslice holds the conversion of the slot's current zgap
    uint8_t *d_stage = nullptr;  // contiguous staging buffer of the host <-> pitched-rows transfers
    size_t stage_bytes = 0;
    std::mutex stage_mu;
    std::mutex mu;
};

namespace cds {

// Wait for the work enqueued on `st` UP TO NOW -- an event, not hipStreamSynchronize: when several host threads feed one stream (the
// processor's pipeline of mask sets), the stream-wide wait of one thread also waits for the scan another thread enqueued meanwhile.
inline hipError_t sync_here(hipStream_t st) {
    hipEvent_t ev = nullptr;
    hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    if (e != hipSuccess) return e;
    e = hipEventRecord(ev, st);
    if (e == hipSuccess) e = hipEventSynchronize(ev);
    hipEventDestroy(ev);
    return e;
}
// Every device allocation of the library that does NOT go through the block cache: hipMalloc, and -- when that reports out of memory
// while released blocks sit idle in the caches of this device's contexts (up to cache_cap bytes each) -- the caches are dropped and the
// allocation is tried once more (a library create could fail with gigabytes idle in the same context).
hipError_t dev_malloc_raw(void **p, size_t bytes);
template <typename T>
inline hipError_t dev_malloc(T **p, size_t bytes) { return dev_malloc_raw(reinterpret_cast<void **>(p), bytes); }
void register_context(cds_context *ctx);     // (the registry dev_malloc_raw walks)
void unregister_context(cds_context *ctx);
hipError_t cache_alloc(cds_context *ctx, void **p, size_t bytes);   // a recycled block of >= bytes (at most twice that), or a fresh hipMalloc
void cache_free(cds_context *ctx, void *p);                         // back to the context's idle list -- or hipFree when that is full / p is foreign
void cache_drop(cds_context *ctx);                                  // hipFree of every idle block (context destruction)

// Device-side view of a prepared pixel-match mask set.
struct ScanParams {
    const uint8_t *lib;  // first image of the scanned range
    size_t img_stride;
    int pitch, W, H, wpad;
    int64_t count;
    int n_masks;          // masks in this launch
    int batch;            // records are in the mask-batch layout (mask set laid out for > 1 mask per launch)
    int nrows;            // m1 - m0
    int m0;               // first mask row of the union span
    int halo;             // 0 or 2
    int cx0, wc;          // column window [cx0, cx0+wc), both multiples of 16 (cx0 may be negative)
    int gpr;              // wc / 4 : 4-pixel groups per staged row
    int rp;               // LDS ring row pitch in dwords = ring_pitch(wc)
    int nseg;             // ceil(gpr / 64): 64-group segments per staged row
    int occw;             // occupancy words (u32, one bit per group) per ring row = 2 * nseg
    int nbands;           // ceil(nrows / kScanBandRows)
    int band_begin, band_end;  // bands of this launch (a band slice; [0, nbands) = everything); slices after the first add to
                               // the counts in `counts`, the last one writes `out`
    float data_thr;
    const uint32_t *bandptr;  // [nbands+1] of this launch's mask group -> chunk index (chunks are stored band, mask, chunk)
    const uint4 *chunks;      // x: pixels | mask-in-group << 8 | has-second-interval << 16, y: r0 | nr<<10,
                              // z/w: plain/mirror g0 | g1<<9 | testable<<31 (64 pixels per chunk)
    const uint32_t *px_pos;   // one mask: cplain | cmir << 11 | yrel << 22; mask batches: LDS dword offsets plain | mirrored << 16
                              // (64 records per chunk, padded)
    const uint4 *px_iv;       // lo1, len1, lo2, len2
    int32_t *out;            // [n_masks][count]  score | mirrored << 31
    int32_t *counts;         // optional [n_masks][count][18]
    int64_t out_stride;      // elements between masks in out (== total count of the call)
    int64_t out_offset;      // offset of this launch's first target in out
    int debug;               // timing-only ablations (CDS_SCAN_DEBUG): 1 = skip probes, 2 = skip conversion
};

int launch_scan(cds_context *ctx, const ScanParams &p, bool xy2, bool mirror, hipStream_t stream);
size_t scan_lds_bytes(int wc, int n_masks, bool batch, bool xy2, bool mirror);  // dynamic LDS of one workgroup
void zgap_written(cds_library *lib, int64_t first, int64_t count);  // cds_shape.hip: slots got a new zgap image (their zslice entries are stale)
int init_rcp_table(cds_context *ctx);
struct MaskPx {
    int x, y;
    uint32_t lo1, len1, lo2, len2;
};
// cds_sparse.hip: build the sparse-scan plans of a mask set from its device pixels (called by cds_maskset_create)
// The per-pair entry points of the reference's interface (cds_score_pair, cds_shape_score_pair) are called from many
// host threads on ONE algorithm object (tools cmd/cdsprocess/LocalColorMIPSearchProcessor.java:88-105).  Each call needs a
// resident slot for its target; a mask set keeps a small POOL of them, each with its own stream (a light copy of the
// context: same device and tables, own stream), so that concurrent calls overlap their uploads, launches and read-backs
// instead of queueing on one lock.
struct PairSlot {
    std::mutex mu;
    cds_context *sub = nullptr;      // own stream
    cds_library *lib = nullptr;      // one dense slot (+ variants for the shape scorer)
    int32_t *d_scores = nullptr;     // [n_masks] (pixel-match only)
    ~PairSlot();
};
struct PairSlotPool {
    std::mutex grow_mu;
    std::vector<std::unique_ptr<PairSlot>> slots;
    std::atomic<unsigned> next{0};
    // a locked slot (created on demand, at most `max_slots`); *out stays locked until the caller unlocks out->mu
    int acquire(cds_context *parent, int W, int H, size_t score_elems, PairSlot **out);
};
cds_context *clone_context_for_slot(cds_context *parent);
// phase timer of the mask-set preparation: CDS_TIME_PREP=1 prints one line per phase to stderr (diagnostic)
struct PrepTimer {
    bool on;
    std::chrono::steady_clock::time_point t;
    PrepTimer() : on(getenv("CDS_TIME_PREP") != nullptr), t(std::chrono::steady_clock::now()) {}
    void lap(const char *what) {
        if (!on) return;
        const auto n = std::chrono::steady_clock::now();
        fprintf(stderr, "[cds prep] %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(n - t).count());
        t = n;
    }
};
// cds_shape.hip: a GradientAreaGap scoring call in two halves (asynchronous begin, blocking end) for cds_gradscore_sharded
struct GradscoreJob;
int gradscore_begin(const cds_shapeset *ss, const cds_library *lib, int64_t n_pairs, const int32_t *pair_mask, const int64_t *pair_slot,
                    hipStream_t on_stream, GradscoreJob **out);
int gradscore_end(GradscoreJob *job, int64_t *gradient_area_gap, int64_t *high_expression_area, uint8_t *mirrored);  // NULL outputs: wait and free
cds_context *shapeset_context(const cds_shapeset *ss);
int shapeset_n_masks(const cds_shapeset *ss);
bool take_density_refusal();   // cds_sparse.hip: did THIS thread's last failing append fail on the density limit (and not on another CDS_ERR_UNSUPPORTED)?  Clears the flag.
int sparse_library_rollback(cds_sparse_library *sl, int64_t n_images, int64_t n_px);   // cds_sparse.hip (for cds_sparse_library_append_files)
int build_sparse_plans(cds_maskset *ms, const std::vector<std::vector<MaskPx>> &per_mask);
void free_sparse_plans(cds_maskset *ms);

}  // namespace cds

namespace cds {

// ---- sparse library (cds_sparse.hip) ------------------------------------------------------------------------------
// Record of one signal pixel (a pixel with a strict maximum channel -- every other pixel has no hue class and can never
// match, AbstractColorDepthSearchAlgorithm.java:195-225): the two channel values its hue ratio is made of, its HUE BAND
// (class * 5 + floor(5 * second / max) with the device's own arithmetic: 5..34; class = band / 5) -- what stage 1 of the scan
// tests against the mask's band map, computed once when the image is converted instead of once per scan -- and its column
// RELATIVE TO ITS STRIP (a run of records lies in one strip, whose first column is wave-uniform in the scan):
//   bit 31 first record of its image row | bits 30..22 x - strip start (< 512) | bits 21..16 band | bits 15..8 second channel | bits 7..0 max
// An image is stored strip by strip (up to 4 column strips of at most 512 pixels), a strip row by row, a row by ascending x.
// A (strip, row) cell without signal pixels holds one filler record (bit 31 only), so "row = rows started so far" holds everywhere.
constexpr uint32_t kRecRowFlag = 0x80000000u;
constexpr int kRecXShift = 22;
constexpr uint32_t kRecXMask = 511u;
constexpr int kRecBandShift = 16;
constexpr uint32_t kRecBandMask = 63u;
constexpr int kMaxStripW = 512;

// One mask prepared for the sparse scan: the mask's bounding box cut into slabs of `slab_rows` mask rows; per slab one
// ready-to-copy LDS image [plane u16 or u8 (slab_rows + 8) x pitch | band map u32 ceil((slab_rows + 4) / 4) x nwpr (one word per 4-row x 8-column cell)]:
//   plane[(my - slab_y0) + 4][(mx - xlo) + 4] = 1-based index of the mask pixel's interval record (0 = no mask pixel);
//   band-map word of cell ((ty - (slab_y0 - 2)) / 4, (tx - (xlo - 2)) / 8), bit b = some transform maps a target position of
//   the cell onto a mask pixel of the slab whose intervals contain a code of hue band b (cds_sparse.hip, band_bits_of).
struct SparseMaskPlan {
    int xlo = 0, cols = 0, ylo = 0, nrows = 0;  // bounding box of the mask's device pixels
    int pitch = 0;        // plane row pitch in u16 elements (multiple of 8, >= cols + 8)
    int nwpr = 0;         // band-map words per cell row (with the guard cell)
    int slab_rows = 0, n_slabs = 0;
    int near_off = 0;     // byte offset of the band map inside a slab image
    int slab_bytes = 0;   // multiple of 16
    int n_iv = 0;         // interval records (index 0 = the empty record) -- iv table has n_iv entries
    bool iv_in_lds = true;
    bool plane8 = false;  // one byte per plane cell (n_iv <= 256)
    size_t slabs_off = 0; // byte offset of this mask's slab images in cds_maskset::d_sp_slabs
    size_t iv_off = 0;    // element offset of its interval table in cds_maskset::d_sp_iv
    size_t lds_bytes = 0; // dynamic LDS of one workgroup
    // the two-orientation band map of sparse_scan16_kernel (8-byte cells: plain word, mirrored word; columns relative to ux0, the
    // left end of the union of the mask's window and its mirror image), one image per slab BEHIND the slab images of the mask
    int ux0 = 0;          // first column of the union window (target coordinates; may be negative)
    int g64 = 0;          // first column (relative to ux0) of the always-empty guard cell
    int rowb64 = 0;       // bytes per cell row (cells x 8)
    int map64_bytes = 0;  // bytes of one slab's map (multiple of 16); 0 = the mask cannot go through the 16-bit stream
};

struct SparseScanParams {
    const uint16_t *px16;     // the 16-bit stage-1 stream of the same records (sparse_scan16_kernel), or NULL
    const uint32_t *px;       // record arena
    const uint64_t *img_off;  // [n+1] first record of every image
    const uint32_t *rowptr;   // [n][nstrips][H+1] record offsets of the (strip, row) cells inside their image
    int64_t first, count;
    int W, H;
    int nstrips, strip_w;     // column strips of the record layout
    const uint8_t *slabs;
    const uint4 *iv;
    int n_iv, iv_in_lds, plane8;
    int xlo, cols, ylo, nrows, pitch, nwpr, slab_rows, n_slabs, near_off, slab_bytes;
    int thr;
    int group;                // targets a wave takes per grab
    int nparts;               // partitions of the targets, each with its own work counter
    uint32_t psize;           // targets per partition (the last one may hold fewer)
    uint32_t *work;           // [n_slabs][nparts] x 32 dwords: next target of the partition (zero before the launch)
    int plane_global;         // sparse_scan16_kernel: the plane is read from global memory (the LDS image is the band map alone)
    int ux0, g64, rowb64, map64_bytes;   // sparse_scan16_kernel's two-orientation band map (SparseMaskPlan); its images follow the n_slabs slab images
    int use16;                // scan through the 16-bit stage-1 stream (sparse_scan16_kernel) instead of the 4-byte records
    unsigned long long *stat; // pilot launches of sparse_scan16_kernel: [0] += records visited, [1] += records queued for a round; else NULL
    int32_t *out;             // [count] score | mirrored << 31 (written directly when there is one slab and no partial)
    int32_t *partial;         // [n_slabs][count][18] per-transform counts of every slab, or NULL
};

// ---- multi-mask sparse scan: a GROUP of up to kSparseGroupMasks masks shares one pass over the records ----------------------
// Per group, over the UNION bounding box of its masks: one u8 plane per mask (interval-record index per mask pixel, the
// single-mask layout), the masks' interval tables (256 x 16 B each, copied to LDS) and ONE transposed band map
//   tmap[cell row][cell column][7 x 64 hue bins] (u8): bit j = mask j of the group accepts, under some xy-shift, a code
//   of that bin (class x floor(64 x ratio)) at some position of the 4-row x 8-column cell
// so that stage 1 (record -> row, band, cell) is done once per record and ONE byte per orientation names the masks it
// can matter to.
constexpr int kSparseGroupMasks = 8;
#ifndef SP_BINS_LOG2
#define SP_BINS_LOG2 7
#endif
constexpr int kSparseGroupBinsLog2 = SP_BINS_LOG2;
constexpr int kSparseGroupBins = 1 << kSparseGroupBinsLog2;   // hue bins per class in the group's map (7 x 64 bytes per cell; class 0 is never set)
constexpr int kSparseGroupIv = 3072;  // interval records of a group's masks together (48 KB of LDS); groups are cut accordingly
struct SparseGroupPlan {
    int first_mask = 0, n_masks = 0;
    bool ok = false;          // representable (the masks' interval tables fit the LDS table, plane offsets < 2^20)
    bool plane8 = false;      // every mask has <= 255 interval records: one byte per plane cell, else two
    int xlo = 0, cols = 0, ylo = 0, nrows = 0, pitch = 0, nwpr = 0;
    int plane_cells = 0;      // cells of one mask's plane (multiple of 16)
    int n_iv = 0;             // interval records of the group (the masks' tables back to back)
    int iv_base[kSparseGroupMasks] = {0, 0, 0, 0, 0, 0, 0, 0};   // first record of mask j's table
    size_t planes_off = 0, tmap_off = 0, iv_off = 0;   // offsets into cds_maskset::d_sg_planes / d_sg_tmap (bytes) / d_sg_iv (elements)
    size_t tmap_bytes = 0;
    size_t umap_off = 0;      // the union band map (u32 per cell) follows the fine map in d_sg_tmap
    int umap_words = 0;       // 0: not used (it would not leave room for two workgroups per CU)
};

struct SparseMultiParams {
    const uint32_t *px;
    const uint64_t *img_off;
    const uint32_t *rowptr;
    int64_t first, count;
    int W, H, nstrips, strip_w;
    const void *planes;       // [n_masks][plane_cells] u8 or u16
    const uint8_t *tmap;      // [cell rows][nwpr][7 * kSparseGroupBins]
    const uint4 *iv;          // [n_iv]
    const uint32_t *umap;     // [umap_words] union of the masks' coarse band maps, one word per cell
    int umap_words;
    int n_iv, iv_base[kSparseGroupMasks];
    int n_masks, plane_cells, tmap_bytes;
    int xlo, cols, ylo, nrows, pitch, nwpr;
    int thr, group, nparts;
    uint32_t psize;
    uint32_t *work;           // [nparts] x 32 dwords, zero before the launch
    int32_t *out;             // score | mirrored << 31 of mask j, target t at out[j * out_stride + t]
    int64_t out_stride;
};

}  // namespace cds

struct cds_sparse_library {
    cds_context *ctx = nullptr;
    int W = 0, H = 0;
    int nstrips = 1, strip_w = 0;   // an image is stored strip by strip (column strips), a strip row by row
    int64_t cap_images = 0, cap_px = 0;
    int64_t n_images = 0, n_px = 0;
    bool growable = false;          // an append that finds a capacity lacking enlarges the arrays (cds_sparse_library_set_growable): the
                                    // counterpart of the reference's MIP cache, which sizes itself with use (cmd/CachedMIPsUtils.java:60-100)
    bool want_px16 = true;          // (CDS_SPARSE_STREAM16 at creation; the stream is dropped when it would not fit, see px16_fits)
    int min_threshold = 0;          // pixels whose maximum channel is <= this are not stored; searches below it are refused
    double density_limit = 0.25;    // appends / imports whose images average more signal pixels than this share of the frame are refused
    uint32_t *d_px = nullptr;
    uint16_t *d_px16 = nullptr;     // [cap_px + 2] the records' upper halves as a stream of their own (stage 1 of the single-mask scan reads
                                    // 2 instead of 4 bytes per visited record; derived from d_px by every append / import); NULL when the
                                    // device had no room for it or CDS_SPARSE_STREAM16=0: the scan then reads d_px as in round 3
    uint64_t *d_img_off = nullptr;  // [cap_images + 1]
    uint32_t *d_rowptr = nullptr;   // [cap_images][nstrips][H + 1]
    uint32_t *d_tmp = nullptr;      // append scratch: per-image totals of one batch
    int64_t tmp_images = 0;
    cds_library *stage = nullptr;   // dense staging slots of cds_sparse_library_append_host
    std::mutex mu;
};

struct cds_maskset {
    cds_context *ctx = nullptr;
    cds_pixmatch_params params{};

    int W = 0, H = 0, n_masks = 0;
    int group = 1;               // masks per launch; the chunk arrays are laid out per group (cds_maskset_create)
    bool batch = false;          // records are in the batch kernel's layout
    std::vector<int32_t> sizes;  // query size P per mask (includes class-less pixels)
    int m0 = 0, m1 = 0, cx0 = 0, wc = 0;
    uint32_t *d_bandptr = nullptr;
    std::vector<uint32_t> h_bandptr;  // host copy (slice planning)
    uint4 *d_chunks = nullptr;
    int nbands = 0;
    uint32_t *d_px_pos = nullptr;
    uint4 *d_px_iv = nullptr;
    size_t n_px = 0;
    // The dense kernel's record layout (bands, chunks, d_px_*) is built by the first search over a dense library
    // (cds_host.hip: ensure_dense_layout) from the pixel lists kept here until then: a mask set that only ever scans
    // sparse libraries -- the processor's mask sets -- never pays for it (half of its preparation time).
    mutable std::vector<std::vector<cds::MaskPx>> lazy_px;
    int px_xlo = 0, px_xhi = 0;      // column range [xlo, xhi) of all device pixels of all masks
    mutable std::mutex dense_mu;
    mutable std::atomic<bool> dense_ready{false};
    // sparse scan (cds_sparse.hip): per mask a plan, slab images and interval tables
    std::vector<cds::SparseMaskPlan> sp_plans;
    uint8_t *d_sp_slabs = nullptr;
    uint4 *d_sp_iv = nullptr;
    uint32_t *d_sp_work = nullptr;   // work counters of one launch
    // The single-mask slab images and interval tables go to the device when a single-mask launch first needs them: a mask set of the
    // processor is scanned by the multi-mask kernel only (its groups have their own arrays) and never pays for that upload (20 MB per
    // 128 masks).  Until then they wait here (guarded by sp_mu).
    mutable hipEvent_t sp_last_use = nullptr;   // behind the last sparse launch that reads this set's device blocks: waited for before they are recycled
    mutable std::vector<std::vector<uint8_t>> sp_host_slabs;
    mutable std::vector<std::vector<uint4>> sp_host_iv;
    mutable size_t sp_slab_total = 0, sp_iv_total = 0;
    mutable bool sp_uploaded = false;
    std::vector<cds::SparseGroupPlan> sg_plans;   // multi-mask scan: groups of consecutive masks (empty for a single mask)
    uint8_t *d_sg_planes = nullptr, *d_sg_tmap = nullptr;
    uint4 *d_sg_iv = nullptr;
    mutable std::mutex sp_mu;        // one sparse search at a time per mask set (the work counters are shared)
    // Which single-mask kernel scans a library that has the 16-bit stage-1 stream (cds_sparse.hip): decided per mask by a PILOT over
    // the first targets of the first large search -- the share of visited records that reach an evaluation round -- and remembered
    // for later searches of the same library (guarded by sp_mu).  -1 undecided, 0 the 4-byte-record kernel, 1 the stream kernel.
    mutable std::vector<int8_t> s16_choice;
    mutable const void *s16_lib = nullptr;
    mutable int64_t s16_lib_images = -1;
    mutable unsigned long long *d_s16_stat = nullptr;
    mutable double last_pilot_ratio = -1;   // queued / (visited x orientations) of the last pilot (diagnostic)
    mutable int last_scan_kernel = -1;      // kernel of the last single-mask launch: 0 / 1 as above
    // cds_score_pair: resident slots + device scores, created on first use
    mutable cds::PairSlotPool pair_pool;
    // timing of the last search
    mutable std::mutex mu;
    mutable double last_ms = 0;
    mutable int last_launches = 0;
};
