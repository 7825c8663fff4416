// The variant shard: kernel 2's inputs of a whole library in ONE file that the GPU decodes.
//
// CalculateGradientScoresCmd reads, per selected target, three image files through its variant suppliers (:541-586: the MIP, its
// GradientImage, its ZGapImage) -- 0.3 MB of zlib / PackBits per target and ~6 ms of a host core to inflate it; at configs[4]'s 230 K
// selected targets the decode is a minute on 16 cores for a quarter of a second of scoring.  The shard holds the same three images
// LOSSLESSLY in codes a wavefront can expand, so that a batch is pread() into pinned memory, copied, and decoded on the device:
//   * the target and its zgap image (8-bit RGB, black almost everywhere): RLE4 -- per row a list of runs `r | g << 8 | b << 16 | len << 24`
//     (len 1..255), the rows found through a table of H + 1 run indices.  A wave takes a row: 64 runs per step, an inclusive scan of the
//     lengths gives every run its first column, black runs are not written (the planes are cleared by one memset per batch);
//   * the gradient (16 bits, a distance ramp: in every GradientImage of the reference's test data neighbouring pixels of a row differ by
//     at most 1): DELTA2 -- per row its first value and two bits per pixel (0 same, 1 up, 2 down); a lane takes a word of 16 pixels, the
//     words' sums go through a wave scan, 32 bytes per lane are stored.  A gradient that is not such a ramp is kept RAW16.
//     (Superseded as the first choice by DIFF2, the non-zero second differences: see grad_encode.)
//     (An RGB image whose signal pixels all differ -- a MIP -- is shorter as SPANS: see spans_encode.)
// The reference's three test targets pack to 0.18-0.33 MB each, 0.27 MB on average (their PackBits TIFF + 16-bit PNG + PackBits TIFF: 0.41 MB)
// and the decode moves from 16 host cores to the GPU: 77.5 K targets/s from the page cache (21.1 GB/s of blobs),
// 17.7 K/s from the box's cold disk, where inflating the files gives 5.3 K/s (tools/gradscore_files_rate.py --real --shard; profiles/
// r05j_variant_shard.txt).  A slot filled from a shard is an ordinary slot of a dense library (image, gradient, zgap: bit-equal to what
// cds_library_load_files makes of the files; tests/test_mipio.py), so everything downstream is unchanged.
//
// File: [64-byte header] [blobs, 16-byte aligned] [index: n x {u64 offset, u32 bytes, u32 flags}]; little-endian.
// Blob: u32 flags, rgb_bytes, grad_bytes, zgap_bytes; then the three sections in that order.
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "cds_imageio.hpp"
#include "cds_internal.hpp"

namespace cds {
namespace {

constexpr uint32_t kVarGrad = CDS_VARIANT_HAS_GRADIENT, kVarDelta2 = CDS_VARIANT_GRADIENT_DELTA2, kVarZgap = CDS_VARIANT_HAS_ZGAP, kVarDiff2 = CDS_VARIANT_GRADIENT_DIFF2,
                   kVarRgbSpans = CDS_VARIANT_RGB_SPANS, kVarZgapSpans = CDS_VARIANT_ZGAP_SPANS;
constexpr char kShardMagic[8] = {'C', 'D', 'S', 'V', 'A', 'R', '0', '1'};

struct ShardHeader {
    char magic[8];
    uint32_t W, H;
    uint64_t n, index_off, file_bytes;
    uint64_t reserved[3];
};
static_assert(sizeof(ShardHeader) == 64, "shard header is 64 bytes");
struct ShardEntry {
    uint64_t off;
    uint32_t bytes, flags;
};
static_assert(sizeof(ShardEntry) == 16, "shard index entry is 16 bytes");

inline uint32_t delta2_row_words(int W) { return 1u + (uint32_t)((W + 15) / 16); }
inline uint32_t raw16_row_words(int W) { return (uint32_t)((W + 1) / 2); }

// ---- host: packing -------------------------------------------------------------------------------------------------------------
void rle4_encode(const uint8_t *rgb, int W, int H, std::vector<uint32_t> &blob) {
    const size_t base = blob.size();
    blob.resize(base + (size_t)H + 1);
    uint32_t nruns = 0;
    for (int y = 0; y < H; y++) {
        blob[base + (size_t)y] = nruns;
        const uint8_t *p = rgb + (size_t)y * W * 3;
        int x = 0;
        while (x < W) {
            const uint32_t c = (uint32_t)p[3 * x] | (uint32_t)p[3 * x + 1] << 8 | (uint32_t)p[3 * x + 2] << 16;
            int len = 1;
            while (x + len < W && len < 255 && p[3 * (x + len)] == p[3 * x] && p[3 * (x + len) + 1] == p[3 * x + 1] && p[3 * (x + len) + 2] == p[3 * x + 2]) len++;
            blob.push_back(c | (uint32_t)len << 24);
            nruns++;
            x += len;
        }
    }
    blob[base + (size_t)H] = nruns;
}

// SPANS: the other code of an RGB image -- per row the spans `x0 | len << 16` of its non-black pixels and their bytes (3 per pixel, row after row,
// padded to a word): `u32 row_span[H + 1]` (span indices), `u32 row_pixel[H + 1]` (pixels in front of the row), the spans, the bytes.  A MIP's signal
// pixels all differ from their neighbours (RLE4 then spends 4 bytes on each); a zgap image, a max filter's output, has long runs (RLE4 wins).
void spans_encode(const uint8_t *rgb, int W, int H, std::vector<uint32_t> &blob) {
    const size_t base = blob.size();
    blob.resize(base + 2 * ((size_t)H + 1));
    std::vector<uint8_t> lit;
    uint32_t nspans = 0, npx = 0;
    for (int y = 0; y < H; y++) {
        blob[base + (size_t)y] = nspans;
        blob[base + (size_t)H + 1 + (size_t)y] = npx;
        const uint8_t *p = rgb + (size_t)y * W * 3;
        int x = 0;
        while (x < W) {
            while (x < W && !(p[3 * x] | p[3 * x + 1] | p[3 * x + 2])) x++;
            if (x >= W) break;
            const int x0 = x;
            while (x < W && (p[3 * x] | p[3 * x + 1] | p[3 * x + 2])) x++;
            blob.push_back((uint32_t)x0 | (uint32_t)(x - x0) << 16);
            lit.insert(lit.end(), p + 3 * x0, p + 3 * x);
            nspans++, npx += (uint32_t)(x - x0);
        }
    }
    blob[base + (size_t)H] = nspans;
    blob[base + 2 * (size_t)H + 1] = npx;
    lit.resize((lit.size() + 3) & ~(size_t)3, 0);
    const size_t at = blob.size();
    blob.resize(at + lit.size() / 4);
    memcpy(blob.data() + at, lit.data(), lit.size());
}
// the shorter of the two codes of an RGB image; true when it is SPANS
bool rgb_encode(const uint8_t *rgb, int W, int H, std::vector<uint32_t> &blob) {
    const size_t base = blob.size();
    rle4_encode(rgb, W, H, blob);
    const size_t rle_words = blob.size() - base;
    static const bool no_spans = getenv("CDS_VARIANT_NO_SPANS") != nullptr;   // (A/B runs of the decoders: run lengths only)
    if (no_spans) return false;
    std::vector<uint32_t> alt;
    spans_encode(rgb, W, H, alt);
    if (alt.size() >= rle_words) return false;
    blob.resize(base);
    blob.insert(blob.end(), alt.begin(), alt.end());
    return true;
}

// returns the flag bits of the gradient section it appended: the smallest of the three codes
//   DIFF2   the second mixed difference e = g[y][x] - g[y-1][x] - g[y][x-1] + g[y-1][x-1] (mod 2^16, outside the image = 0) where it is not 0:
//           per row a list of `x | e << 16`, found through a table of H + 1 list indices.  A distance ramp is piecewise linear: in the reference's
//           GradientImages 99.7-99.9 % of the pixels have e = 0 (a few thousand exceptions: the image's first row and column, the ridges) --
//           the device gets the image back as the 2-D prefix sum of the exceptions, a row pass and a column pass;
//   DELTA2  two bits per pixel, for a ramp whose exceptions are many;
//   RAW16   anything else.
uint32_t grad_encode(const uint16_t *g, int W, int H, std::vector<uint32_t> &blob) {
    bool ramp = true;
    for (int y = 0; y < H && ramp; y++) {
        const uint16_t *r = g + (size_t)y * W;
        for (int x = 1; x < W; x++) {
            const int d = (int)r[x] - (int)r[x - 1];
            if (d > 1 || d < -1) { ramp = false; break; }
        }
    }
    const size_t other_words = (size_t)H * (ramp ? delta2_row_words(W) : raw16_row_words(W));
    {   // DIFF2, given up as soon as it is longer than the alternative
        const size_t base = blob.size();
        blob.resize(base + (size_t)H + 1);
        uint32_t n = 0;
        bool fits = true;
        for (int y = 0; y < H && fits; y++) {
            blob[base + (size_t)y] = n;
            const uint16_t *r = g + (size_t)y * W, *u = y ? r - W : nullptr;
            for (int x = 0; x < W; x++) {
                const uint32_t e = ((uint32_t)r[x] - (u ? u[x] : 0u) - (x ? r[x - 1] : 0u) + ((u && x) ? u[x - 1] : 0u)) & 0xffffu;
                if (e) {
                    blob.push_back((uint32_t)x | e << 16);
                    n++;
                }
            }
            fits = (size_t)H + 1 + n < other_words;
        }
        if (fits) {
            blob[base + (size_t)H] = n;
            return kVarGrad | kVarDiff2;
        }
        blob.resize(base);
    }
    if (ramp) {
        const uint32_t rw = delta2_row_words(W);
        const size_t base = blob.size();
        blob.resize(base + (size_t)H * rw, 0u);
        for (int y = 0; y < H; y++) {
            const uint16_t *r = g + (size_t)y * W;
            uint32_t *o = blob.data() + base + (size_t)y * rw;
            o[0] = r[0];
            for (int x = 1; x < W; x++) {
                const int d = (int)r[x] - (int)r[x - 1];
                o[1 + (x >> 4)] |= (uint32_t)(d == 1 ? 1 : d == -1 ? 2 : 0) << (2 * (x & 15));
            }
        }
        return kVarGrad | kVarDelta2;
    }
    const uint32_t rw = raw16_row_words(W);
    const size_t base = blob.size();
    blob.resize(base + (size_t)H * rw, 0u);
    for (int y = 0; y < H; y++) {
        const uint16_t *r = g + (size_t)y * W;
        uint32_t *o = blob.data() + base + (size_t)y * rw;
        for (int x = 0; x < W; x++) o[x >> 1] |= (uint32_t)r[x] << (16 * (x & 1));
    }
    return kVarGrad;
}

void pack_target(const uint8_t *rgb, const uint16_t *grad, const uint8_t *zgap, int W, int H, std::vector<uint32_t> &blob, uint32_t &flags) {
    blob.clear();
    blob.resize(4, 0u);
    flags = rgb_encode(rgb, W, H, blob) ? kVarRgbSpans : 0u;
    const size_t a = blob.size();
    if (grad) flags |= grad_encode(grad, W, H, blob);
    const size_t b = blob.size();
    if (zgap) flags |= kVarZgap | (rgb_encode(zgap, W, H, blob) ? kVarZgapSpans : 0u);
    blob[0] = flags;
    blob[1] = (uint32_t)((a - 4) * 4);
    blob[2] = (uint32_t)((b - a) * 4);
    blob[3] = (uint32_t)((blob.size() - b) * 4);
}

// A blob as it came from a file: do its section sizes and row tables fit its length?  (The device clamps columns; it trusts these.)
bool blob_ok(const uint32_t *blob, size_t bytes, uint32_t flags_in_index, int W, int H) {
    if (bytes < 16 || (bytes & 3)) return false;
    const uint32_t flags = blob[0], rb = blob[1], gb = blob[2], zb = blob[3];
    if (flags != flags_in_index || (rb & 3) || (gb & 3) || (zb & 3)) return false;
    if ((uint64_t)16 + rb + gb + zb != bytes) return false;
    auto rle_ok = [&](const uint32_t *sec, uint32_t sb) {
        if (sb < ((uint32_t)H + 1) * 4) return false;
        const uint32_t nruns = sb / 4 - ((uint32_t)H + 1);
        if (sec[0] != 0 || sec[H] != nruns) return false;
        for (int y = 0; y < H; y++)
            if (sec[y] > sec[y + 1]) return false;
        return true;
    };
    auto spans_ok = [&](const uint32_t *sec, uint32_t sb) {
        const uint64_t tables = 2 * ((uint64_t)H + 1);
        if (sb < tables * 4) return false;
        const uint32_t nspans = sec[H], npx = sec[2 * H + 1];
        if (sec[0] != 0 || sec[H + 1] != 0 || (tables + nspans) * 4 + (((uint64_t)npx * 3 + 3) & ~(uint64_t)3) != sb) return false;
        for (int y = 0; y < H; y++)
            if (sec[y] > sec[y + 1] || sec[H + 1 + y] > sec[H + 2 + y]) return false;
        return true;
    };
    if ((flags & kVarZgapSpans) && !(flags & kVarZgap)) return false;
    if (!((flags & kVarRgbSpans) ? spans_ok(blob + 4, rb) : rle_ok(blob + 4, rb))) return false;
    if ((flags & kVarDelta2) && (flags & kVarDiff2)) return false;
    if ((flags & kVarGrad) && (flags & kVarDiff2)) {
        if (!rle_ok(blob + 4 + rb / 4, gb)) return false;   // (the same shape: a table of H + 1 list indices, then the lists)
    } else if (flags & kVarGrad) {
        const uint64_t want = (uint64_t)H * ((flags & kVarDelta2) ? delta2_row_words(W) : raw16_row_words(W)) * 4;
        if (gb != want) return false;
    } else if (gb != 0 || (flags & (kVarDelta2 | kVarDiff2))) return false;
    if (flags & kVarZgap) {
        if (!((flags & kVarZgapSpans) ? spans_ok(blob + 4 + rb / 4 + gb / 4, zb) : rle_ok(blob + 4 + rb / 4 + gb / 4, zb))) return false;
    } else if (zb != 0) return false;
    return true;
}

bool write_all(int fd, const void *p, size_t n, uint64_t at) {
    const uint8_t *q = static_cast<const uint8_t *>(p);
    while (n) {
        const ssize_t w = pwrite(fd, q, n, (off_t)at);
        if (w <= 0) return false;
        q += w, n -= (size_t)w, at += (uint64_t)w;
    }
    return true;
}
bool read_all(int fd, void *p, size_t n, uint64_t at) {
    uint8_t *q = static_cast<uint8_t *>(p);
    while (n) {
        const ssize_t r = pread(fd, q, n, (off_t)at);
        if (r <= 0) return false;
        q += r, n -= (size_t)r, at += (uint64_t)r;
    }
    return true;
}

// work(item) for item in [0, items), the calling thread included.  An exception in `work` (an allocation in a packer) does not leave its
// thread: the remaining items are skipped and the call returns false.
template <typename F>
bool run_threads(int nthreads, int64_t items, F &&work) {
    std::atomic<int64_t> next{0};
    std::atomic<bool> threw{false};
    auto worker = [&]() {
        try {
            for (int64_t i = next.fetch_add(1); i < items && !threw.load(); i = next.fetch_add(1)) work(i);
        } catch (...) {
            threw.store(true);
        }
    };
    std::vector<std::thread> pool;
    try {
        for (int t = 1; t < std::min<int64_t>(nthreads, items); t++) pool.emplace_back(worker);
    } catch (const std::exception &) {   // (fewer threads than asked for: the rest of the work is done by those that started)
    }
    worker();
    for (auto &th : pool) th.join();
    return !threw.load();
}

// ---- device: decoding ----------------------------------------------------------------------------------------------------------
struct VarDecodeParams {
    const uint8_t *stage;   // the batch on the device: u64 blob offset per target (relative to `stage`), then the blobs
    int64_t nb;
    int W, H;
    uint8_t *rgb, *zgap;    // slot of the batch's first target in each plane
    uint16_t *grad;
    int pitch, gpitch;
    size_t img_stride, grad_stride;
};

// RLE4 rows -> pitched RGB rows.  One wave per (target, row, plane); the planes were cleared, black runs are skipped.
__global__ __launch_bounds__(256) void var_rle4_kernel(VarDecodeParams p) {
    const int lane = threadIdx.x & 63;
    const int64_t items = p.nb * p.H * 2;
    const uint64_t *offs = reinterpret_cast<const uint64_t *>(p.stage);
    for (int64_t it = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); it < items; it += (int64_t)gridDim.x * 4) {   // wave-uniform
        const int plane = (int)(it & 1);
        const int64_t row = it >> 1;
        const int y = (int)(row % p.H);
        const int64_t i = row / p.H;
        const uint32_t *blob = reinterpret_cast<const uint32_t *>(p.stage + offs[i]);
        const uint32_t flags = blob[0];
        if (plane == 1 && !(flags & kVarZgap)) continue;
        const uint32_t *sec = blob + 4 + (plane ? (blob[1] + blob[2]) / 4 : 0);
        uint8_t *dst = (plane ? p.zgap : p.rgb) + (size_t)i * p.img_stride + (size_t)y * p.pitch;
        if (flags & (plane ? kVarZgapSpans : kVarRgbSpans)) {
            // SPANS: 64 spans per step, a lane each: an inclusive scan of the lengths gives every span its first literal pixel.  Then the
            // step's literal pixels 64 at a time, a lane each (neighbouring lanes read neighbouring bytes): a lane finds the span its
            // pixel belongs to by walking the step's spans (a row has a handful) and writes it to that span's column
            const uint32_t s0 = sec[y], s1 = sec[y + 1], npx = sec[2 * p.H + 1];
            const uint32_t *spans = sec + 2 * (p.H + 1);
            const uint8_t *lit = reinterpret_cast<const uint8_t *>(spans + sec[p.H]);
            uint32_t first = sec[p.H + 1 + y];   // literal pixel of the step's first span (wave-uniform)
            for (uint32_t c = s0; c < s1; c += 64) {
                const uint32_t sp = c + (uint32_t)lane < s1 ? spans[c + lane] : 0u;
                const int len = (int)(sp >> 16), x0 = (int)(sp & 0xffffu);
                int incl = len;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const int t = __shfl_up(incl, o, 64);
                    if (lane >= o) incl += t;
                }
                const int start = incl - len;                       // the span's first pixel among the step's literal pixels
                const int total = __shfl(incl, 63, 64);             // the step's literal pixels (wave-uniform)
                const int here = (int)min(64u, s1 - c);
                for (int q0 = 0; q0 < total; q0 += 64) {            // wave-uniform
                    const int q = q0 + lane;
                    int x = -1;
                    for (int j = 0; j < here; j++) {                // wave-uniform
                        const int sj = __shfl(start, j, 64), lj = __shfl(len, j, 64), xj = __shfl(x0, j, 64);
                        if (q >= sj && q < sj + lj) x = xj + (q - sj);
                    }
                    const uint32_t l = first + (uint32_t)q;
                    if (x >= 0 && x < p.W && l < npx) {
                        const uint8_t *g = lit + (size_t)l * 3;
                        uint8_t *d3 = dst + (size_t)x * 3;
                        d3[0] = g[0], d3[1] = g[1], d3[2] = g[2];
                    }
                }
                first += (uint32_t)total;
            }
            continue;
        }
        const uint32_t r0 = sec[y], r1 = sec[y + 1];
        const uint32_t *runs = sec + p.H + 1;
        int x = 0;   // first column of the step's first run (wave-uniform)
        for (uint32_t c = r0; c < r1 && x < p.W; c += 64) {
            const uint32_t r = c + (uint32_t)lane < r1 ? runs[c + lane] : 0u;
            const int len = (int)(r >> 24);
            int incl = len;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_up(incl, o, 64);
                if (lane >= o) incl += t;
            }
            const int xs = x + incl - len;
            const int n = (r & 0xffffffu) ? len : 0;
            const uint8_t cr = (uint8_t)r, cg = (uint8_t)(r >> 8), cb = (uint8_t)(r >> 16);
            for (int k = 0; __any(k < n); k++) {
                if (k < n && xs + k < p.W) {
                    uint8_t *q = dst + (size_t)(xs + k) * 3;
                    q[0] = cr, q[1] = cg, q[2] = cb;
                }
            }
            x += __shfl(incl, 63, 64);
        }
    }
}

// DELTA2 / RAW16 rows -> pitched 16-bit rows; DIFF2 rows -> the ROW prefix sums of their exceptions (var_colsum_kernel finishes them).
// One wave per (target, row).
__global__ __launch_bounds__(256) void var_grad_kernel(VarDecodeParams p) {
    __shared__ uint16_t rowbuf[4][2048];   // DIFF2: a wave's row of exceptions (W <= 2000)
    const int lane = threadIdx.x & 63;
    const int64_t items = p.nb * p.H;
    const uint64_t *offs = reinterpret_cast<const uint64_t *>(p.stage);
    for (int64_t it = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); it < items; it += (int64_t)gridDim.x * 4) {   // wave-uniform
        const int y = (int)(it % p.H);
        const int64_t i = it / p.H;
        const uint32_t *blob = reinterpret_cast<const uint32_t *>(p.stage + offs[i]);
        const uint32_t flags = blob[0];
        if (!(flags & kVarGrad)) continue;
        const uint32_t *sec = blob + 4 + blob[1] / 4;
        uint8_t *dst = reinterpret_cast<uint8_t *>(p.grad) + (size_t)i * p.grad_stride + (size_t)y * p.gpitch;
        if (flags & kVarDiff2) {
            // the row's exceptions scattered into LDS, then an inclusive scan along the row (mod 2^16), two pixels per lane and step
            uint16_t *rb = rowbuf[threadIdx.x >> 6];
            const int wpad = p.gpitch / 2;   // pixels per pitched row (even, >= W): the padding is written as 0
            for (int x = lane * 2; x < wpad; x += 128) *reinterpret_cast<uint32_t *>(rb + x) = 0u;
            __builtin_amdgcn_wave_barrier();
            const uint32_t e0 = sec[y], e1 = sec[y + 1];
            const uint32_t *ex = sec + p.H + 1;
            for (uint32_t e = e0 + (uint32_t)lane; e < e1; e += 64) {
                const uint32_t v = ex[e];
                if ((int)(v & 0xffffu) < p.W) rb[v & 0xffffu] = (uint16_t)(v >> 16);
            }
            __builtin_amdgcn_wave_barrier();
            uint32_t carry = 0;
            for (int x0 = 0; x0 < wpad; x0 += 128) {
                const int x = x0 + 2 * lane;
                const uint32_t pr = x < wpad ? *reinterpret_cast<const uint32_t *>(rb + x) : 0u;
                const uint32_t a = pr & 0xffffu, b2 = pr >> 16;
                uint32_t incl = a + b2;   // (sums may exceed 16 bits on the way: only their low halves are kept)
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const uint32_t t = (uint32_t)__shfl_up((int)incl, o, 64);
                    if (lane >= o) incl += t;
                }
                const uint32_t before = carry + incl - (a + b2);
                uint32_t lo = (before + a) & 0xffffu, hi = (before + a + b2) & 0xffffu;
                if (x >= p.W) lo = 0;
                if (x + 1 >= p.W) hi = 0;
                if (x < wpad) *reinterpret_cast<uint32_t *>(dst + (size_t)x * 2) = lo | hi << 16;
                carry += (uint32_t)__shfl((int)incl, 63, 64);
            }
            __builtin_amdgcn_wave_barrier();
        } else if (flags & kVarDelta2) {
            const int nw = (p.W + 15) / 16;
            const uint32_t *rowp = sec + (size_t)y * (1 + nw);
            int carry = (int)(rowp[0] & 0xffffu);   // the value in front of the step's first word
            for (int w0 = 0; w0 < nw; w0 += 64) {
                const int w = w0 + lane;
                const uint32_t word = w < nw ? rowp[1 + w] : 0u;
                const uint32_t lo = word & 0x55555555u, hi = (word >> 1) & 0x55555555u;
                const int s = __popc(lo & ~hi) - __popc(hi & ~lo);
                int incl = s;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const int t = __shfl_up(incl, o, 64);
                    if (lane >= o) incl += t;
                }
                if (w < nw) {
                    int v = carry + incl - s;
                    uint32_t out[8];
#pragma unroll
                    for (int k = 0; k < 16; k++) {
                        const uint32_t cde = (word >> (2 * k)) & 3u;
                        v += (cde == 1u) - (cde == 2u);
                        const uint32_t val = (16 * w + k < p.W) ? ((uint32_t)v & 0xffffu) : 0u;
                        if (k & 1) out[k >> 1] |= val << 16;
                        else out[k >> 1] = val;
                    }
                    // (rows are 16-byte aligned and 16-byte padded: the second half of the row's last word may lie beyond the pitch)
                    uint4 *q = reinterpret_cast<uint4 *>(dst + (size_t)w * 32);
                    q[0] = make_uint4(out[0], out[1], out[2], out[3]);
                    if (w * 32 + 32 <= p.gpitch) q[1] = make_uint4(out[4], out[5], out[6], out[7]);
                }
                carry += __shfl(incl, 63, 64);
            }
        } else {
            const int rw = (p.W + 1) / 2;
            const uint32_t *rowp = sec + (size_t)y * rw;
            for (int j = lane; j < rw; j += 64) {
                uint32_t word = rowp[j];
                if (2 * j + 1 >= p.W) word &= 0xffffu;
                reinterpret_cast<uint32_t *>(dst)[j] = word;
            }
        }
    }
}

// DIFF2, second pass: prefix sums down the columns (mod 2^16), in place.  One wave per (target, 128 columns): two pixels per lane, the
// rows' loads do not depend on the sums, so they run ahead of the adds.
__global__ __launch_bounds__(256) void var_colsum_kernel(VarDecodeParams p) {
    const int lane = threadIdx.x & 63;
    const int chunks = (p.W + 127) / 128;
    const int64_t items = p.nb * chunks;
    const uint64_t *offs = reinterpret_cast<const uint64_t *>(p.stage);
    for (int64_t it = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); it < items; it += (int64_t)gridDim.x * 4) {   // wave-uniform
        const int64_t i = it / chunks;
        const int x = (int)(it % chunks) * 128 + 2 * lane;
        const uint32_t flags = *reinterpret_cast<const uint32_t *>(p.stage + offs[i]);
        if ((flags & (kVarGrad | kVarDiff2)) != (kVarGrad | kVarDiff2) || x >= p.W) continue;
        uint8_t *col = reinterpret_cast<uint8_t *>(p.grad) + (size_t)i * p.grad_stride + (size_t)x * 2;
        uint32_t lo = 0, hi = 0;
        constexpr int U = 8;
        for (int y0 = 0; y0 < p.H; y0 += U) {
            uint32_t v[U];
#pragma unroll
            for (int k = 0; k < U; k++) v[k] = y0 + k < p.H ? *reinterpret_cast<const uint32_t *>(col + (size_t)(y0 + k) * p.gpitch) : 0u;
#pragma unroll
            for (int k = 0; k < U; k++) {
                if (y0 + k >= p.H) break;
                lo = (lo + (v[k] & 0xffffu)) & 0xffffu;
                hi = (hi + (v[k] >> 16)) & 0xffffu;
                *reinterpret_cast<uint32_t *>(col + (size_t)(y0 + k) * p.gpitch) = lo | hi << 16;
            }
        }
    }
}

}  // namespace
}  // namespace cds

using namespace cds;

extern "C" int cds_library_ensure_variants(cds_library *lib);   // cds_host.hip

struct cds_variant_writer {
    int fd = -1;
    int W = 0, H = 0;
    uint64_t pos = 64;
    std::vector<ShardEntry> idx;
    bool failed = false;
    ~cds_variant_writer() { if (fd >= 0) close(fd); }
};
struct cds_variant_shard {
    int fd = -1;
    int W = 0, H = 0;
    uint64_t file_bytes = 0;
    std::vector<ShardEntry> idx;
    ~cds_variant_shard() { if (fd >= 0) close(fd); }
};

#define CDS_NOEXCEPT_BEGIN try {
#define CDS_NOEXCEPT_END                                                                  \
    } catch (const std::bad_alloc &) { return fail(CDS_ERR_OUT_OF_MEMORY, "host allocation failed"); } \
    catch (const std::exception &e) { return fail(CDS_ERR_INVALID_ARGUMENT, e.what()); }

// blobs packed by `pack(i, blob, flags)` for i in [0, n), `threads` at a time, written in order
template <typename F>
static int writer_append(cds_variant_writer *w, int64_t n, int32_t threads, F &&pack, std::string &first_err) {
    const int nthreads = std::max(1, std::min(threads > 0 ? threads : (int)std::thread::hardware_concurrency(), 64));
    const int64_t batch = std::max<int64_t>(2 * nthreads, 16);
    std::vector<std::vector<uint32_t>> blobs((size_t)batch);
    std::vector<uint32_t> flags((size_t)batch);
    std::mutex mu;
    // a call is a transaction: when a target of a later batch fails, what the call has written already is forgotten (the index and the
    // write position go back; the bytes are overwritten by the next append or left behind the last blob)
    const size_t idx_before = w->idx.size();
    const uint64_t pos_before = w->pos;
    struct Rollback {
        cds_variant_writer *w;
        size_t n;
        uint64_t pos;
        bool armed = true;
        ~Rollback() {
            if (armed) w->idx.resize(n), w->pos = pos;
        }
    } rollback{w, idx_before, pos_before};
    for (int64_t b0 = 0; b0 < n; b0 += batch) {
        const int64_t nb = std::min(batch, n - b0);
        const bool ran = run_threads(nthreads, nb, [&](int64_t i) {
            std::string err;
            if (!pack(b0 + i, blobs[(size_t)i], flags[(size_t)i], err)) {
                std::lock_guard<std::mutex> lk(mu);
                if (first_err.empty()) first_err = err;
            }
        });
        if (!ran && first_err.empty()) {
            first_err = "out of host memory while packing";
            return CDS_ERR_OUT_OF_MEMORY;
        }
        if (!first_err.empty()) return CDS_ERR_INVALID_ARGUMENT;
        for (int64_t i = 0; i < nb; i++) {
            const std::vector<uint32_t> &b = blobs[(size_t)i];
            w->pos = (w->pos + 15) & ~(uint64_t)15;
            if (b.size() * 4 > 0xffffffffull || !write_all(w->fd, b.data(), b.size() * 4, w->pos)) {
                w->failed = true;
                first_err = "cannot write the variant shard";
                return CDS_ERR_INVALID_ARGUMENT;
            }
            w->idx.push_back(ShardEntry{w->pos, (uint32_t)(b.size() * 4), flags[(size_t)i]});
            w->pos += b.size() * 4;
        }
    }
    rollback.armed = false;
    return CDS_OK;
}

extern "C" {

int cds_variant_shard_create(const char *path, int32_t W, int32_t H, cds_variant_writer **out) {
    CDS_NOEXCEPT_BEGIN
    if (!path || !out) return fail(CDS_ERR_INVALID_ARGUMENT, "NULL argument");
    if (!image_size_ok(W, H)) return fail(CDS_ERR_UNSUPPORTED, kImageSizeMessage);
    const int fd = open(path, O_CREAT | O_TRUNC | O_WRONLY | O_CLOEXEC, 0644);
    if (fd < 0) return failf(CDS_ERR_INVALID_ARGUMENT, "cannot create %s", path);
    ShardHeader h{};   // (no magic until cds_variant_shard_finish: a file that was never finished does not open)
    if (!write_all(fd, &h, sizeof h, 0)) {
        close(fd);
        return failf(CDS_ERR_INVALID_ARGUMENT, "cannot write to %s", path);
    }
    cds_variant_writer *w = new (std::nothrow) cds_variant_writer();
    if (!w) {
        close(fd);
        return fail(CDS_ERR_OUT_OF_MEMORY, "out of host memory");
    }
    w->fd = fd, w->W = W, w->H = H;
    *out = w;
    return CDS_OK;
    CDS_NOEXCEPT_END
}

int cds_variant_shard_append(cds_variant_writer *w, int64_t n, const uint8_t *rgb, const uint16_t *gradients, const uint8_t *zgaps, int32_t threads) {
    CDS_NOEXCEPT_BEGIN
    if (!w || n < 0 || (n > 0 && !rgb)) return fail(CDS_ERR_INVALID_ARGUMENT, "NULL argument");
    if (w->failed) return fail(CDS_ERR_INVALID_ARGUMENT, "an earlier write to this shard failed");
    const size_t px = (size_t)w->W * w->H;
    std::string err;
    const int rc = writer_append(w, n, threads, [&](int64_t i, std::vector<uint32_t> &blob, uint32_t &flags, std::string &) {
        pack_target(rgb + (size_t)i * 3 * px, gradients ? gradients + (size_t)i * px : nullptr, zgaps ? zgaps + (size_t)i * 3 * px : nullptr, w->W, w->H, blob, flags);
        return true;
    }, err);
    return rc ? fail(rc, err) : CDS_OK;
    CDS_NOEXCEPT_END
}

int cds_variant_shard_append_files(cds_variant_writer *w, int64_t n, const char *const *target_paths, const char *const *gradient_paths,
                                   const char *const *zgap_paths, int32_t threads) {
    CDS_NOEXCEPT_BEGIN
    if (!w || n < 0 || (n > 0 && !target_paths)) return fail(CDS_ERR_INVALID_ARGUMENT, "NULL argument");
    if (w->failed) return fail(CDS_ERR_INVALID_ARGUMENT, "an earlier write to this shard failed");
    for (int64_t i = 0; i < n; i++)
        if (!target_paths[i]) return fail(CDS_ERR_INVALID_ARGUMENT, "a target path is NULL");
    const int W = w->W, H = w->H;
    const size_t px = (size_t)W * H;
    std::string err;
    const int rc = writer_append(w, n, threads, [&](int64_t i, std::vector<uint32_t> &blob, uint32_t &flags, std::string &e) {
        thread_local std::vector<uint8_t> t, z;
        thread_local std::vector<uint16_t> g;
        const char *gp = gradient_paths ? gradient_paths[i] : nullptr, *zp = zgap_paths ? zgap_paths[i] : nullptr;
        t.resize(3 * px);
        const char *what = target_paths[i];
        bool ok = decode_file_into(what, W, H, 0, t.data(), e);
        if (ok && gp) {
            g.resize(px);
            ok = decode_file_into(what = gp, W, H, 1, g.data(), e);
        }
        if (ok && zp) {
            z.resize(3 * px);
            ok = decode_file_into(what = zp, W, H, 0, z.data(), e);
        }
        if (!ok) {
            e = std::string(what) + ": " + e;
            return false;
        }
        pack_target(t.data(), gp ? g.data() : nullptr, zp ? z.data() : nullptr, W, H, blob, flags);
        return true;
    }, err);
    return rc ? fail(rc, err) : CDS_OK;
    CDS_NOEXCEPT_END
}

int cds_variant_shard_finish(cds_variant_writer *w) {
    CDS_NOEXCEPT_BEGIN
    if (!w) return fail(CDS_ERR_INVALID_ARGUMENT, "NULL argument");
    std::unique_ptr<cds_variant_writer> own(w);
    bool ok = !w->failed;
    ShardHeader h{};
    memcpy(h.magic, kShardMagic, 8);
    h.W = (uint32_t)w->W, h.H = (uint32_t)w->H, h.n = w->idx.size();
    h.index_off = (w->pos + 15) & ~(uint64_t)15;
    h.file_bytes = h.index_off + w->idx.size() * sizeof(ShardEntry);
    if (ok && !w->idx.empty()) ok = write_all(w->fd, w->idx.data(), w->idx.size() * sizeof(ShardEntry), h.index_off);
    if (ok) ok = ftruncate(w->fd, (off_t)h.file_bytes) == 0;   // (cuts off what a failed append left behind; pads up to an empty index)
    if (ok) ok = write_all(w->fd, &h, sizeof h, 0);   // the header last: the file opens only once everything else is in place
    if (ok) ok = fsync(w->fd) == 0;
    return ok ? CDS_OK : fail(CDS_ERR_INVALID_ARGUMENT, "cannot write the variant shard");
    CDS_NOEXCEPT_END
}

int cds_variant_shard_open(const char *path, cds_variant_shard **out) {
    CDS_NOEXCEPT_BEGIN
    if (!path || !out) return fail(CDS_ERR_INVALID_ARGUMENT, "NULL argument");
    const int fd = open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) return failf(CDS_ERR_INVALID_ARGUMENT, "cannot open %s", path);
    std::unique_ptr<cds_variant_shard> s(new cds_variant_shard());
    s->fd = fd;
    auto bad = [&](const char *why) {   // (`s` closes the file)
        return failf(CDS_ERR_INVALID_ARGUMENT, "%s is not a variant shard: %s", path, why);
    };
    struct stat sb;
    ShardHeader h{};
    if (fstat(fd, &sb) != 0 || (uint64_t)sb.st_size < sizeof h || !read_all(fd, &h, sizeof h, 0)) return bad("no header");
    if (memcmp(h.magic, kShardMagic, 8) != 0) return bad("wrong magic (unfinished file?)");
    if (h.W > (uint32_t)kMaxW || h.H > (uint32_t)kMaxH || !image_size_ok((int)h.W, (int)h.H)) return bad(kImageSizeMessage);
    if (h.file_bytes > (uint64_t)sb.st_size || h.index_off < 64 || h.index_off > h.file_bytes || h.n > (h.file_bytes - h.index_off) / sizeof(ShardEntry))
        return bad("index outside the file");
    s->W = (int)h.W, s->H = (int)h.H, s->file_bytes = h.file_bytes;
    s->idx.resize((size_t)h.n);
    if (h.n && !read_all(fd, s->idx.data(), (size_t)h.n * sizeof(ShardEntry), h.index_off)) return bad("index unreadable");
    for (const ShardEntry &e : s->idx)
        if (e.off < 64 || (e.off & 15) || e.bytes < 16 || (e.bytes & 3) || e.off + e.bytes > h.index_off) return bad("a blob lies outside the file");
    *out = s.release();
    return CDS_OK;
    CDS_NOEXCEPT_END
}

int cds_variant_shard_info(const cds_variant_shard *s, int32_t *W, int32_t *H, int64_t *n_targets, int64_t *file_bytes) {
    if (!s) return fail(CDS_ERR_INVALID_ARGUMENT, "NULL argument");
    if (W) *W = s->W;
    if (H) *H = s->H;
    if (n_targets) *n_targets = (int64_t)s->idx.size();
    if (file_bytes) *file_bytes = (int64_t)s->file_bytes;
    return CDS_OK;
}

int cds_variant_shard_target_info(const cds_variant_shard *s, int64_t first, int64_t count, uint32_t *flags, uint32_t *bytes) {
    if (!s) return fail(CDS_ERR_INVALID_ARGUMENT, "NULL argument");
    if (first < 0 || count < 0 || first + count > (int64_t)s->idx.size()) return fail(CDS_ERR_INVALID_ARGUMENT, "target range outside the shard");
    for (int64_t i = 0; i < count; i++) {
        if (flags) flags[i] = s->idx[(size_t)(first + i)].flags;
        if (bytes) bytes[i] = s->idx[(size_t)(first + i)].bytes;
    }
    return CDS_OK;
}

void cds_variant_shard_close(cds_variant_shard *s) { delete s; }

}  // extern "C"

// images_only: of every blob only the header and the MIP's section are read, staged and decoded (the ingest of a sparse library wants the images
// of ALL targets and none of their variants: a third to a half of the bytes); the slots come out without gradient and zgap
static int load_variants_impl(cds_library *lib, int64_t first, const cds_variant_shard *s, const int64_t *targets, int64_t n, int32_t threads, bool images_only) {
    CDS_NOEXCEPT_BEGIN
    if (!lib || !s || (n > 0 && !targets)) return fail(CDS_ERR_INVALID_ARGUMENT, "NULL argument");
    if (first < 0 || n < 0 || first + n > lib->capacity) return fail(CDS_ERR_INVALID_ARGUMENT, "slot range outside the library");
    if (lib->W != s->W || lib->H != s->H) return fail(CDS_ERR_INVALID_ARGUMENT, "the shard's image size differs from the library's");
    for (int64_t i = 0; i < n; i++)
        if (targets[i] < 0 || targets[i] >= (int64_t)s->idx.size()) return fail(CDS_ERR_INVALID_ARGUMENT, "target index outside the shard");
    if (n == 0) return CDS_OK;
    if (!images_only) {
        const int rc = cds_library_ensure_variants(lib);
        if (rc) return rc;
    }
    cds_context *ctx = lib->ctx;
    CDS_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int nthreads = std::max(1, std::min(threads > 0 ? threads : (int)std::thread::hardware_concurrency(), 64));
    // bytes[i]: what is staged of target i -- its whole blob, or header + MIP section (their headers are read first: 16 bytes each)
    std::vector<uint32_t> bytes((size_t)n);
    for (int64_t i = 0; i < n; i++) bytes[(size_t)i] = s->idx[(size_t)targets[i]].bytes;
    if (images_only) {
        std::atomic<int> bad_header{0};
        const bool ran = run_threads(nthreads, n, [&](int64_t i) {
            const ShardEntry &e = s->idx[(size_t)targets[i]];
            uint32_t h[4];
            if (!read_all(s->fd, h, 16, e.off) || h[0] != e.flags || (h[1] & 3) || (uint64_t)16 + h[1] > e.bytes) bad_header.store(1);
            else bytes[(size_t)i] = 16 + h[1];
        });
        if (!ran) return fail(CDS_ERR_OUT_OF_MEMORY, "a reader thread failed");
        if (bad_header.load()) return fail(CDS_ERR_INVALID_ARGUMENT, "the variant shard is damaged: a target's blob does not match its index entry");
    }
    // a batch: as many targets as fit one staging buffer -- [u64 offsets][blobs, 16-byte aligned]
    constexpr int64_t kMaxBatch = 1024;
    size_t biggest = 0;
    for (int64_t i = 0; i < n; i++) biggest = std::max<size_t>(biggest, bytes[(size_t)i]);
    const size_t table = (size_t)kMaxBatch * 8;
    const size_t stage_bytes = std::max<size_t>((size_t)64 << 20, table + biggest + 16);
    struct Buffers {   // (device blocks from the context's cache: hipFree would wait for the scoring another thread has in flight)
        cds_context *ctx;
        uint8_t *h[2] = {nullptr, nullptr};
        void *d[2] = {nullptr, nullptr};
        hipEvent_t ev[2] = {nullptr, nullptr};
        bool busy[2] = {false, false};
        ~Buffers() {
            for (int k = 0; k < 2; k++) {
                if (busy[k]) hipEventSynchronize(ev[k]);
                pinned_put(h[k]);
                if (d[k]) cache_free(ctx, d[k]);
                if (ev[k]) hipEventDestroy(ev[k]);
            }
        }
    } B{ctx};
    const bool two = [&]() {   // (a call that fits one batch needs one pair of buffers)
        size_t used = table;
        for (int64_t i = 0; i < n; i++) {
            used += ((size_t)bytes[(size_t)i] + 15) & ~(size_t)15;
            if (used > stage_bytes || i >= kMaxBatch) return true;
        }
        return false;
    }();
    for (int k = 0; k < (two ? 2 : 1); k++) {
        if (!(B.h[k] = pinned_get(stage_bytes))) return fail(CDS_ERR_OUT_OF_MEMORY, "no pinned host memory for the shard's blobs");
        CDS_HIP(cache_alloc(ctx, &B.d[k], stage_bytes));
        CDS_HIP(hipEventCreateWithFlags(&B.ev[k], hipEventDisableTiming));
    }
    static const bool timed = getenv("CDS_TIME_LOAD") != nullptr;
    double t_read = 0, t_wait = 0;
    const auto t_call = std::chrono::steady_clock::now();
    std::vector<uint8_t> slot_flags((size_t)n);
    int64_t b0 = 0;
    for (int64_t k = 0; b0 < n; k++) {
        const int kk = two ? (int)(k & 1) : 0;
        // the batch's extent
        int64_t nb = 0;
        size_t used = table;
        std::vector<uint64_t> offs;
        while (b0 + nb < n && nb < kMaxBatch) {
            const size_t room = ((size_t)bytes[(size_t)(b0 + nb)] + 15) & ~(size_t)15;
            if (used + room > stage_bytes) break;
            offs.push_back(used);
            used += room;
            nb++;
        }
        const auto t0 = std::chrono::steady_clock::now();
        if (B.busy[kk]) {   // the copy that read this buffer two batches ago
            CDS_HIP(hipEventSynchronize(B.ev[kk]));
            B.busy[kk] = false;
        }
        const auto t1 = std::chrono::steady_clock::now();
        memcpy(B.h[kk], offs.data(), (size_t)nb * 8);
        std::atomic<int> bad{0};
        const bool ran = run_threads(nthreads, nb, [&](int64_t i) {
            const ShardEntry &e = s->idx[(size_t)targets[b0 + i]];
            uint32_t *dst = reinterpret_cast<uint32_t *>(B.h[kk] + offs[(size_t)i]);
            const uint32_t want = bytes[(size_t)(b0 + i)];
            if (!read_all(s->fd, dst, want, e.off)) { bad.store(1); return; }
            uint32_t flags = e.flags;
            if (images_only) {   // the staged blob is a blob of its own: the MIP alone
                if (dst[0] != e.flags || 16 + dst[1] != want) { bad.store(1); return; }
                flags = e.flags & kVarRgbSpans;
                dst[0] = flags, dst[2] = 0, dst[3] = 0;
            }
            if (!blob_ok(dst, want, flags, s->W, s->H)) bad.store(1);
        });
        if (!ran) return fail(CDS_ERR_OUT_OF_MEMORY, "a reader thread failed");
        if (bad.load()) return fail(CDS_ERR_INVALID_ARGUMENT, "the variant shard is damaged: a target's blob does not match its index entry");
        const auto t2 = std::chrono::steady_clock::now();
        t_wait += std::chrono::duration<double>(t1 - t0).count();
        t_read += std::chrono::duration<double>(t2 - t1).count();
        const int64_t slot0 = first + b0;
        bool any_grad = false, any_zgap = false, any_diff2 = false;
        for (int64_t i = 0; i < nb; i++) {
            const uint32_t f = images_only ? 0u : s->idx[(size_t)targets[b0 + i]].flags;
            slot_flags[(size_t)(b0 + i)] = (uint8_t)f;
            any_grad |= (f & kVarGrad) != 0, any_zgap |= (f & kVarZgap) != 0, any_diff2 |= (f & kVarDiff2) != 0;
        }
        CDS_HIP(hipMemcpyAsync(B.d[kk], B.h[kk], used, hipMemcpyHostToDevice, st));
        CDS_HIP(hipMemsetAsync(lib->d_rgb + (size_t)slot0 * lib->img_stride, 0, (size_t)nb * lib->img_stride, st));
        if (any_zgap) CDS_HIP(hipMemsetAsync(lib->d_zgap + (size_t)slot0 * lib->img_stride, 0, (size_t)nb * lib->img_stride, st));
        VarDecodeParams p{};
        p.stage = static_cast<const uint8_t *>(B.d[kk]);
        p.nb = nb, p.W = lib->W, p.H = lib->H;
        p.rgb = lib->d_rgb + (size_t)slot0 * lib->img_stride;
        p.zgap = lib->d_zgap ? lib->d_zgap + (size_t)slot0 * lib->img_stride : nullptr;                 // (images_only: the library may have no variant planes)
        p.grad = lib->d_grad ? reinterpret_cast<uint16_t *>(reinterpret_cast<uint8_t *>(lib->d_grad) + (size_t)slot0 * lib->grad_stride) : nullptr;
        p.pitch = lib->pitch, p.gpitch = lib->gpitch, p.img_stride = lib->img_stride, p.grad_stride = lib->grad_stride;
        const int64_t cap_grid = (int64_t)ctx->n_cus * 32;
        hipLaunchKernelGGL(var_rle4_kernel, dim3((unsigned)std::min<int64_t>((nb * lib->H * 2 + 3) / 4, cap_grid)), dim3(256), 0, st, p);
        if (any_grad) hipLaunchKernelGGL(var_grad_kernel, dim3((unsigned)std::min<int64_t>((nb * lib->H + 3) / 4, cap_grid)), dim3(256), 0, st, p);
        if (any_diff2) hipLaunchKernelGGL(var_colsum_kernel, dim3((unsigned)std::min<int64_t>((nb * ((lib->W + 127) / 128) + 3) / 4, cap_grid)), dim3(256), 0, st, p);
        CDS_HIP(hipGetLastError());
        CDS_HIP(hipEventRecord(B.ev[kk], st));
        B.busy[kk] = true;
        b0 += nb;
    }
    for (int k = 0; k < 2; k++)
        if (B.busy[k]) {
            CDS_HIP(hipEventSynchronize(B.ev[k]));
            B.busy[k] = false;
        }
    {
        std::lock_guard<std::mutex> lk(lib->mu);
        for (int64_t i = 0; i < n; i++) {
            lib->has_grad[(size_t)(first + i)] = (slot_flags[(size_t)i] & kVarGrad) ? 1 : 0;
            lib->has_zgap[(size_t)(first + i)] = (slot_flags[(size_t)i] & kVarZgap) ? 1 : 0;
            lib->zslice_ok[(size_t)(first + i)] = 0;   // (a plane converted from the slot's previous zgap image is stale)
        }
    }
    if (timed)
        fprintf(stderr, "[cds load_variants] %lld slots: %.3f s (reading %.3f s, waiting for the device %.3f s, %d threads)\n", (long long)n,
                std::chrono::duration<double>(std::chrono::steady_clock::now() - t_call).count(), t_read, t_wait, nthreads);
    return CDS_OK;
    CDS_NOEXCEPT_END
}

extern "C" {

int cds_library_load_variants(cds_library *lib, int64_t first, const cds_variant_shard *s, const int64_t *targets, int64_t n, int32_t threads) {
    return load_variants_impl(lib, first, s, targets, n, threads, false);
}
int cds_library_load_variant_images(cds_library *lib, int64_t first, const cds_variant_shard *s, const int64_t *targets, int64_t n, int32_t threads) {
    return load_variants_impl(lib, first, s, targets, n, threads, true);
}

}  // extern "C"
