// cds_sparse.hip -- kernel 1 over a SPARSE resident library: the signal pixels of every target instead of its frame.
//
// What it computes: exactly what cds_scan.hip computes (reference: API/cds/PixelMatchColorDepthSearchAlgorithm.java:166-263,
// predicate API/cds/AbstractColorDepthSearchAlgorithm.java:157-390) -- per (mask, target) and transform (9 xy-shifts x
// {plain, mirrored}) the number of mask pixels whose transformed target pixel passes the data threshold and lies within
// zTolerance in hue-ratio space; max over shifts; mirrored only if strictly greater.
//
// Why another layout.  A color-depth MIP is a few percent signal on black (the reference's LM fixtures: 0.3-9 %); a pixel
// without a strict maximum channel (black, grey, two equal maxima) has no hue class and can never match anything.  The
// dense scan streams 300 KB of mostly zeros per target to find the ~6 K pixels that matter; 1 M targets are 2 TB and do not
// fit one GPU.  Here the library keeps, per image, its signal pixels only, as 4-byte records in row-major order
// (strip-relative x | hue band | second channel | max channel -- independent of every search parameter) plus one row-pointer array:
// ~80 KB per synthetic target instead of 2.06 MB, so a 1 M-target library (80 GB) is RESIDENT in one MI355X's 288 GB.
//
// The scan turns around accordingly (target-centric instead of the reference's mask-centric gather):
//   * the MASK lives in LDS: a u8 / u16 plane over its bounding box (index of the pixel's acceptance-interval record,
//     0 = no mask pixel), the table of distinct interval records, and a BAND MAP: per cell of 4 rows x 8 columns of target
//     positions one word whose bit b says "some xy-shift maps a position of this cell onto a mask pixel that accepts a code
//     of hue band b" (30 bands = class x floor(5 * ratio); conservative, band_bits_of below).
//     A mask taller than the LDS budget is cut into slabs of rows; per-transform counts of the slabs are summed at the end.
//   * a WAVE streams one target's records (coalesced dword loads through a buffer descriptor, the rows a slab can reach in
//     a column strip are one contiguous run thanks to the row pointers), 64 pixels per step: row from the row-start flags
//     (ballot + mbcnt), the pixel's band (or a never-set bit below the data threshold), one band-map probe per orientation
//     "can any shift of this pixel land on a mask pixel of a compatible colour?" -- the survivors (1.5 % when the target's
//     colours are unrelated to the mask's, about half on overlapping real images) go to a per-wave LDS queue;
//   * queued pixels are evaluated 64 at a time with every lane busy: hue code = class << 29 | bits(second * v_rcp_f32(max)) >> 1
//     (the same code the dense kernel computes, so the host's interval tables apply unchanged and the integer range test
//     is bit-exact with the reference's double arithmetic, cds_host.hip), 9 plane lookups with immediate offsets, 9
//     interval fetches, 2 compares each, matches counted by s_bcnt1 of the ballot into wave-uniform (SGPR) counters;
//   * no barriers, no staging, no atomics on the data path: waves are independent, targets are drawn in small groups
//     from partition counters (persistent grid, every workgroup visits every slab), ~14 KB of HBM traffic per target.
//   * no MFMA: integer compare / count.
#include <algorithm>
#include <atomic>
#include <thread>
#include <system_error>
#include <climits>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <unordered_map>

#include "cds_internal.hpp"

namespace cds {

constexpr int SP_QCAP = 128;                          // queue entries per wave and orientation (appends <= 64, drained at >= 64)
constexpr int SP_QBYTES = 2 * SP_QCAP * 8;            // two queues of (u32 record, u32 plane offset): one 8-byte LDS access per entry
constexpr int SP_MAX_IV_LDS = 2048;                   // interval records kept in LDS (32 KB); larger tables are read from L2
constexpr int SP_MAX_SLABS = 256;
constexpr int SP_MAX_PARTS = 64;                      // work counters per slab (one per partition of the targets, 128 B apart)
constexpr int SP_PAD = 4;                             // zero rows / columns around a slab's plane (two shifts of 2)

// =================================================================================================
//                         dense resident slots  ->  sparse records
// =================================================================================================
struct BuildParams {
    const uint8_t *rgb;  // first source slot
    size_t img_stride;
    int pitch, W, H;
    int nstrips, strip_w;  // column strips of the record layout (strip_w is a multiple of 4 pixels)
    int64_t count;       // images of this batch
    int64_t dst_first;   // index of the batch's first image in the sparse library
    uint32_t *px;
    uint64_t *img_off;
    uint32_t *rowptr;
    uint32_t *tmp;       // [count] records per image
    uint64_t *total;     // [1] records of the whole batch (read back by the host)
    uint32_t min_thr;    // pixels whose maximum channel is <= this are not stored (cds_sparse_library_set_min_threshold; 0: every signal pixel)
};

// hue class (AbstractColorDepthSearchAlgorithm.java:195-225; the nibble table of cds_scan.hip), the two channels its ratio is
// made of and the hue band class * 5 + floor(5 * ratio) -- the ratio as the scan's evaluation computes it, second * v_rcp_f32(max),
// so that the band of a record and the code the interval test sees come from the same float; 0 for a pixel without a strict
// maximum channel
__device__ __forceinline__ uint32_t px_record(uint32_t r, uint32_t g, uint32_t b, uint32_t min_thr) {
    const uint32_t mx = max(r, max(g, b)), mn = min(r, min(g, b)), md = r + g + b - mx - mn;
    const uint32_t idx = (r > g ? 4u : 0u) | (g > b ? 8u : 0u) | (b > r ? 16u : 0u);
    const uint32_t cls = (0x03125460u >> idx) & 7u;
    const float hf = (float)md * __builtin_amdgcn_rcpf((float)mx);
    const uint32_t band = cls * 5u + (uint32_t)(hf * 5.0f);
    return (mx > md && mx > min_thr) ? (band << kRecBandShift) | (md << 8) | mx : 0u;
}

// One wave per image row.  An image is stored strip by strip (column strips of strip_w pixels), a strip row by row, a row
// by ascending x; rp = rowptr[img] is [nstrips][H + 1].  FILL = false: records of every (strip, row) cell (at least 1:
// the filler) -> rp[k][row + 1]; FILL = true: write them at img_off[img] + rp[k][row].
template <bool FILL>
__global__ __launch_bounds__(256) void sparse_rows_kernel(BuildParams p) {
    const int lane = threadIdx.x & 63;
    const int64_t wid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wid >= p.count * p.H) return;  // wave-uniform
    const int64_t img = wid / p.H;
    const int row = (int)(wid - img * p.H);
    const uint8_t *rowp = p.rgb + (size_t)img * p.img_stride + (size_t)row * p.pitch;
    uint32_t *rp = p.rowptr + (size_t)(p.dst_first + img) * p.nstrips * (p.H + 1);
    const int gpr = p.pitch / 12;   // 4-pixel groups per row (the padding columns are black)
    const int gps = p.strip_w / 4;  // groups per strip
    for (int k = 0; k < p.nstrips; k++) {
        uint32_t *dst = nullptr;
        if (FILL) dst = p.px + p.img_off[p.dst_first + img] + rp[(size_t)k * (p.H + 1) + row];
        const int gend = min(gpr, (k + 1) * gps);
        int n = 0;
        for (int g0 = k * gps; g0 < gend; g0 += 64) {
            const int g = g0 + lane;
            uint32_t a = 0, b = 0, c = 0;
            if (g < gend) {
                const uint32_t *q = reinterpret_cast<const uint32_t *>(rowp + (size_t)g * 12);
                a = q[0], b = q[1], c = q[2];
            }
            uint32_t rec[4];
            rec[0] = px_record(a & 255, (a >> 8) & 255, (a >> 16) & 255, p.min_thr);
            rec[1] = px_record(a >> 24, b & 255, (b >> 8) & 255, p.min_thr);
            rec[2] = px_record((b >> 16) & 255, b >> 24, c & 255, p.min_thr);
            rec[3] = px_record((c >> 8) & 255, (c >> 16) & 255, c >> 24, p.min_thr);
            int below = 0, all = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const unsigned long long bal = __ballot(rec[j] != 0);
                below += __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0));
                all += __popcll(bal);
            }
            if (FILL) {
                int pos = n + below;
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if (rec[j]) {
                        dst[pos] = rec[j] | ((uint32_t)(4 * g + j - k * p.strip_w) << kRecXShift) | (pos == 0 ? kRecRowFlag : 0u);
                        pos++;
                    }
            }
            n += all;
        }
        if (!FILL) {
            if (lane == 0) rp[(size_t)k * (p.H + 1) + row + 1] = (uint32_t)(n > 0 ? n : 1);
        } else if (n == 0 && lane == 0) {
            dst[0] = kRecRowFlag;  // filler of an empty cell
        }
    }
}

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t u = __shfl_up(v, o, 64);
        if (lane >= o) v += u;
    }
    return v;
}

// One workgroup (256 threads) per image: rowptr[img][k][1..H] (record counts) -> exclusive offsets in place, running on
// from strip to strip (rowptr[img][k][H] = rowptr[img][k + 1][0] = end of strip k); image total -> tmp[img].
__global__ __launch_bounds__(256) void sparse_rowscan_kernel(BuildParams p) {
    __shared__ uint32_t wsum[4];
    __shared__ uint32_t carry_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t *rp0 = p.rowptr + (size_t)(p.dst_first + blockIdx.x) * p.nstrips * (p.H + 1);
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int k = 0; k < p.nstrips; k++) {
        uint32_t *rp = rp0 + (size_t)k * (p.H + 1);
        // the counts sit one slot to the right of the offsets they turn into: read a tile, barrier, write it
        for (int base = 0; base < p.H; base += 256) {
            const int i = base + (int)threadIdx.x;
            const uint32_t v = i < p.H ? rp[i + 1] : 0u;
            const uint32_t inc = wave_incl_scan(v, lane);
            if (lane == 63) wsum[wave] = inc;
            __syncthreads();
            uint32_t off = carry_s;
            for (int w = 0; w < wave; w++) off += wsum[w];
            const uint32_t excl = off + inc - v;
            __syncthreads();  // every count of the tile has been read, carry_s and wsum have been consumed
            if (i < p.H) rp[i] = excl;
            if (threadIdx.x == 255) carry_s = off + inc;
            __syncthreads();
        }
        if (threadIdx.x == 0) rp[p.H] = carry_s;
    }
    if (threadIdx.x == 0) p.tmp[blockIdx.x] = carry_s;
}

// One workgroup (1024 threads): img_off[dst_first + i + 1] = img_off[dst_first] + sum of tmp[0..i]; batch total -> *total.
__global__ __launch_bounds__(1024) void sparse_imgscan_kernel(BuildParams p) {
    __shared__ unsigned long long wsum[16];
    __shared__ unsigned long long carry_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = p.img_off[p.dst_first];
    __syncthreads();
    const unsigned long long start = carry_s;
    for (int64_t base = 0; base < p.count; base += 1024) {
        const int64_t i = base + threadIdx.x;
        unsigned long long v = i < p.count ? p.tmp[i] : 0ull;
        const unsigned long long own = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned long long u = __shfl_up(v, o, 64);
            if (lane >= o) v += u;
        }
        if (lane == 63) wsum[wave] = v;
        __syncthreads();
        unsigned long long off = carry_s;
        for (int w = 0; w < wave; w++) off += wsum[w];
        __syncthreads();
        (void)own;
        if (i < p.count) p.img_off[p.dst_first + i + 1] = off + v;
        if (threadIdx.x == 1023) carry_s = off + v;
        __syncthreads();
    }
    if (threadIdx.x == 0) *p.total = carry_s - start;
}

// The 16-bit stage-1 stream of records [first, first + count) (see sparse_scan16_kernel): the record's upper half -- row flag, column,
// band; a filler's band field set to 3 -- XOR 3, and a flagged filler behind the last record.
__global__ __launch_bounds__(256) void sparse_px16_kernel(const uint32_t *px, uint16_t *px16, uint64_t first, uint64_t count) {
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x; k <= count; k += stride) {
        uint32_t h = (uint32_t)kRecRowFlag >> 16 | 3u;   // the sentinel behind the last record
        if (k < count) {
            const uint32_t v = px[first + k];
            h = v >> 16;
            if ((v & 0xffffu) == 0u) h = (h & 0xffc0u) | 3u;   // a filler: max channel 0
        }
        px16[first + k] = (uint16_t)(h ^ 3u);
    }
}

// Consistency of imported records (a shard file is untrusted input).  One thread per (image, strip, row) cell: the cell
// holds at least one record; its first record -- and no other -- carries the row flag; a cell is either ONE filler record
// or signal records with maximum > second channel, the hue band that those two channels and a class 1..6 give on this device,
// and x strictly ascending inside the strip (and the image).  The scan derives rows from the flags and plane offsets from x: a record that breaks these rules would be read as
// another pixel, so such a shard is refused instead of giving silently wrong scores.
struct ValidateParams {
    const uint32_t *px;
    const uint64_t *img_off;
    const uint32_t *rowptr;
    int64_t first, count;
    int W, H, nstrips, strip_w;
    unsigned long long *err;   // [0] bad cells, [1] index of one bad image + 1
};
__global__ __launch_bounds__(256) void sparse_validate_kernel(ValidateParams p) {
    const int64_t cell = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t per_img = (int64_t)p.nstrips * p.H;
    if (cell >= p.count * per_img) return;
    const int64_t img = cell / per_img;
    const int k = (int)((cell - img * per_img) / p.H), row = (int)(cell - img * per_img - (int64_t)k * p.H);
    const uint32_t *rp = p.rowptr + ((size_t)(p.first + img) * p.nstrips + k) * (p.H + 1);
    const uint32_t a = rp[row], b = rp[row + 1];
    const uint64_t base = p.img_off[p.first + img], n_img = p.img_off[p.first + img + 1] - base;
    bool ok = b > a && (uint64_t)b <= n_img;
    if (ok) {
        const uint32_t *r = p.px + base;
        const int xlo = k * p.strip_w, xhi = min(p.W, (k + 1) * p.strip_w);
        if (b - a == 1 && r[a] == kRecRowFlag) {
            // a filler: the cell has no signal pixel
        } else {
            int prev = xlo - 1;
            for (uint32_t i = a; i < b && ok; i++) {
                const uint32_t v = r[i];
                const int x = xlo + (int)((v >> kRecXShift) & kRecXMask);
                const uint32_t band = (v >> kRecBandShift) & kRecBandMask, cls = (band * 13u) >> 6, mx = v & 255u, md = (v >> 8) & 255u;
                const float hf = (float)md * __builtin_amdgcn_rcpf((float)mx);
                ok = ((v >> 31) == (i == a ? 1u : 0u)) && band >= 5u && band <= 34u && mx > md && band == cls * 5u + (uint32_t)(hf * 5.0f) &&
                     x > prev && x < xhi;
                prev = x;
            }
        }
    }
    if (!ok) {
        atomicAdd(p.err, 1ull);
        atomicMax(p.err + 1, (unsigned long long)(img + 1));
    }
}

// =================================================================================================
//                                       the sparse scan
// =================================================================================================
// P8: the plane holds one byte per pixel (a mask with at most 255 distinct interval records) instead of two.
// WPE: waves per SIMD the register budget is sized for -- 8 (64 VGPRs) when two workgroups share a CU, 4 (128 VGPRs) when a
// workgroup's LDS image takes more than half the CU: the latter keeps all nine interval records of a round and the bitmap
// words of four pixel blocks in flight (a wave's time is its chain of LDS round trips, not instruction issue).
#ifndef SP_BAND
#define SP_BAND 1  // 1: stage 1 tests a (4 x 8-pixel cell) x (30 hue band) map; 0: the plain near bitmap (A/B builds)
#endif
#ifndef SP_NB
#define SP_NB 4
#endif
#ifndef SM_TMAP_BINROWS
#define SM_TMAP_BINROWS 1  // the multi-mask kernel's fine map as [cell row][hue bin][cell column] (1) or [cell][hue bin] (0, round 3)
#endif
#ifndef SP16_EXP
#define SP16_EXP 0  // (timing experiments of sparse_scan16_kernel only: 1 = no rounds, 2 = rounds without the record gather, 3 = nothing queued)
#endif
#ifndef SP_EXP
#define SP_EXP 0  // (timing experiments only: 1 = no evaluation rounds, 2 = nothing queued, 3 = no interval reads, 4 = no plane reads,
                  // 5 = interval reads on a quarter of the lanes, 6 = record loads only; the scores are wrong in all of them)
#endif
// LO ("loads only", cds_debug_sparse_load_only): the kernel with its record loads, headers and work distribution and no examination
// of the records -- the time the memory system needs for this access pattern, measured beside the real scan by bench.py.
template <bool XY2, bool MIRROR, bool IVLDS, bool P8, int WPE, bool LO = false>
__global__ __launch_bounds__(1024, WPE) void sparse_scan_kernel(SparseScanParams p) {
    extern __shared__ uint4 lds4[];
    constexpr int NS = XY2 ? 9 : 1, NSX = XY2 ? 3 : 1, NO = MIRROR ? 2 : 1, NC = NS * NO;
    constexpr int HALO = XY2 ? 2 : 0;
    constexpr int NB = SP_NB;  // record blocks (64 records each) in flight per wave (4).  (8 with the 128-VGPR budget: no faster, measured
                           // twice; requests two steps ahead need three statically named register sets, i.e. the loop body three
                           // times -- a rotation by copies makes hipcc wait for the loads it copies -- for the same <= 8 %.)
    typedef typename std::conditional<P8, uint8_t, uint16_t>::type plane_t;
    uint8_t *lds = reinterpret_cast<uint8_t *>(lds4);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    uint4 *ivt = lds4;
    uint8_t *slab = lds + (IVLDS ? (size_t)p.n_iv * 16 : 0);
    const plane_t *plane = reinterpret_cast<const plane_t *>(slab);
    const uint32_t *nearb = reinterpret_cast<const uint32_t *>(slab + p.near_off);
    uint8_t *qb = slab + p.slab_bytes + (size_t)wave * SP_QBYTES;
    uint2 *q0 = reinterpret_cast<uint2 *>(qb), *q1 = q0 + SP_QCAP;
    if (IVLDS)
        for (int i = threadIdx.x; i < p.n_iv; i += blockDim.x) ivt[i] = p.iv[i];
    const int safe_off = SP_PAD * p.pitch + SP_PAD;  // an in-range plane offset for idle lanes

    for (int si = 0; si < p.n_slabs; si++) {
        const int s = (int)((blockIdx.x + (unsigned)si) % (unsigned)p.n_slabs);  // every workgroup visits every slab
        __syncthreads();  // the previous slab's waves are done with the LDS image
        {
            const uint4 *src = reinterpret_cast<const uint4 *>(p.slabs + (size_t)s * p.slab_bytes);
            uint4 *dst = reinterpret_cast<uint4 *>(slab);
            for (int i = threadIdx.x; i < p.slab_bytes / 16; i += blockDim.x) dst[i] = src[i];
        }
        __syncthreads();
        const int ys0 = p.ylo + s * p.slab_rows;  // first mask row of the slab
        const int rows_s = min(p.slab_rows, p.nrows - s * p.slab_rows);
        const int ra = max(ys0 - HALO, 0), rb = min(ys0 + rows_s + HALO, p.H);  // target rows [ra, rb)
        const int ry0 = ys0 - 2;
        const uint32_t cx0 = (uint32_t)(p.xlo - 2), cm0 = (uint32_t)(p.W - 1) - cx0;
#if SP_BAND
        const uint32_t wnear = (uint32_t)((p.cols + 4 + 7) & ~7);  // first column of the band map's guard cell (always 0): the clamp target
#else
        const uint32_t wnear = (uint32_t)(p.cols + 4);  // bit `wnear` of every bitmap row exists and is 0: the clamp target
#endif
        const uint32_t pitch = (uint32_t)p.pitch;
        // The per-pixel constants live in VGPRs (uniform values, pinned by the asm so that hipcc does not move them back):
        // the kernel's scalar registers are needed for the 18 counters, the queue state and the loop state, and every
        // scalar spilled to a VGPR lane comes back as a v_readlane inside the 64-pixel loop.
        uint32_t v_thr, v_wnear, v_nwpr, v_pitch, poff;
#if SP_BAND
        uint32_t near_lds;  // LDS address of the band map
        asm volatile("v_mov_b32 %0, %1" : "=v"(near_lds) : "s"((uint32_t)(uintptr_t)(__attribute__((address_space(3))) const uint32_t *)nearb));
#endif
        asm volatile("v_mov_b32 %0, %1" : "=v"(v_thr) : "s"(p.thr));
        asm volatile("v_mov_b32 %0, %1" : "=v"(v_wnear) : "s"(wnear));
#if SP_BAND
        asm volatile("v_mov_b32 %0, %1" : "=v"(v_nwpr) : "s"(p.nwpr * 4));  // bytes per cell row of the band map
#else
        asm volatile("v_mov_b32 %0, %1" : "=v"(v_nwpr) : "s"(p.nwpr));
#endif
        asm volatile("v_mov_b32 %0, %1" : "=v"(v_pitch) : "s"(p.pitch));
        asm volatile("v_mov_b32 %0, %1" : "=v"(poff) : "s"(2 * p.pitch + 2));
        // column strips of the library that the mask's plain window [xlo - 2, xhi + 2] and its mirror image can touch
        int wlo = p.xlo - HALO, whi = p.xlo + p.cols - 1 + HALO;
        if (MIRROR) {
            const int mlo = p.W - 1 - whi, mhi = p.W - 1 - wlo;
            wlo = min(wlo, mlo), whi = max(whi, mhi);
        }
        const int ks0 = max(wlo, 0) / p.strip_w, nsv = min(whi, p.W - 1) / p.strip_w - ks0 + 1;

        // The targets are dealt in `nparts` contiguous partitions, each with its own counter (an atomic on ONE address
        // retires every ~8 ns chip-wide: 8 K waves drawing small grabs from a single counter would wait for each other);
        // a workgroup starts at its own partition and moves on round-robin when one is exhausted.
        for (int pi = 0; pi < p.nparts; pi++) {
            const int part = (int)((blockIdx.x + (unsigned)pi) % (unsigned)p.nparts);
            const uint32_t pbeg = (uint32_t)part * p.psize, pend = min(pbeg + p.psize, (uint32_t)p.count);  // count < 2^31
            if (pbeg >= pend) continue;
            uint32_t *ctr = p.work + ((size_t)s * p.nparts + part) * 32;
            // (a partition other workgroups have already drained is recognised by a plain read: no atomic round trip)
            if (pi > 0 && __builtin_nontemporal_load(ctr) >= pend - pbeg) continue;
            for (;;) {
                uint32_t tb = 0;
                if (lane == 0) tb = atomicAdd(ctr, (uint32_t)p.group);
                tb = (uint32_t)__builtin_amdgcn_readfirstlane((int)tb);
                if (tb >= pend - pbeg) break;
                const uint32_t t0 = pbeg + tb;
                const int ng = (int)min((uint32_t)p.group, pend - t0);  // targets of this grab
                // A grab is a sequence of RUNS: (target g, visited strip k) -> one contiguous record range.  The headers of
                // all runs come in one round trip: lane r holds run r (g = r / nsv, strip ks0 + r % nsv); ng * nsv <= 64.
                const int nruns = ng * nsv;
                uint64_t h_off = 0;
                uint32_t h_a = 0, h_b = 0;
                if (lane < nruns) {
                    const int g = lane / nsv, k = ks0 + lane - g * nsv;
                    const int64_t ti = p.first + (int64_t)(t0 + (uint32_t)g);
                    const uint32_t *rp = p.rowptr + ((size_t)ti * p.nstrips + k) * (p.H + 1);
                    h_off = p.img_off[ti];
                    h_a = rp[ra];
                    h_b = rp[rb];
                }
                auto header = [&](int r, const uint32_t *&P, uint32_t &a, uint32_t &b) {
                    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)h_off, r);
                    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(h_off >> 32), r);
                    P = p.px + (((uint64_t)hi << 32) | lo);
                    a = (uint32_t)__builtin_amdgcn_readlane((int)h_a, r);
                    b = (uint32_t)__builtin_amdgcn_readlane((int)h_b, r);
                };
                // NB unconditional loads through a buffer descriptor of the image's records [0, b): a lane beyond the run reads
                // 0 (the hardware's range check) -- no flag, below every threshold -- so neither the index nor the record
                // needs a per-lane clamp, and the four blocks share one offset register (immediates 0 / 256 / 512 / 768).
                // (A conditional load would make hipcc wait with vmcnt(0) at the first use of the CURRENT blocks, i.e. for the
                // prefetch it has just issued.)
                auto load_blocks = [&](const uint32_t *P, uint32_t i, uint32_t b, uint32_t (&r)[NB]) {
                    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(P), 0, (int)(b * 4u), 0x00020000);
#if SP_EXP == 7  // (timing only, with SP_EXP 6's loop body: the same bytes as one 16-byte load per lane)
                    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)((i + 4u * (uint32_t)lane) * 4u), 0, 0);
                    r[0] = v.x, r[1] = v.y, r[2] = v.z, r[3] = v.w;
#else
                    const uint32_t vo = (i + (uint32_t)lane) * 4u;
#pragma unroll
                    for (int j = 0; j < NB; j++) r[j] = __builtin_amdgcn_raw_buffer_load_b32(rs, (int)(vo + 256u * j), 0, 0);
#endif
                };
                const uint32_t *P;
                uint32_t a, b;
                header(0, P, a, b);
#if SP_EXP == 8  // (timing only, with the loads-only body: HALF the bytes of every run -- the floor of a 2-byte stage-1 stream)
                a >>= 1, b = (b + 1) >> 1;
#endif
                uint32_t cur[NB], nxt[NB];
                load_blocks(P, a, b, cur);
                int c[NC];
#pragma unroll
                for (int k = 0; k < NC; k++) c[k] = 0;
                int qh0 = 0, qt0 = 0, qh1 = 0, qt1 = 0;  // queue heads / tails (monotonic, wave-uniform)
                int ql0 = 64, ql1 = 64;                  // head + 64: a full round is due when the tail reaches it
                int kin = 0;                             // strip of the current run, relative to ks0

                // evaluate up to 64 queued pixels of one orientation: every lane busy
                auto round = [&](const uint2 *q, int &qh, int &ql, const int qt, int *cc) {
                    const int n = min(qt - qh, 64);
#if SP_EXP == 1
                    qh += n; ql = qh + 64; cc[0] += n; return;
#endif
                    __builtin_amdgcn_wave_barrier();
                    const bool valid = lane < n;
                    const uint2 ent = q[(qh + lane) & (SP_QCAP - 1)];
                    const uint32_t r = ent.x;
                    const int off = valid ? (int)ent.y : safe_off;
                    const float fn = (float)((r >> 8) & 255u), fd = (float)(r & 255u);
                    uint32_t cls = __umul24((r >> kRecBandShift) & kRecBandMask, 13u) >> 6;  // band / 5 for bands 5..34 (kept apart from the << 29:
                    asm volatile("" : "+v"(cls));                                                // folded, it becomes a quarter-rate 32-bit multiply)
                    uint32_t code = (cls << 29) | (__float_as_uint(fn * __builtin_amdgcn_rcpf(fd)) >> 1);
                    code = valid ? code : 0u;  // 0 lies in no interval
                    const plane_t *c0 = plane + off - (XY2 ? 2 : 0);  // (column offsets 0, 2, 4 below)
                    uint32_t idx[NS];
#pragma unroll
                    for (int iy = 0; iy < NSX; iy++) {
                        const plane_t *rowp = XY2 ? c0 + (1 - iy) * 2 * (int)pitch : c0;  // mask row = target row - dy
#pragma unroll
                        for (int ix = 0; ix < NSX; ix++) {
#if SP_EXP == 4  // (timing only: no plane reads)
                            idx[ix * NSX + iy] = (uint32_t)(off * 7 + (ix * NSX + iy) * 13) & 127u;
#else
                            idx[ix * NSX + iy] = rowp[XY2 ? (2 - ix) * 2 : 0];  // mask column = target column - dx
#endif
                        }
                        // (one unaligned ds_read_b64 per kernel row instead of three byte reads -- gfx950 serves it correctly,
                        // tools/valubench/ldsalign.hip -- was measured 1.5x SLOWER overall: 5.5 vs 3.6 ms per 524,288 targets)
                    }
                    constexpr int GS = WPE <= 4 ? 9 : 3;  // interval records in flight together
#pragma unroll
                    for (int k0 = 0; k0 < NS; k0 += GS) {
                        uint4 iv[GS];
#pragma unroll
                        for (int k = 0; k < GS && k0 + k < NS; k++) {
#if SP_EXP == 3  // (timing only: no interval reads)
                            iv[k] = make_uint4(idx[k0 + k], idx[k0 + k], idx[k0 + k] * 3u, idx[k0 + k]);
#elif SP_EXP == 5  // (timing only: a quarter of the lanes read their interval record)
                            iv[k] = make_uint4(1u, 0u, 1u, 0u);
                            if ((idx[k0 + k] & 3u) == 0u) iv[k] = ivt[idx[k0 + k]];
#else
                            iv[k] = IVLDS ? ivt[idx[k0 + k]] : p.iv[idx[k0 + k]];
#endif
                        }
#pragma unroll
                        for (int k = 0; k < GS && k0 + k < NS; k++) {
                            // the OR is taken on the wave masks (s_or_b64), not on per-lane booleans.  (Counting in per-lane VGPR
                            // counters -- v_addc of the compare, one DPP sum per target -- was measured slower: 138 vs 146 M
                            // comparisons/s; the vector pipe is the tighter one.)
                            const unsigned long long mb = __ballot((code - iv[k].x) <= iv[k].y) | __ballot((code - iv[k].z) <= iv[k].w);
                            cc[k0 + k] += __popcll(mb);
                        }
                    }
                    qh += n;
                    ql = qh + 64;
                };
                auto append = [&](uint2 *q, int &qt, const unsigned long long hb, const bool hit, const uint32_t r, const uint32_t off) {
#if SP_EXP == 2
                    c[0] += __popcll(hb); return;
#endif
                    if (hit) {
                        const int slot = (qt + __builtin_amdgcn_mbcnt_hi((uint32_t)(hb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hb, 0))) & (SP_QCAP - 1);
                        q[slot] = make_uint2(r, off);
                    }
                    qt += __popcll(hb);
                };

                // end of a target: the rest of its queues, then its score (or this slab's per-transform counts)
                auto finish_target = [&](const uint32_t t) {
                    while (qt0 > qh0) round(q0, qh0, ql0, qt0, c);
                    if (MIRROR)
                        while (qt1 > qh1) round(q1, qh1, ql1, qt1, c + NS);
                    if (p.partial) {  // per-transform counts of this slab: lane k holds transform k (the 18-entry layout of the ABI)
                        int v = 0;
#pragma unroll
                        for (int k = 0; k < NS; k++) {
                            asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(v) : "s"(__builtin_amdgcn_readfirstlane(c[k])), "n"(k));
                            if (MIRROR) asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(v) : "s"(__builtin_amdgcn_readfirstlane(c[NS + k])), "n"(9 + k));
                        }
                        if (lane < CDS_N_TRANSFORMS) p.partial[((size_t)s * p.count + t) * CDS_N_TRANSFORMS + lane] = v;
                    } else {
                        int best = 0, bm = 0;
#pragma unroll
                        for (int k = 0; k < NS; k++) best = max(best, c[k]);
                        if (MIRROR) {
#pragma unroll
                            for (int k = 0; k < NS; k++) bm = max(bm, c[NS + k]);
                        }
                        int32_t packed = best;
                        if (MIRROR && bm > best) packed = (int32_t)((uint32_t)bm | 0x80000000u);
                        if (lane == 0) p.out[t] = packed;
                    }
#pragma unroll
                    for (int k = 0; k < NC; k++) c[k] = 0;
                    qh0 = qt0 = qh1 = qt1 = 0;
                    ql0 = ql1 = 64;
                };
                // (Running a target's last rounds inside its last step, BEFORE that step waits for the blocks it has requested -- so that
                // the rounds' chain of LDS round trips has a request of this wave in flight behind it -- was measured 3 % slower.)
                for (int r = 0; r < nruns; r++) {  // wave-uniform
                    int yrel = ra - 1 - ry0;  // (rows started before the current block) - 1, relative to the band map
                    // a record's x is relative to its strip: the strip's first column goes into the two column origins of the run
                    uint32_t v_cx0, v_cm0;
                    {
                        const uint32_t sbase = (uint32_t)(ks0 + kin) * (uint32_t)p.strip_w;
                        asm volatile("v_mov_b32 %0, %1" : "=v"(v_cx0) : "s"(cx0 - sbase));
                        asm volatile("v_mov_b32 %0, %1" : "=v"(v_cm0) : "s"(cm0 - sbase));
                    }
                    for (uint32_t i = a; i < b; i += 64 * NB) {  // wave-uniform
                        const uint32_t rem = b - i;  // records of this run from block 0 of this iteration on
                        // the next NB blocks -- of this run, or the first ones of the grab's next run -- are in flight while
                        // these are examined (always NB loads, so that the wait for the current blocks is a counted one:
                        // the last iteration of the grab's last run re-reads its own blocks)
                        {
                            const uint32_t *Pq = P;
                            uint32_t iq = i, bq = b;
                            if (rem > 64 * NB) iq = i + 64 * NB;
                            else if (r + 1 < nruns) {
                                header(r + 1, Pq, iq, bq);  // (read from the header lanes only now: no scalars held)
#if SP_EXP == 8
                                iq >>= 1, bq = (bq + 1) >> 1;
#endif
                            }
                            load_blocks(Pq, iq, bq, nxt);
                        }
                        // phase 1 of a block: row, columns and the two near-bitmap words of its pixels; a column outside the
                        // bitmap is clamped to the guard bit (always 0) instead of being branched around.  phase 2: queue the hits,
                        // evaluate a full queue.  With the 128-VGPR budget (WPE 4) phase 1 of all NB blocks comes first, so
                        // the 2 * NB bitmap words are requested together; with 64 VGPRs the blocks go one at a time.
                        uint32_t rec[NB], cxp[NB], cxm[NB], wp[NB], wm[NB], prow[NB];
#if SP_BAND
                        uint32_t band[NB];
#else
                        uint32_t passbit[NB];
#endif
                        auto phase1 = [&](int j) {
                            rec[j] = cur[j];  // (lanes and whole blocks beyond the run were loaded as 0: no flags, no hits)
                            const unsigned long long fb = __ballot((int)rec[j] < 0);
                            const uint32_t ry = (uint32_t)(yrel + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(fb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)fb, 0))) + (rec[j] >> 31);
                            yrel += __popcll(fb);
                            const uint32_t x = (rec[j] >> kRecXShift) & kRecXMask;  // (relative to the strip, see v_cx0 / v_cm0)
#if !SP_BAND
                            passbit[j] = (rec[j] & 255u) > v_thr ? 1u : 0u;
#endif
                            cxp[j] = min(x - v_cx0, v_wnear);
                            cxm[j] = MIRROR ? min(v_cm0 - x, v_wnear) : 0u;
#if SP_BAND
                            // hue band of the pixel (class * 5 + floor(5 * ratio), 5..34: it comes with the record; v_bfe takes it mod 32);
                            // the map cell = 4 rows x 8 columns
                            band[j] = (rec[j] >> kRecBandShift) & kRecBandMask;
                            band[j] = (rec[j] & 255u) > v_thr ? band[j] : 3u;  // below the threshold (a filler too): bit 3, which no band maps to
                            // LDS address of the cell's word = map + (ry / 4) * row bytes + (column / 8) * 4: one mad for the row, one
                            // shift-add per orientation (written as asm: hipcc turns (c >> 3) << 2 into shift, mask and a separate add)
                            typedef __attribute__((address_space(3))) const uint32_t lds_u32;
                            uint32_t rowa, ap, am;
                            asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(rowa) : "v"(ry >> 2), "v"(v_nwpr), "v"(near_lds));
                            asm("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(ap) : "v"(cxp[j] >> 3), "v"(rowa));
                            wp[j] = *reinterpret_cast<lds_u32 *>((uintptr_t)ap);
                            if (MIRROR) {
                                asm("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(am) : "v"(cxm[j] >> 3), "v"(rowa));
                                wm[j] = *reinterpret_cast<lds_u32 *>((uintptr_t)am);
                            } else {
                                wm[j] = 0u;
                            }
#else
                            const uint32_t rowi = __umul24(ry, v_nwpr);
                            wp[j] = nearb[rowi + (cxp[j] >> 5)];
                            wm[j] = MIRROR ? nearb[rowi + (cxm[j] >> 5)] : 0u;
#endif
                            // plane offset of column cx0 in the pixel's row: (ry + 2) * pitch + 2
                            asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(prow[j]) : "v"(ry), "v"(v_pitch), "v"(poff));
                        };
                        auto phase2 = [&](int j) {
                            {
#if SP_BAND
                                const bool hit = __builtin_amdgcn_ubfe(wp[j], band[j], 1u) != 0;  // (one value for the ballot and the branch)
#else
                                const bool hit = ((wp[j] >> (cxp[j] & 31u)) & passbit[j]) != 0;  // (one value for the ballot and the branch)
#endif
                                const unsigned long long hb = __ballot(hit);
                                if (hb) {  // (wave-uniform: a block without a hit -- a third of them on sparse targets -- costs one scalar
                                           // branch instead of the exec-mask region, the tail update and the queue test)
                                    append(q0, qt0, hb, hit, rec[j], prow[j] + cxp[j]);
                                    if (qt0 >= ql0) round(q0, qh0, ql0, qt0, c);
                                }
                            }
                            if (MIRROR) {
#if SP_BAND
                                const bool hit = __builtin_amdgcn_ubfe(wm[j], band[j], 1u) != 0;
#else
                                const bool hit = ((wm[j] >> (cxm[j] & 31u)) & passbit[j]) != 0;
#endif
                                const unsigned long long hb = __ballot(hit);
                                if (hb) {
                                    append(q1, qt1, hb, hit, rec[j], prow[j] + cxm[j]);
                                    if (qt1 >= ql1) round(q1, qh1, ql1, qt1, c + NS);
                                }
                            }
                        };
                        if (LO || SP_EXP == 6 || SP_EXP == 7 || SP_EXP == 8) {  // (timing only: the record loads and nothing else)
                            uint32_t acc = 0;
#pragma unroll
                            for (int j = 0; j < NB; j++) acc ^= cur[j];
                            c[0] += __popcll(__ballot(acc == 0x12345u));
                        } else
                        if (WPE <= 4) {
#pragma unroll
                            for (int j = 0; j < NB; j++) phase1(j);
#pragma unroll
                            for (int j = 0; j < NB; j++) phase2(j);  // (a block beyond the run is all zero: no hit, nothing queued -- cheaper
                                                                     // than a scalar compare and branch per block)
                        } else {
#pragma unroll
                            for (int j = 0; j < NB; j++) {
                                if (rem <= 64u * j) break;  // wave-uniform
                                phase1(j);
                                phase2(j);
                            }
                        }
#pragma unroll
                        for (int j = 0; j < NB; j++) cur[j] = nxt[j];
                    }
                    if (r + 1 < nruns) {
                        header(r + 1, P, a, b);
#if SP_EXP == 8
                        a >>= 1, b = (b + 1) >> 1;
#endif
                    }
                    if (++kin < nsv) continue;  // more strips of this target
                    kin = 0;
                    finish_target(t0 + (uint32_t)(r / nsv));
                }
            }
        }
    }
}

// =================================================================================================
//             the sparse scan over the 16-BIT STAGE-1 STREAM (round 4) -- the benchmark's `value`
// =================================================================================================
// Round 3 left the scan above waiting for its records: its loads-only twin took 0.85 of its time, at 0.66 of what a plain read of
// the same bytes reaches.  Of a record's 32 bits stage 1 reads 16 -- row flag, strip-relative column, hue band -- and on a library
// whose colours are unrelated to the mask's 98.5 % of the records never get past stage 1.  A sparse library therefore keeps those
// upper halves a second time as a stream of their own (px16: 2 bytes per record, same record indices, same row pointers; +50 %
// of record memory) and this kernel streams THAT: half the bytes per visited record, 128 records per load instruction and lane
// pair.  A record that passes the band map is queued by its INDEX; an evaluation round fetches the 64 queued records themselves
// (4 bytes each, one gather through the image's buffer descriptor) and applies the data threshold there -- stage 1 no longer sees
// the maximum channel, so it passes the few signal pixels at or below the threshold on to a round, which drops them.
// Everything else -- work distribution, run headers, band map, rounds, counters, slabs -- is the kernel above.
// Stream encoding: px16[i] = (px[i] >> 16, a filler's band field set to 3) ^ 3, so that the zero a load beyond the run's end returns
// (hardware range check) decodes to "no flag, band 3" = never a hit.  A run whose rest is odd reads one record past its end (the
// dword that holds its last record): the block in which a run ends replaces that record by the same neutral value (a wave-uniform
// branch, once per run); px16[n] behind the library's last record is a filler, so that this read never sees unwritten memory.
typedef uint16_t u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_sub_min_u16(uint32_t a, uint32_t b, uint32_t cap) {   // min(a - b, cap) per 16-bit half (wrapping subtract)
    const u16x2 d = __builtin_bit_cast(u16x2, a) - __builtin_bit_cast(u16x2, b);
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(d, __builtin_bit_cast(u16x2, cap)));
}
__device__ __forceinline__ uint32_t pk_shr2_mad_u16(uint32_t a, uint32_t b, uint32_t c) {   // (a >> 2) * b + c per 16-bit half
    const u16x2 d = (__builtin_bit_cast(u16x2, a) >> (uint16_t)2) * __builtin_bit_cast(u16x2, b) + __builtin_bit_cast(u16x2, c);
    return __builtin_bit_cast(uint32_t, d);
}

// PG ("plane global"): the mask's plane stays in global memory (L2-resident: every workgroup reads the same ~100 KB) and only the band map,
// the interval table and the queues live in LDS -- a third of the LDS, so two workgroups share a CU (8 waves per SIMD instead of 4).  Rounds
// read the plane from L2 then; right when rounds are rare (the pilot's low-hit-rate case).
template <bool XY2, bool MIRROR, bool IVLDS, bool P8, int WPE, bool LO = false, bool PG = false>
__global__ __launch_bounds__(1024, WPE) void sparse_scan16_kernel(SparseScanParams p) {
    extern __shared__ uint4 lds4[];
    constexpr int NS = XY2 ? 9 : 1, NSX = XY2 ? 3 : 1, NO = MIRROR ? 2 : 1, NC = NS * NO;
    constexpr int HALO = XY2 ? 2 : 0;
#ifndef SP16_NB
#define SP16_NB 4
#endif
    constexpr int NB = SP16_NB;      // dword loads (128 records each) in flight per wave
    typedef typename std::conditional<P8, uint8_t, uint16_t>::type plane_t;
    uint8_t *lds = reinterpret_cast<uint8_t *>(lds4);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // LDS: [two-orientation map (8-byte cells)] [interval table] [plane, unless PG] [queues].  The map comes FIRST: its cell addresses are then
    // the 16-bit offsets themselves -- the dynamic LDS of a kernel without static LDS starts at address 0, which launch_sparse_one checks
    // (hipFuncGetAttributes: sharedSizeBytes == 0) before it launches this kernel
    uint4 *ivt = reinterpret_cast<uint4 *>(lds + p.map64_bytes);
    uint8_t *slab = lds + p.map64_bytes + (IVLDS ? (size_t)p.n_iv * 16 : 0);
    const plane_t *plane = reinterpret_cast<const plane_t *>(slab);   // (PG: set per slab, below)
    uint8_t *qb = slab + (PG ? 0 : p.near_off) + (size_t)wave * SP_QBYTES;
    uint2 *q0 = reinterpret_cast<uint2 *>(qb), *q1 = q0 + SP_QCAP;
    if (IVLDS)
        for (int i = threadIdx.x; i < p.n_iv; i += blockDim.x) ivt[i] = p.iv[i];
    const int safe_off = SP_PAD * p.pitch + SP_PAD;  // an in-range plane offset for idle lanes

    for (int si = 0; si < p.n_slabs; si++) {
        const int s = (int)((blockIdx.x + (unsigned)si) % (unsigned)p.n_slabs);  // every workgroup visits every slab
        __syncthreads();  // the previous slab's waves are done with the LDS image
        if (!PG) {   // the plane
            const uint4 *src = reinterpret_cast<const uint4 *>(p.slabs + (size_t)s * p.slab_bytes);
            uint4 *dst = reinterpret_cast<uint4 *>(slab);
            for (int i = threadIdx.x; i < p.near_off / 16; i += blockDim.x) dst[i] = src[i];
        }
        {   // this slab's two-orientation map: the images follow the mask's n_slabs slab images
            const uint4 *src = reinterpret_cast<const uint4 *>(p.slabs + (size_t)p.n_slabs * p.slab_bytes + (size_t)s * p.map64_bytes);
            for (int i = threadIdx.x; i < p.map64_bytes / 16; i += blockDim.x) lds4[i] = src[i];
        }
        if (PG) plane = reinterpret_cast<const plane_t *>(p.slabs + (size_t)s * p.slab_bytes);
        __syncthreads();
        const int ys0 = p.ylo + s * p.slab_rows;  // first mask row of the slab
        const int rows_s = min(p.slab_rows, p.nrows - s * p.slab_rows);
        const int ra = max(ys0 - HALO, 0), rb = min(ys0 + rows_s + HALO, p.H);  // target rows [ra, rb)
        const int ry0 = ys0 - 2;
        const uint32_t cx0 = (uint32_t)(p.xlo - 2), cm0 = (uint32_t)(p.W - 1) - cx0;
        const uint32_t ux0 = (uint32_t)p.ux0;   // map columns: u = x - ux0, clamped to g64 = the first column of the guard cell (always 0)
        // per-pixel constants in VGPRs (uniform values, pinned by the asm: the scalar registers hold the 18 counters and the loop state)
        uint32_t v_thr, v_wnearpk, v_rowbpk, v_pitch, poff, v_colmax, v_kp, v_km, v_k10000, v_k10001;
        asm volatile("v_mov_b32 %0, 0x10000" : "=v"(v_k10000));
        asm volatile("v_mov_b32 %0, 0x10001" : "=v"(v_k10001));
        asm volatile("v_mov_b32 %0, %1" : "=v"(v_thr) : "s"(p.thr));
        asm volatile("v_mov_b32 %0, %1" : "=v"(v_wnearpk) : "s"((uint32_t)p.g64 * 0x10001u));
        asm volatile("v_mov_b32 %0, %1" : "=v"(v_rowbpk) : "s"((uint32_t)p.rowb64 * 0x10001u));  // bytes per cell row of the map (both halves)
        asm volatile("v_mov_b32 %0, %1" : "=v"(v_pitch) : "s"(p.pitch));
        asm volatile("v_mov_b32 %0, %1" : "=v"(poff) : "s"(2 * p.pitch + 2));
        // a queued pixel's window column from its map column u: plain u - (cx0 - ux0), mirrored (cm0 - ux0) - u; both within [0, cols + 3]
        // or the pixel sits in a cell shared with the window's edge and is dropped by the round
        asm volatile("v_mov_b32 %0, %1" : "=v"(v_kp) : "s"(cx0 - ux0));
        asm volatile("v_mov_b32 %0, %1" : "=v"(v_km) : "s"(cm0 - ux0));
        asm volatile("v_mov_b32 %0, %1" : "=v"(v_colmax) : "s"(p.cols + 3));
        // column strips of the library that the mask's plain window [xlo - 2, xhi + 2] and its mirror image can touch
        int wlo = p.xlo - HALO, whi = p.xlo + p.cols - 1 + HALO;
        if (MIRROR) {
            const int mlo = p.W - 1 - whi, mhi = p.W - 1 - wlo;
            wlo = min(wlo, mlo), whi = max(whi, mhi);
        }
        const int ks0 = max(wlo, 0) / p.strip_w, nsv = min(whi, p.W - 1) / p.strip_w - ks0 + 1;
        uint32_t st_visited = 0, st_queued = 0;   // (pilot launches only: p.stat)

        for (int pi = 0; pi < p.nparts; pi++) {
            const int part = (int)((blockIdx.x + (unsigned)pi) % (unsigned)p.nparts);
            const uint32_t pbeg = (uint32_t)part * p.psize, pend = min(pbeg + p.psize, (uint32_t)p.count);  // count < 2^31
            if (pbeg >= pend) continue;
            uint32_t *ctr = p.work + ((size_t)s * p.nparts + part) * 32;
            if (pi > 0 && __builtin_nontemporal_load(ctr) >= pend - pbeg) continue;
            for (;;) {
                uint32_t tb = 0;
                if (lane == 0) tb = atomicAdd(ctr, (uint32_t)p.group);
                tb = (uint32_t)__builtin_amdgcn_readfirstlane((int)tb);
                if (tb >= pend - pbeg) break;
                const uint32_t t0 = pbeg + tb;
                const int ng = (int)min((uint32_t)p.group, pend - t0);  // targets of this grab
                const int nruns = ng * nsv;                              // runs: (target, visited strip), one header lane each
                uint64_t h_off = 0;
                uint32_t h_a = 0, h_b = 0;
                if (lane < nruns) {
                    const int g = lane / nsv, k = ks0 + lane - g * nsv;
                    const int64_t ti = p.first + (int64_t)(t0 + (uint32_t)g);
                    const uint32_t *rp = p.rowptr + ((size_t)ti * p.nstrips + k) * (p.H + 1);
                    h_off = p.img_off[ti];
                    h_a = rp[ra];
                    h_b = rp[rb];
                }
                auto header = [&](int r, uint64_t &off, uint32_t &a, uint32_t &b) {
                    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)h_off, r);
                    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(h_off >> 32), r);
                    off = ((uint64_t)hi << 32) | lo;
                    a = (uint32_t)__builtin_amdgcn_readlane((int)h_a, r);
                    b = (uint32_t)__builtin_amdgcn_readlane((int)h_b, r);
                };
                // NB unconditional dword loads (two records each) through a descriptor of the image's stream [0, b) -- plus one record when
                // the rest of the run is odd, see the head of this kernel; lanes and whole blocks beyond the run read 0
                auto load_blocks = [&](uint64_t off, uint32_t i, uint32_t b, uint32_t (&r)[NB]) {
                    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(p.px16 + off), 0, (int)((b + ((b - i) & 1u)) * 2u), 0x00020000);
                    const uint32_t vo = (i + 2u * (uint32_t)lane) * 2u;
#pragma unroll
                    for (int j = 0; j < NB; j++) r[j] = __builtin_amdgcn_raw_buffer_load_b32(rs, (int)(vo + 256u * j), 0, 0);
                };
                uint64_t off;
                uint32_t a, b;
                header(0, off, a, b);
                uint32_t cur[NB], nxt[NB];
                load_blocks(off, a, b, cur);
                int c[NC];
#pragma unroll
                for (int k = 0; k < NC; k++) c[k] = 0;
                int qh0 = 0, qt0 = 0, qh1 = 0, qt1 = 0;  // queue heads / tails (monotonic over the grab, wave-uniform)
                int ql0 = 64, ql1 = 64;                  // head + 64: a full round is due when the tail reaches it
                int kin = 0;                             // strip of the current run, relative to ks0
                // Round 5: an evaluation round is cut in two.  ISSUE takes up to 64 entries out of the queue and REQUESTS their records (one
                // gather through the image's descriptor) -- nothing waits; EVAL, later, does the round's work on the records that have arrived
                // meanwhile.  Between the two the wave goes on with stage 1 of the following blocks, whose stream loads were requested before
                // the gather (vmcnt retires in order: waiting for those blocks waits for the gather too, so it costs no wait of its own).
                // A full queue evaluates the round issued at the previous trigger and issues the next; a target's LAST (partial) rounds are
                // issued at its end and evaluated one step into the next target of the grab (`tail_t`), or at the end of the grab.  Round 4
                // evaluated every round where it was issued: 0.3-0.45 ms of a 1 M-target launch went into waiting for dependent gathers.
                uint32_t pr0 = 0, py0 = 0, pr1 = 0, py1 = 0;   // the issued round of each orientation: gathered records, entries' row | column
                int pn0 = 0, pn1 = 0;                          // its entries (0: none issued)
                uint32_t tail_t = 0;                           // the target whose last rounds are issued but not evaluated ...
                bool have_tail = false;                        // ... if any

                auto issue = [&](const uint2 *q, int &qh, int &ql, const int qt, uint32_t &pr, uint32_t &py, int &pn) {
                    const int n = min(qt - qh, 64);
                    __builtin_amdgcn_wave_barrier();
                    const uint2 ent = q[(qh + lane) & (SP_QCAP - 1)];
#if SP16_EXP == 2   // (timing only: rounds without the gather of the records)
                    pr = ent.x * 2654435761u | 0x00050001u;
#else
                    const __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(p.px + off), 0, (int)0x7ffffffc, 0x00020000);
                    pr = __builtin_amdgcn_raw_buffer_load_b32(ri, lane < n ? (int)(ent.x * 4u) : 0, 0, 0);
#endif
                    py = ent.y;
                    pn = n;
                    qh += n;
                    ql = qh + 64;
                };
                // evaluate an issued round: up to 64 queued pixels of one orientation, every lane busy; the data threshold is applied here
                auto eval = [&](const uint32_t r, const uint32_t y, int &pn, int *cc, const bool mirrored) {
                    const int n = pn;
                    pn = 0;
#if SP16_EXP == 1   // (timing only: no evaluation rounds)
                    cc[0] += n; return;
#endif
                    bool valid = lane < n;
                    uint32_t po;   // plane offset of the pixel: (row + 2) * pitch + 2 + window column
                    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(po) : "v"(y >> 16), "v"(v_pitch), "v"(poff));
                    const uint32_t col = mirrored ? v_km - (y & 0xffffu) : (y & 0xffffu) - v_kp;
                    valid = valid && col <= v_colmax;   // (outside the window nothing can match: all nine probes miss the mask)
                    const int poffs = valid ? (int)(po + col) : safe_off;
                    valid = valid && (r & 255u) > v_thr;
                    const float fn = (float)((r >> 8) & 255u), fd = (float)(r & 255u);
                    uint32_t cls = __umul24((r >> kRecBandShift) & kRecBandMask, 13u) >> 6;  // band / 5 for bands 5..34 (kept apart from the << 29:
                    asm volatile("" : "+v"(cls));                                                // folded, it becomes a quarter-rate 32-bit multiply)
                    uint32_t code = (cls << 29) | (__float_as_uint(fn * __builtin_amdgcn_rcpf(fd)) >> 1);
                    code = valid ? code : 0u;  // 0 lies in no interval
                    const plane_t *c0 = plane + poffs - (XY2 ? 2 : 0);  // (column offsets 0, 2, 4 below)
                    uint32_t idx[NS];
#pragma unroll
                    for (int iy = 0; iy < NSX; iy++) {
                        const plane_t *rowp = XY2 ? c0 + (1 - iy) * 2 * (int)p.pitch : c0;  // mask row = target row - dy
                        if (PG && XY2) {
                            // the plane is in global memory: ONE unaligned load per mask row brings the three columns (elements 0, 2, 4 of a
                            // five-element span) -- a round is a chain of gathers, and 64 scattered lanes cost the texture path the same per
                            // instruction whether they ask for one byte or eight: 3 gathers instead of 9.  (The elements behind the span are the
                            // row's zero padding or the next row's, always inside the mask's slab images.)
                            uint32_t w[4];
                            __builtin_memcpy(w, rowp, P8 ? 8 : 16);
#pragma unroll
                            for (int ix = 0; ix < NSX; ix++) {
                                const int e = (2 - ix) * 2;   // mask column = target column - dx
                                idx[ix * NSX + iy] = P8 ? (w[e >> 2] >> ((e & 3) * 8)) & 255u : (w[e >> 1] >> ((e & 1) * 16)) & 0xffffu;
                            }
                        } else {
#pragma unroll
                            for (int ix = 0; ix < NSX; ix++) idx[ix * NSX + iy] = rowp[XY2 ? (2 - ix) * 2 : 0];  // mask column = target column - dx
                        }
                    }
                    constexpr int GS = WPE <= 4 ? 9 : 3;  // interval records in flight together
#pragma unroll
                    for (int k0 = 0; k0 < NS; k0 += GS) {
                        uint4 iv[GS];
#pragma unroll
                        for (int k = 0; k < GS && k0 + k < NS; k++) iv[k] = IVLDS ? ivt[idx[k0 + k]] : p.iv[idx[k0 + k]];
#pragma unroll
                        for (int k = 0; k < GS && k0 + k < NS; k++) {
                            const unsigned long long mb = __ballot((code - iv[k].x) <= iv[k].y) | __ballot((code - iv[k].z) <= iv[k].w);
                            cc[k0 + k] += __popcll(mb);
                        }
                    }
                };
                // the score of a target whose every round has been evaluated (or this slab's per-transform counts); the counters start over
                auto write_target = [&](const uint32_t t) {
                    if (p.partial) {  // per-transform counts of this slab: lane k holds transform k (the 18-entry layout of the ABI)
                        int v = 0;
#pragma unroll
                        for (int k = 0; k < NS; k++) {
                            asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(v) : "s"(__builtin_amdgcn_readfirstlane(c[k])), "n"(k));
                            if (MIRROR) asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(v) : "s"(__builtin_amdgcn_readfirstlane(c[NS + k])), "n"(9 + k));
                        }
                        if (lane < CDS_N_TRANSFORMS) p.partial[((size_t)s * p.count + t) * CDS_N_TRANSFORMS + lane] = v;
                    } else {
                        int best = 0, bm = 0;
#pragma unroll
                        for (int k = 0; k < NS; k++) best = max(best, c[k]);
                        if (MIRROR) {
#pragma unroll
                            for (int k = 0; k < NS; k++) bm = max(bm, c[NS + k]);
                        }
                        int32_t packed = best;
                        if (MIRROR && bm > best) packed = (int32_t)((uint32_t)bm | 0x80000000u);
                        if (lane == 0) p.out[t] = packed;
                    }
#pragma unroll
                    for (int k = 0; k < NC; k++) c[k] = 0;
                };
                // the previous target's issued last rounds: evaluate them, the target is complete
                auto finalize_tail = [&]() {
                    if (pn0) eval(pr0, py0, pn0, c, false);
                    if (MIRROR && pn1) eval(pr1, py1, pn1, c + NS, true);
                    write_target(tail_t);
                    have_tail = false;
                };
                // a queue holds 64 entries: the round issued before (if any) is evaluated, the next one issued.  (Entries of the CURRENT target
                // may only be counted once the previous target is complete: its tail goes first.)
                auto full0 = [&]() {
                    if (have_tail) finalize_tail();
                    if (pn0) eval(pr0, py0, pn0, c, false);
                    issue(q0, qh0, ql0, qt0, pr0, py0, pn0);
                };
                auto full1 = [&]() {
                    if (have_tail) finalize_tail();
                    if (pn1) eval(pr1, py1, pn1, c + NS, true);
                    issue(q1, qh1, ql1, qt1, pr1, py1, pn1);
                };
                auto append = [&](uint2 *q, int &qt, const unsigned long long hb, const bool hit, const uint32_t idx, const uint32_t po) {
#if SP16_EXP == 3   // (timing only: nothing queued)
                    c[0] += __popcll(hb); return;
#endif
                    if (hit) {
                        const int slot = (qt + __builtin_amdgcn_mbcnt_hi((uint32_t)(hb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hb, 0))) & (SP_QCAP - 1);
                        q[slot] = make_uint2(idx, po);
                    }
                    qt += __popcll(hb);
                };
                // end of a target's records: what its queues still hold (< 64 entries each) is issued; `tail_t` is evaluated later
                auto end_target = [&](const uint32_t t) {
                    if (have_tail) finalize_tail();   // (a target of one step: the previous tail has not been taken yet)
                    if (pn0) eval(pr0, py0, pn0, c, false);
                    if (MIRROR && pn1) eval(pr1, py1, pn1, c + NS, true);
                    if (qt0 > qh0) issue(q0, qh0, ql0, qt0, pr0, py0, pn0);
                    if (MIRROR && qt1 > qh1) issue(q1, qh1, ql1, qt1, pr1, py1, pn1);
                    tail_t = t;
                    have_tail = true;
                };
                for (int r = 0; r < nruns; r++) {  // wave-uniform
                    int yrel = ra - 1 - ry0;  // (rows started before the current block) - 1, relative to the band map
                    if (p.stat) st_visited += b - a;
                    // a record's x is relative to its strip: the strip's first column goes into the run's two column origins (both halves)
                    uint32_t v_ux0pk;
                    {
                        const uint32_t sbase = (uint32_t)(ks0 + kin) * (uint32_t)p.strip_w;
                        asm volatile("v_mov_b32 %0, %1" : "=v"(v_ux0pk) : "s"(((ux0 - sbase) & 0xffffu) * 0x10001u));
                    }
                    for (uint32_t i = a; i < b; i += 128 * NB) {  // wave-uniform
                        const uint32_t rem = b - i;  // records of this run from block 0 of this iteration on
                        {   // the next NB blocks -- of this run, or the first ones of the grab's next run -- are in flight while these are examined
                            uint64_t oq = off;
                            uint32_t iq = i, bq = b;
                            if (rem > 128 * NB) iq = i + 128 * NB;
                            else if (r + 1 < nruns) header(r + 1, oq, iq, bq);
                            load_blocks(oq, iq, bq, nxt);
                        }
                        if (LO) {  // (the loads-only twin: the record loads and nothing else)
                            uint32_t acc = 0;
#pragma unroll
                            for (int j = 0; j < NB; j++) acc ^= cur[j];
                            c[0] += __popcll(__ballot(acc == 0x12345u));
                        } else {
#pragma unroll
                            for (int j = 0; j < NB; j++) {
                                if (rem <= 128u * j) break;  // wave-uniform: a block beyond the run
                                uint32_t rec = cur[j];   // (band fields as the stream has them, XOR 3: the map's bits are laid out for that)
                                if (rem < 128u * (j + 1)) {   // wave-uniform, once per run: the run ends inside this block.  When the rest of the run
                                    // is odd, the dword holding its last record also holds the record BEHIND it (the first of the next row, strip or
                                    // image): that one is not part of the scan -- turned into 0, "no flag, bit 0 of the map", like everything beyond
                                    // the run.  (The empty asm keeps this a scalar branch: as a select it would cost every block vector instructions.)
                                    if (2u * (uint32_t)lane + 1u >= rem - 128u * j) rec &= 0xffffu;
                                    asm volatile("" : "+v"(rec));
                                }
                                // rows: records come in the order (lane 0 low, lane 0 high, lane 1 low, ...).  Row of lane L's LOW record = rows
                                // started before the block + flags of all records of lanes < L + its own flag = flags of the low records of lanes
                                // <= L + flags of the high records of lanes < L: one mbcnt chain over (low flags >> 1) and the high flags, lane 0's
                                // low flag added on the scalar side; the HIGH record's row = that + its own flag, an add-with-carry whose carry-in
                                // is the high-flag mask itself.  Both rows then live in ONE register (16-bit halves), like the columns.
                                // (gfx950 issues the two-operand encodings -- add, and, shifts, cndmask with vcc -- in ~2.5 cycles per wave and the
                                // three-operand, packed, SDWA and compare ones in ~4.2, tools/valubench/opbench: the choices below follow that)
                                unsigned long long fbl, fbh;
                                uint32_t hf16;   // 0x10000 where the high record starts a row
                                asm("v_cmp_gt_i16_e64 %0, 0, %1" : "=s"(fbl) : "v"(rec));   // sign of the low 16 bits = the low record's row flag
                                asm("v_cmp_gt_i32_e32 vcc, 0, %2\n\tv_cndmask_b32_e32 %0, 0, %3, vcc\n\ts_mov_b64 %1, vcc" : "=&v"(hf16), "=s"(fbh) : "v"(rec), "v"(v_k10000) : "vcc");
                                const unsigned long long fbl1 = fbl >> 1;
                                const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(fbh >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)fbh,
                                                       __builtin_amdgcn_mbcnt_hi((uint32_t)(fbl1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)fbl1, 0))));
                                const uint32_t ryl = below + (uint32_t)(yrel + (int)((uint32_t)fbl & 1u));
                                uint32_t rypk;   // both rows: ryl | (ryl + high flag) << 16
                                asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(rypk) : "v"(ryl), "v"(v_k10001), "v"(hf16));
                                yrel += __popcll(fbl) + __popcll(fbh);
                                // map columns of both records at once: x - ux0, clamped to the guard cell; the cell's byte offset in its row is
                                // (u / 8) * 8 = u with the low three bits cleared; the row's = (ry / 4) * row bytes: one packed multiply-add
                                const uint32_t xpk = (rec >> 6) & 0x01FF01FFu;
                                const uint32_t upk = pk_sub_min_u16(xpk, v_ux0pk, v_wnearpk);
                                const uint32_t offpk = pk_shr2_mad_u16(rypk, v_rowbpk, upk & 0xFFF8FFF8u);
                                const uint32_t bh = rec >> 16;                             // the high record alone (v_bfe takes the band mod 32 from the low bits)
                                const uint32_t idl = i + 128u * j + 2u * (uint32_t)lane;   // index of the low record in the image
                                // one 8-byte cell per record: .x = the bands the plain mask accepts around here, .y = the mirrored mask's
                                typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                                typedef __attribute__((address_space(3))) const u32x2 lds_u64;
                                const uint32_t al = offpk & 0xffffu, ah = offpk >> 16;   // (the map starts at LDS address 0: checked by the host, launch_sparse_one)
                                const u32x2 wl = *reinterpret_cast<lds_u64 *>((uintptr_t)al), wh = *reinterpret_cast<lds_u64 *>((uintptr_t)ah);
                                // (an entry carries row << 16 | map column, one v_perm; the round turns it into the plane offset: the vector pipe is
                                // the kernel's bottleneck, and two multiply-adds per block here cost more than one per round there)
                                {
                                    const bool hl = (int)(wl.x << (rec & 31u)) < 0, hh = (int)(wh.x << (bh & 31u)) < 0;   // (bit 31 - field: see the map)
                                    const unsigned long long hbl = __ballot(hl), hbh = __ballot(hh);
                                    if (hbl) {
                                        append(q0, qt0, hbl, hl, idl, __builtin_amdgcn_perm(rypk, upk, 0x05040100u));
                                        if (qt0 >= ql0) full0();
                                    }
                                    if (hbh) {
                                        append(q0, qt0, hbh, hh, idl + 1u, __builtin_amdgcn_perm(rypk, upk, 0x07060302u));
                                        if (qt0 >= ql0) full0();
                                    }
                                }
                                if (MIRROR) {
                                    const bool hl = (int)(wl.y << (rec & 31u)) < 0, hh = (int)(wh.y << (bh & 31u)) < 0;
                                    const unsigned long long hbl = __ballot(hl), hbh = __ballot(hh);
                                    if (hbl) {
                                        append(q1, qt1, hbl, hl, idl, __builtin_amdgcn_perm(rypk, upk, 0x05040100u));
                                        if (qt1 >= ql1) full1();
                                    }
                                    if (hbh) {
                                        append(q1, qt1, hbh, hh, idl + 1u, __builtin_amdgcn_perm(rypk, upk, 0x07060302u));
                                        if (qt1 >= ql1) full1();
                                    }
                                }
                            }
                        }
                        if (have_tail) finalize_tail();   // (one step into the next target: the tail's records have arrived behind this step's blocks)
#pragma unroll
                        for (int j = 0; j < NB; j++) cur[j] = nxt[j];
                    }
                    // the target's last rounds gather from ITS image: issued before the header of the next run replaces `off`
                    if (kin + 1 == nsv) end_target(t0 + (uint32_t)(r / nsv));
                    if (r + 1 < nruns) header(r + 1, off, a, b);
                    kin = kin + 1 == nsv ? 0 : kin + 1;
                }
                if (have_tail) finalize_tail();   // the grab's last target
                if (p.stat) st_queued += (uint32_t)(qt0 + qt1);
            }
        }
        if (p.stat && lane == 0 && (st_visited | st_queued)) {   // (pilot launches: this wave's share of the statistics)
            atomicAdd(p.stat, (unsigned long long)st_visited);
            atomicAdd(p.stat + 1, (unsigned long long)st_queued);
        }
    }
}

// sum of the slabs' per-transform counts -> score | mirrored << 31 (and, optionally, the 18 counts)
__global__ __launch_bounds__(256) void sparse_finalize_kernel(const int32_t *partial, int n_slabs, int64_t count, int32_t *out,
                                                              int32_t *counts) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= count) return;
    int best = 0, bm = 0;
    for (int k = 0; k < CDS_N_TRANSFORMS; k++) {
        int v = 0;
        for (int s = 0; s < n_slabs; s++) v += partial[((size_t)s * count + t) * CDS_N_TRANSFORMS + k];
        if (counts) counts[(size_t)t * CDS_N_TRANSFORMS + k] = v;
        if (k < 9) best = v > best ? v : best;
        else bm = v > bm ? v : bm;
    }
    out[t] = bm > best ? (int32_t)((uint32_t)bm | 0x80000000u) : best;
}


// =================================================================================================
//                      the multi-mask sparse scan (BASELINE configs[2]: many masks x one library)
// =================================================================================================
// The single-mask kernel above spends most of its instructions on work that does not depend on the mask: fetching a
// record, finding its row (ballot + mbcnt), its hue band, its cell.  Here a GROUP of up to 8 masks shares that work
// (reference: the masks x target-partitions loop of tools cmd/cdsprocess/LocalColorMIPSearchProcessor.java:82-116):
//   * stage 1, once per record: row, columns (plain / mirrored), band -> ONE byte per orientation from the group's
//     transposed band map (global memory, L2-resident: cells x 32 bands, bit j = mask j), i.e. the set of (mask, orientation)
//     pairs the record can matter to -- 16 bits per record;
//   * expansion: every set bit becomes an entry (record, plane offset | bit << 20) of the wave's ONE queue (a loop over
//     "lowest set bit of every lane": as many turns as the busiest lane has bits);
//   * stage 2, 64 entries at a time, every lane busy, the lanes of a round belonging to DIFFERENT masks: hue code, 9 plane
//     bytes of the entry's own mask (u8 planes of all masks in global memory / L2: they do not fit LDS together), 9 interval
//     records from the masks' tables in LDS, 2 range compares each; a match increments the wave's LDS counter of
//     (mask, transform) -- matches are rare, so the atomics are few;
//   * end of a target: lane j turns mask j's 18 counters into score | mirrored << 31 and clears them.
// A group = consecutive masks, at most 8, whose interval tables together hold at most 3,072 records (a synthetic arbor
// mask has 150-380 distinct colours, the reference's EM masks 105-389).
// LDS per workgroup: 3,072 interval records (48 KB) + per wave a queue of 128 entries and 8 x 18 counters = 74 KB: two
// workgroups per CU, 8 waves per SIMD (64 VGPRs) to cover the L2 round trips of the map and plane bytes.
constexpr int SM_QCAP = 128;
constexpr int SM_WAVE_BYTES = SM_QCAP * 8 + kSparseGroupMasks * CDS_N_TRANSFORMS * 4;
// dynamic LDS of one workgroup: interval records | the masks' table bases (64 B) | union band map | per-wave queue + counters
static size_t multi_fixed_bytes(int n_iv, int umap_words) { return (size_t)round_up(n_iv, 4) * 16 + 64 + (size_t)round_up(umap_words, 4) * 4; }
static size_t multi_lds_bytes(int threads, int n_iv, int umap_words) { return multi_fixed_bytes(n_iv, umap_words) + (size_t)(threads / 64) * SM_WAVE_BYTES; }

template <bool XY2, bool MIRROR, bool P8, bool PF>
__global__ __launch_bounds__(1024, 8) void sparse_multi_kernel(SparseMultiParams p) {
    extern __shared__ uint4 lds4[];
    constexpr int NS = XY2 ? 9 : 1;
    constexpr int HALO = XY2 ? 2 : 0;
    #ifndef SM_NB
#define SM_NB 2
#endif
    constexpr int NB = SM_NB;
    constexpr int K = kSparseGroupMasks;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // the planes and the map are read through buffer descriptors: a 32-bit offset register per access (the three columns
    // of a plane row as instruction immediates) instead of a 64-bit address pair -- all nine plane reads of a round fit the
    // 64-VGPR budget and are in flight together
    constexpr uint32_t ES = P8 ? 1u : 2u;  // bytes per plane cell
    const __amdgpu_buffer_rsrc_t planes_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.planes), 0, (int)((uint32_t)p.n_masks * (uint32_t)p.plane_cells * ES), 0x00020000);
    const __amdgpu_buffer_rsrc_t tmap_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(p.tmap), 0, (int)p.tmap_bytes, 0x00020000);
    uint4 *ivt = lds4;
    const int n_iv4 = (p.n_iv + 3) & ~3, um4 = (p.umap_words + 3) & ~3;
    int *ivb = reinterpret_cast<int *>(lds4 + n_iv4);
    uint32_t *umap = reinterpret_cast<uint32_t *>(ivb + 16);
    uint8_t *wb = reinterpret_cast<uint8_t *>(umap + um4) + (size_t)wave * SM_WAVE_BYTES;
    uint2 *q = reinterpret_cast<uint2 *>(wb);
    int *cnt = reinterpret_cast<int *>(wb + SM_QCAP * 8);
    for (int i = threadIdx.x; i < p.n_iv; i += blockDim.x) ivt[i] = p.iv[i];
    for (int i = threadIdx.x; i < p.umap_words; i += blockDim.x) umap[i] = p.umap[i];
    // (PF = false: a union map too large for the LDS is left out -- every record then asks the L2 map)
    if (threadIdx.x < K) ivb[threadIdx.x] = p.iv_base[threadIdx.x];
    for (int i = lane; i < K * CDS_N_TRANSFORMS; i += 64) cnt[i] = 0;
    __syncthreads();
    const uint32_t safe_off = (uint32_t)(SP_PAD * p.pitch + SP_PAD);  // an in-range plane offset for idle lanes
    const int ra = max(p.ylo - HALO, 0), rb = min(p.ylo + p.nrows + HALO, p.H);  // target rows [ra, rb)
    const int ry0 = p.ylo - 2;
    const uint32_t cx0 = (uint32_t)(p.xlo - 2), cm0 = (uint32_t)(p.W - 1) - cx0;
    const uint32_t wnear = (uint32_t)((p.cols + 4 + 7) & ~7);  // first column of the map's guard cell (always 0): the clamp target
    const uint32_t pitch = (uint32_t)p.pitch, nwpr = (uint32_t)p.nwpr, poff = (uint32_t)(2 * p.pitch + 2);
    const uint32_t thr = (uint32_t)p.thr, pcells = (uint32_t)p.plane_cells;
    int wlo = p.xlo - HALO, whi = p.xlo + p.cols - 1 + HALO;
    if (MIRROR) {
        const int mlo = p.W - 1 - whi, mhi = p.W - 1 - wlo;
        wlo = min(wlo, mlo), whi = max(whi, mhi);
    }
    const int ks0 = max(wlo, 0) / p.strip_w, nsv = min(whi, p.W - 1) / p.strip_w - ks0 + 1;

    for (int pi = 0; pi < p.nparts; pi++) {
        const int part = (int)((blockIdx.x + (unsigned)pi) % (unsigned)p.nparts);
        const uint32_t pbeg = (uint32_t)part * p.psize, pend = min(pbeg + p.psize, (uint32_t)p.count);
        if (pbeg >= pend) continue;
        uint32_t *ctr = p.work + (size_t)part * 32;
        if (pi > 0 && __builtin_nontemporal_load(ctr) >= pend - pbeg) continue;
        for (;;) {
            uint32_t tb = 0;
            if (lane == 0) tb = atomicAdd(ctr, (uint32_t)p.group);
            tb = (uint32_t)__builtin_amdgcn_readfirstlane((int)tb);
            if (tb >= pend - pbeg) break;
            const uint32_t t0 = pbeg + tb;
            const int ng = (int)min((uint32_t)p.group, pend - t0);
            const int nruns = ng * nsv;  // (target, visited strip) -> one contiguous record range; lane r holds run r's header
            uint64_t h_off = 0;
            uint32_t h_a = 0, h_b = 0;
            if (lane < nruns) {
                const int g = lane / nsv, k = ks0 + lane - g * nsv;
                const int64_t ti = p.first + (int64_t)(t0 + (uint32_t)g);
                const uint32_t *rp = p.rowptr + ((size_t)ti * p.nstrips + k) * (p.H + 1);
                h_off = p.img_off[ti];
                h_a = rp[ra];
                h_b = rp[rb];
            }
            auto header = [&](int r, const uint32_t *&P, uint32_t &a, uint32_t &b) {
                const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)h_off, r);
                const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(h_off >> 32), r);
                P = p.px + (((uint64_t)hi << 32) | lo);
                a = (uint32_t)__builtin_amdgcn_readlane((int)h_a, r);
                b = (uint32_t)__builtin_amdgcn_readlane((int)h_b, r);
            };
            auto load_blocks = [&](const uint32_t *P, uint32_t i, uint32_t b, uint32_t (&r)[NB]) {  // (see sparse_scan_kernel)
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(P), 0, (int)(b * 4u), 0x00020000);
                const uint32_t vo = (i + (uint32_t)lane) * 4u;
#pragma unroll
                for (int j = 0; j < NB; j++) r[j] = __builtin_amdgcn_raw_buffer_load_b32(rs, (int)(vo + 256u * j), 0, 0);
            };
            const uint32_t *P;
            uint32_t a, b;
            header(0, P, a, b);
            uint32_t cur[NB], nxt[NB];
            load_blocks(P, a, b, cur);
            int qh = 0, qt = 0;  // queue head / tail (monotonic, wave-uniform)
            int kin = 0;

            // evaluate up to 64 queued (record, mask, orientation) entries: every lane busy, lanes of different masks
            auto round = [&]() {
                const int n = min(qt - qh, 64);
#if defined(SM_EXP) && SM_EXP == 3   // (timing experiment: entries are queued, rounds do nothing)
                qh += n;
                return;
#endif
                __builtin_amdgcn_wave_barrier();
                const bool valid = lane < n;
                const uint2 ent = q[(qh + lane) & (SM_QCAP - 1)];
                const uint32_t r = ent.x;
                const uint32_t e = valid ? ent.y : safe_off;
                const uint32_t off = e & 0xFFFFFu, bit = e >> 20, m = bit & 7u, o = bit >> 3;
                const float fn = (float)((r >> 8) & 255u), fd = (float)(r & 255u);
                // band / 5 for bands 5..34 -- as a 24-bit multiply and two shifts: hipcc folds `(band * 13 >> 6) << 29` into ONE full 32-bit
                // multiply by 13 << 23, a quarter-rate instruction, and the vector pipes are this kernel's tightest resource
                uint32_t rcls = __umul24((r >> kRecBandShift) & kRecBandMask, 13u) >> 6;
                asm volatile("" : "+v"(rcls));
                uint32_t code = (rcls << 29) | (__float_as_uint(fn * __builtin_amdgcn_rcpf(fd)) >> 1);
                code = valid ? code : 0u;  // 0 lies in no interval
                // plane bytes of the entry's own mask: mask row = target row - dy, mask column = target column - dx
                const uint4 *ivm = ivt + ivb[m];  // (requested before the plane reads: the asm below is a memory barrier to hipcc)
                const uint32_t top = (__umul24(m, pcells) + off + (XY2 ? 2u * pitch - 2u : 0u)) * ES;  // byte offset of cell (row + 2, column - 2) of mask m's plane (< 2^24)
                uint32_t idx[NS];
#if defined(SM_EXP) && SM_EXP == 4   // (timing experiment: no plane gathers)
                for (int k = 0; k < NS; k++) idx[k] = (top + 13u * k) & 127u;
                if (false)
#else
                if (XY2)
#endif
                {
                    // The three cells of a plane row (columns -2, 0, +2 around the pixel = 5 consecutive cells) come with ONE
                    // unaligned load per row -- 12 bytes of u16 cells, 8 bytes of u8 cells -- instead of three narrow ones: a gather
                    // costs the texture addresser the same whatever its width, and the gathers are this kernel's tightest pipe.
                    // Issued back to back, then ONE wait (hipcc splits them when it is short of registers).
                    typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
                    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                    const uint32_t r0 = top, r1 = top - 2u * pitch * ES, r2 = top - 4u * pitch * ES;
                    if (P8) {
                        u32x2 d0, d1, d2;
                        asm volatile("buffer_load_dwordx2 %0, %3, %6, 0 offen\n\tbuffer_load_dwordx2 %1, %4, %6, 0 offen\n\t"
                                     "buffer_load_dwordx2 %2, %5, %6, 0 offen\n\ts_waitcnt vmcnt(0)"
                                     : "=&v"(d0), "=&v"(d1), "=&v"(d2) : "v"(r0), "v"(r1), "v"(r2), "s"(planes_rs) : "memory");
                        idx[0] = d0.y & 255u, idx[3] = (d0.x >> 16) & 255u, idx[6] = d0.x & 255u;   // idx[ix * 3 + iy]: column +4 - 2 ix, row iy
                        idx[1] = d1.y & 255u, idx[4] = (d1.x >> 16) & 255u, idx[7] = d1.x & 255u;
                        idx[2] = d2.y & 255u, idx[5] = (d2.x >> 16) & 255u, idx[8] = d2.x & 255u;
                    } else {
                        u32x3 d0, d1, d2;
                        asm volatile("buffer_load_dwordx3 %0, %3, %6, 0 offen\n\tbuffer_load_dwordx3 %1, %4, %6, 0 offen\n\t"
                                     "buffer_load_dwordx3 %2, %5, %6, 0 offen\n\ts_waitcnt vmcnt(0)"
                                     : "=&v"(d0), "=&v"(d1), "=&v"(d2) : "v"(r0), "v"(r1), "v"(r2), "s"(planes_rs) : "memory");
                        idx[0] = d0.z & 0xFFFFu, idx[3] = d0.y & 0xFFFFu, idx[6] = d0.x & 0xFFFFu;
                        idx[1] = d1.z & 0xFFFFu, idx[4] = d1.y & 0xFFFFu, idx[7] = d1.x & 0xFFFFu;
                        idx[2] = d2.z & 0xFFFFu, idx[5] = d2.y & 0xFFFFu, idx[8] = d2.x & 0xFFFFu;
                    }
                } else {
                    idx[0] = P8 ? (uint32_t)__builtin_amdgcn_raw_buffer_load_b8(planes_rs, (int)top, 0, 0)
                                : (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(planes_rs, (int)top, 0, 0);
                }
                int *cm = cnt + m * CDS_N_TRANSFORMS + o * 9u;
                // compares first (interval records three at a time), the rare counter updates afterwards: an LDS atomic between
                // the reads would pin their order and make every read a round trip of its own
                bool hit[NS];
#pragma unroll
                for (int k0 = 0; k0 < NS; k0 += 3) {
                    uint4 iv[3];
#pragma unroll
                    for (int k = 0; k < 3 && k0 + k < NS; k++) iv[k] = ivm[idx[k0 + k]];
#pragma unroll
                    for (int k = 0; k < 3 && k0 + k < NS; k++) hit[k0 + k] = ((code - iv[k].x) <= iv[k].y) | ((code - iv[k].z) <= iv[k].w);
                }
#pragma unroll
                for (int k = 0; k < NS; k++)
                    if (hit[k]) atomicAdd(&cm[k], 1);
                qh += n;
            };

            for (int r = 0; r < nruns; r++) {  // wave-uniform
                int yrel = ra - 1 - ry0;  // (rows started before the current block) - 1, relative to the map's first row
                const uint32_t sbase = (uint32_t)(ks0 + kin) * (uint32_t)p.strip_w;  // the run's strip: record columns are relative to it
                const uint32_t cx0r = cx0 - sbase, cm0r = cm0 - sbase;
                for (uint32_t i = a; i < b; i += 64 * NB) {  // wave-uniform
                    const uint32_t rem = b - i;
                    // Stage 1 of the NB blocks in PHASES, so that no block waits alone: (A) row, columns, cells, and the two words of
                    // the group's UNION band map (LDS, one word per cell, the single-mask kernel's 30 coarse bands) of every block;
                    // (B) the hue bin and, only for the lanes some mask of the group can accept at all, the byte per orientation of
                    // the fine map in L2 -- the gathers are what this kernel waits for, and their cost is the number of distinct
                    // lines the ACTIVE lanes touch; (C), after the next record blocks have been requested, the set bits become
                    // queue entries.  One LDS wait and one L2 wait per NB blocks.
                    uint32_t wp4[NB], wm4[NB], prow4[NB], cellp4[NB], cellm4[NB], up4[NB], um4[NB];
#if SM_TMAP_BINROWS
                    uint32_t rowq4[NB];
#endif
#pragma unroll
                    for (int j = 0; j < NB; j++) {
                        const uint32_t rec = cur[j];  // (lanes and whole blocks beyond the run were loaded as 0: no flag, below every threshold)
                        const unsigned long long fb = __ballot((int)rec < 0);
                        const uint32_t ry = (uint32_t)(yrel + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(fb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)fb, 0))) + (rec >> 31);
                        yrel += __popcll(fb);
                        const uint32_t x = (rec >> kRecXShift) & kRecXMask;  // relative to the strip: cx0r / cm0r carry its first column
                        const uint32_t cxp = min(x - cx0r, wnear);
                        const uint32_t cxm = MIRROR ? min(cm0r - x, wnear) : 0u;
                        const uint32_t rowc = __umul24(ry >> 2, nwpr);
                        cellp4[j] = rowc + (cxp >> 3);
                        cellm4[j] = rowc + (cxm >> 3);
                        prow4[j] = __umul24(ry, pitch) + poff;  // plane offset of column cx0 in the record's row
                        if (PF) {
                            up4[j] = umap[cellp4[j]];
                            um4[j] = MIRROR ? umap[cellm4[j]] : 0u;
                        }
#if SM_TMAP_BINROWS
                        rowq4[j] = ry >> 2;
                        cellp4[j] = cxp >> 3, cellm4[j] = cxm >> 3;   // (from here on: the cell COLUMNS; the fine map is [cell row][bin][cell column])
#endif
                    }
#pragma unroll
                    for (int j = 0; j < NB; j++) {
                        const uint32_t rec = cur[j];
                        // hue bin = class * 64 + floor(64 * ratio) (448 per cell; class 0 = bins no mask accepts, where a pixel below
                        // the data threshold is sent; a filler record gives NaN -> 0 and fails the threshold anyway)
                        const float hf = (float)((rec >> 8) & 255u) * __builtin_amdgcn_rcpf((float)(rec & 255u));
                        const uint32_t band = (rec >> kRecBandShift) & kRecBandMask;
                        const bool bright = (rec & 255u) > thr;
                        uint32_t bin = ((__umul24(band, 13u) >> 6) << kSparseGroupBinsLog2) + (uint32_t)(hf * (float)kSparseGroupBins);
                        if (!PF) bin = bright ? bin : 0u;   // (with the union map a dim pixel fails the map test below and reads byte 0 anyway)
                        bool passp = true, passm = MIRROR;
                        uint32_t bitp = 1u, bitm = MIRROR ? 1u : 0u;   // the union map's verdict as 0 / 1 (round 5: the offsets are MULTIPLIED by it --
                                                                      // hipcc turned the select into an exec-mask region per gather: a vector ->
                                                                      // scalar -> vector hop four times per block)

                        if (PF) {
                            const uint32_t cband = bright ? band : 3u;  // bit 3 is never set (v_bfe takes the band mod 32)
                            bitp = __builtin_amdgcn_ubfe(up4[j], cband, 1u);
                            if (MIRROR) bitm = __builtin_amdgcn_ubfe(um4[j], cband, 1u);
                            passp = bitp != 0, passm = bitm != 0;
                            (void)passp, (void)passm;
                        }
                        wp4[j] = 0, wm4[j] = 0;
#if defined(SM_EXP) && SM_EXP == 2   // (timing experiment: no map loads at all -- the floor of stage 1)
                        wp4[j] = (cellp4[j] + cellm4[j] + bin + (passp ? 1u : 0u) + (passm ? 2u : 0u)) == 0xFFFFFFFFu ? 1u : 0u;
#else
                        // (unconditional loads: a lane that did not pass reads byte 0 of the map -- cell 0, class 0: never set -- together
                        // with every other such lane, ONE line; a branch around the load would make hipcc wait for it at the join)
#if SM_TMAP_BINROWS
                        // fine map laid out [cell row][bin][cell column]: the records of one row that share a hue bin -- the pixels of one
                        // stroke, and of its neighbours in depth -- read neighbouring BYTES, so the active lanes of a gather touch a handful
                        // of lines instead of one per cell (this kernel waits for the number of distinct lines its gathers touch)
                        const uint32_t rowbin = __umul24(rowq4[j], 7u * kSparseGroupBins) + bin;
                        const uint32_t offp = (__umul24(rowbin, nwpr) + cellp4[j]) & (0u - bitp);   // (an AND, not a 24-bit multiply: a full-frame mask's map exceeds 2^24 bytes)
#else
                        const uint32_t offp = passp ? __umul24(cellp4[j], 7u * kSparseGroupBins) + bin : 0u;
#endif
                        wp4[j] = __builtin_amdgcn_raw_buffer_load_b8(tmap_rs, (int)offp, 0, 0);
                        if (MIRROR) {
#if SM_TMAP_BINROWS
                            const uint32_t offm = (__umul24(rowbin, nwpr) + cellm4[j]) & (0u - bitm);
#else
                            const uint32_t offm = passm ? __umul24(cellm4[j], 7u * kSparseGroupBins) + bin : 0u;
#endif
                            wm4[j] = __builtin_amdgcn_raw_buffer_load_b8(tmap_rs, (int)offm, 0, 0);
                        }
#endif
                    }
                    {   // the next NB blocks -- of this run or the first of the grab's next run -- are in flight meanwhile
                        const uint32_t *Pq = P;
                        uint32_t iq = i, bq = b;
                        if (rem > 64 * NB) iq = i + 64 * NB;
                        else if (r + 1 < nruns) header(r + 1, Pq, iq, bq);
                        load_blocks(Pq, iq, bq, nxt);
                    }
#pragma unroll
                    for (int j = 0; j < NB; j++) {
                        const uint32_t rec = cur[j], prow = prow4[j];
                        uint32_t w = wp4[j] | (wm4[j] << 8);
#if defined(SM_EXP) && SM_EXP == 1   // (timing experiment: the map bytes are loaded and not used -- no entries, no rounds; scores become 0)
                        asm volatile("" ::"v"(w));
                        w = 0;
#endif
                        unsigned long long bal = __ballot(w != 0);
                        while (bal) {  // one turn per set bit of the busiest lane
                            if (w != 0) {
                                const uint32_t bit = (uint32_t)__builtin_ctz(w);
                                const int slot = (qt + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0))) & (SM_QCAP - 1);
                                const uint32_t x = (rec >> kRecXShift) & kRecXMask;
                                const uint32_t cx = (MIRROR && bit >= 8u) ? min(cm0r - x, wnear) : min(x - cx0r, wnear);
                                q[slot] = make_uint2(rec, (prow + cx) | (bit << 20));
                                w &= w - 1u;
                            }
                            qt += __popcll(bal);
                            if (qt - qh >= 64) round();
                            bal = __ballot(w != 0);
                        }
                    }
#pragma unroll
                    for (int j = 0; j < NB; j++) cur[j] = nxt[j];
                }
                if (r + 1 < nruns) header(r + 1, P, a, b);
                if (++kin < nsv) continue;  // more strips of this target
                kin = 0;
                const uint32_t t = t0 + (uint32_t)(r / nsv);
                while (qt > qh) round();
                __builtin_amdgcn_wave_barrier();
                if (lane < p.n_masks) {
                    const int *cl = cnt + lane * CDS_N_TRANSFORMS;
                    int best = 0, bm = 0;
#pragma unroll
                    for (int k = 0; k < 9; k++) {
                        best = max(best, cl[k]);
                        bm = max(bm, cl[9 + k]);
                    }
                    int32_t packed = best;
                    if (MIRROR && bm > best) packed = (int32_t)((uint32_t)bm | 0x80000000u);
                    p.out[(int64_t)lane * p.out_stride + t] = packed;
                }
                __builtin_amdgcn_wave_barrier();
                for (int i = lane; i < K * CDS_N_TRANSFORMS; i += 64) cnt[i] = 0;
                __builtin_amdgcn_wave_barrier();
                qh = qt = 0;
            }
        }
    }
}

// =================================================================================================
//                                   host: mask plans
// =================================================================================================
static size_t slab_image_bytes(int R, int pitch, int nwpr, int esize, int *near_off) {
    const size_t plane = (size_t)round_up((R + 2 * SP_PAD) * pitch * esize, 16);
    if (near_off) *near_off = (int)plane;
#if SP_BAND
    // band map: one word per 4-row x 8-column cell, rows 0 .. R + 3 of target positions -- and one more, always empty, cell row for
    // row R + 4, where the 16-bit-stream scan lands the one record it may read past the end of a run (sparse_scan16_kernel)
    return plane + (size_t)round_up(((R + 8) / 4) * nwpr * 4, 16);
#else
    return plane + (size_t)round_up((R + 4) * nwpr * 4, 16);
#endif
}

#if SP_BAND
// Hue bands (bit = (class * 5 + floor(5 * ratio)) & 31, classes 1..6) that codes of the inclusive interval [lo, lo + len] can
// fall in.  The band is a monotone function of the code inside a class, so the bands of the two ends bound those of
// every code between them; the arithmetic is the kernel's (float multiply, truncating conversion).
static uint32_t band_bits_of(uint32_t lo, uint32_t len) {
    if (lo == 1u && len == 0u) return 0;  // the empty record
    const uint32_t hi = lo + len;
    uint32_t bits = 0;
    auto u2f = [](uint32_t u) { float f; memcpy(&f, &u, 4); return f; };
    // classes 1..6 and bins 0..4 only (a record's ratio is < 1): bands 5..34, i.e. bits 5..31 and 0..2 -- bits 3 and 4 stay
    // clear in every word, bit 3 is where the kernel sends the pixels below the data threshold
    for (uint32_t c = std::max(1u, lo >> 29); c <= std::min(6u, hi >> 29); c++) {
        const float flo = c == (lo >> 29) ? u2f((lo & 0x1FFFFFFFu) << 1) : 0.0f;
        const float fhi = c == (hi >> 29) ? u2f(((hi & 0x1FFFFFFFu) << 1) | 1u) : 1.0f;  // (the code dropped the ratio's last bit)
        const uint32_t b0 = (uint32_t)(flo * 5.0f), b1 = std::min(4u, (uint32_t)(fhi * 5.0f));
        for (uint32_t b = b0; b <= b1; b++) bits |= 1u << ((c * 5u + b) & 31u);
    }
    return bits;
}
#endif

void free_sparse_plans(cds_maskset *ms) {
    // (back to the context's block cache: hipFree would wait for every stream of the device, i.e. for the scan of the NEXT mask set when
    // the caller pipelines them; a mask set is destroyed only when no search of it is in flight)
    cds_context *cx = ms->ctx;
    if (ms->sp_last_use) {   // an asynchronous search (cds_search_sparse_device) may still be reading them: wait for THAT launch, not for the stream
        hipEventSynchronize(ms->sp_last_use);
        hipEventDestroy(ms->sp_last_use);
        ms->sp_last_use = nullptr;
    }
    cache_free(cx, ms->d_s16_stat);
    ms->d_s16_stat = nullptr;
    cache_free(cx, ms->d_sp_slabs);
    cache_free(cx, ms->d_sp_iv);
    cache_free(cx, ms->d_sp_work);
    cache_free(cx, ms->d_sg_planes);
    cache_free(cx, ms->d_sg_tmap);
    cache_free(cx, ms->d_sg_iv);
    ms->d_sg_planes = ms->d_sg_tmap = nullptr;
    ms->d_sg_iv = nullptr;
    ms->sg_plans.clear();
    ms->d_sp_slabs = nullptr;
    ms->d_sp_iv = nullptr;
    ms->d_sp_work = nullptr;
    ms->sp_plans.clear();
    std::vector<std::vector<uint8_t>>().swap(ms->sp_host_slabs);
    std::vector<std::vector<uint4>>().swap(ms->sp_host_iv);
    ms->sp_uploaded = false;
}

static int sparse_threads() {
    int t = 1024;
    if (const char *e = getenv("CDS_SPARSE_THREADS")) t = atoi(e);
    t = std::max(64, std::min(1024, t)) / 64 * 64;
    return t;
}


#if SP_BAND
// Hue bins (class * kSparseGroupBins + floor(kSparseGroupBins * ratio), classes 1..6) that codes of the inclusive interval
// [lo, lo + len] can fall in -- the fine-grained counterpart of band_bits_of for the multi-mask scan's map (same argument:
// the bin is monotone in the code inside a class; the arithmetic is the kernel's).
static void bins_of(uint32_t lo, uint32_t len, std::vector<uint16_t> &out) {
    if (lo == 1u && len == 0u) return;  // the empty record
    const uint32_t hi = lo + len;
    auto u2f = [](uint32_t u) { float f; memcpy(&f, &u, 4); return f; };
    for (uint32_t c = std::max(1u, lo >> 29); c <= std::min(6u, hi >> 29); c++) {
        const float flo = c == (lo >> 29) ? u2f((lo & 0x1FFFFFFFu) << 1) : 0.0f;
        const float fhi = c == (hi >> 29) ? u2f(((hi & 0x1FFFFFFFu) << 1) | 1u) : 1.0f;  // (the code dropped the ratio's last bit)
        const uint32_t b0 = (uint32_t)(flo * (float)kSparseGroupBins), b1 = std::min((uint32_t)kSparseGroupBins - 1u, (uint32_t)(fhi * (float)kSparseGroupBins));
        for (uint32_t b = b0; b <= b1; b++) out.push_back((uint16_t)(c * kSparseGroupBins + b));
    }
}
#endif

// distinct interval records of a mask (index 0 = the empty record) and, per mask pixel, the index of its record;
// false when the mask has more than `limit` - 1 distinct records
struct IvKey {
    uint32_t a, b, c, d;
    bool operator==(const IvKey &o) const { return a == o.a && b == o.b && c == o.c && d == o.d; }
};
struct IvKeyHash {
    size_t operator()(const IvKey &k) const {
        uint64_t h = (uint64_t)k.a * 0x9E3779B97F4A7C15ull ^ ((uint64_t)k.b << 32 | k.c) * 0xC2B2AE3D27D4EB4Full ^ k.d;
        return (size_t)(h ^ (h >> 29));
    }
};
static bool mask_dictionary(const std::vector<MaskPx> &L, size_t limit, std::vector<uint4> &tab, std::vector<uint16_t> &pidx) {
    std::unordered_map<IvKey, uint32_t, IvKeyHash> dict;
    tab.clear();
    tab.push_back(make_uint4(1u, 0u, 1u, 0u));
    pidx.resize(L.size());
    for (size_t i = 0; i < L.size(); i++) {
        const IvKey k{L[i].lo1, L[i].len1, L[i].lo2, L[i].len2};
        auto it = dict.find(k);
        if (it == dict.end()) {
            if (tab.size() >= limit) return false;
            it = dict.emplace(k, (uint32_t)tab.size()).first;
            tab.push_back(make_uint4(k.a, k.b, k.c, k.d));
        }
        pidx[i] = (uint16_t)it->second;
    }
    return true;
}

// Plans of the multi-mask scan: groups of kSparseGroupMasks consecutive masks over the union of their bounding boxes.
static int build_group_plans(cds_maskset *ms, const std::vector<std::vector<MaskPx>> &per_mask) {
    const size_t n_masks = per_mask.size();
    if ((n_masks < 2 && !getenv("CDS_SPARSE_GROUP_SINGLE")) || !SP_BAND || getenv("CDS_SPARSE_NO_GROUPS")) return CDS_OK;
#if SP_BAND
    const bool xy2 = ms->params.xy_shift == 2;
    constexpr int K = kSparseGroupMasks;
    // consecutive masks, at most K, whose interval tables fit the LDS table together (sp_plans[m].n_iv = records of mask m,
    // the empty record included; a mask without device pixels has none and joins any group)
    int kmax = K;
    if (const char *e = getenv("CDS_SPARSE_GROUP_MASKS")) kmax = std::max(1, std::min(K, atoi(e)));   // (experiments: smaller groups)
    for (size_t m = 0; m < n_masks;) {
        SparseGroupPlan gp;
        gp.first_mask = (int)m;
        int sum = 0;
        while (m < n_masks && gp.n_masks < kmax) {
            const int niv = std::max(1, ms->sp_plans[m].n_iv);
            if (gp.n_masks > 0 && sum + niv > kSparseGroupIv) break;
            sum += niv;
            gp.n_masks++;
            m++;
        }
        ms->sg_plans.push_back(gp);
    }
    const size_t n_groups = ms->sg_plans.size();
    std::vector<std::vector<uint8_t>> planes_of(n_groups), tmap_of(n_groups);
    std::vector<std::vector<uint4>> iv_of(n_groups);
    auto plan_group = [&](size_t g) {
        SparseGroupPlan &gp = ms->sg_plans[g];
        const size_t m0 = (size_t)gp.first_mask;
        int xlo = ms->W, xhi = -1, ylo = ms->H, yhi = -1;
        for (int j = 0; j < gp.n_masks; j++) {
            if (ms->sp_plans[m0 + j].n_slabs < 0) return;  // (a mask the single-mask scan cannot lay out either)
            for (const MaskPx &q : per_mask[m0 + j]) {
                xlo = std::min(xlo, q.x), xhi = std::max(xhi, q.x);
                ylo = std::min(ylo, q.y), yhi = std::max(yhi, q.y);
            }
        }
        if (xhi < 0) return;  // no mask of the group can match anything: the per-mask path writes the zeros
        gp.xlo = xlo, gp.cols = xhi - xlo + 1, gp.ylo = ylo, gp.nrows = yhi - ylo + 1;
        const int G = round_up(gp.cols + 4, 8);  // (the single-mask geometry: guard cell at column G, plane wide enough for its reads)
        gp.pitch = round_up(G + 4, 8);
        gp.nwpr = G / 8 + 1;
        if ((size_t)(gp.nrows + 2 * SP_PAD) * gp.pitch >= ((size_t)1 << 20)) return;  // plane offsets travel in 20 bits
        gp.plane_cells = round_up((gp.nrows + 2 * SP_PAD) * gp.pitch, 16);
        gp.tmap_bytes = (size_t)((gp.nrows + 4 + 3) / 4) * gp.nwpr * 7 * kSparseGroupBins;
        gp.plane8 = true;
        for (int j = 0; j < gp.n_masks; j++) gp.plane8 = gp.plane8 && ms->sp_plans[m0 + j].n_iv <= 256;
        const int esize = gp.plane8 ? 1 : 2;
        std::vector<uint8_t> &planes = planes_of[g], &tmap = tmap_of[g];
        std::vector<uint4> &ivs = iv_of[g];
        planes.assign((size_t)gp.n_masks * gp.plane_cells * esize, 0);
        // the fine map, followed by the union band map (one word per cell)
        const size_t n_cells = (size_t)((gp.nrows + 4 + 3) / 4) * gp.nwpr;
        gp.umap_off = (size_t)round_up((int)gp.tmap_bytes, 16);
        tmap.assign(gp.umap_off + n_cells * 4, 0);
        uint32_t *umap = reinterpret_cast<uint32_t *>(tmap.data() + gp.umap_off);
        ivs.clear();
        std::vector<uint4> tab;
        std::vector<uint16_t> pidx;
        std::vector<std::vector<uint16_t>> tab_bins;
        std::vector<uint32_t> cells;   // (cell << 12 | record) pairs of the mask, deduplicated before the bins are written
        for (int j = 0; j < gp.n_masks; j++) {
            const std::vector<MaskPx> &L = per_mask[m0 + j];
            if (!mask_dictionary(L, kSparseGroupIv + 1, tab, pidx) || ivs.size() + tab.size() > (size_t)kSparseGroupIv) {
                planes.clear(), tmap.clear(), ivs.clear();
                return;  // (a single mask with more records than the LDS table holds: scanned by the single-mask kernel)
            }
            gp.iv_base[j] = (int)ivs.size();
            ivs.insert(ivs.end(), tab.begin(), tab.end());
            tab_bins.assign(tab.size(), std::vector<uint16_t>());
            std::vector<uint32_t> tab_bits(tab.size());
            for (size_t k = 0; k < tab.size(); k++) {
                tab_bits[k] = band_bits_of(tab[k].x, tab[k].y) | band_bits_of(tab[k].z, tab[k].w);
                bins_of(tab[k].x, tab[k].y, tab_bins[k]);
                bins_of(tab[k].z, tab[k].w, tab_bins[k]);
            }
            cells.clear();
            uint8_t *plane = planes.data() + (size_t)j * gp.plane_cells * esize;
            uint16_t *plane16 = reinterpret_cast<uint16_t *>(plane);
            for (size_t i = 0; i < L.size(); i++) {
                const int rr = L[i].y - ylo, cc = L[i].x - xlo;
                const size_t pc = (size_t)(rr + SP_PAD) * gp.pitch + cc + SP_PAD;
                if (gp.plane8) plane[pc] = (uint8_t)pidx[i];
                else plane16[pc] = pidx[i];
                for (int dy = xy2 ? -2 : 0; dy <= (xy2 ? 2 : 0); dy += 2)
                    for (int dx = xy2 ? -2 : 0; dx <= (xy2 ? 2 : 0); dx += 2) {
                        const int ry = rr + 2 + dy, cx = cc + 2 + dx;  // map origin = (xlo - 2, ylo - 2)
                        cells.push_back((uint32_t)(((ry >> 2) * gp.nwpr + (cx >> 3)) << 12) | pidx[i]);
                    }
            }
            std::sort(cells.begin(), cells.end());
            cells.erase(std::unique(cells.begin(), cells.end()), cells.end());
            for (const uint32_t cr : cells) {
#if SM_TMAP_BINROWS
                const size_t crow = (size_t)(cr >> 12) / (size_t)gp.nwpr, ccol = (size_t)(cr >> 12) % (size_t)gp.nwpr;
                for (const uint16_t b : tab_bins[cr & 4095u]) tmap[(crow * 7 * kSparseGroupBins + b) * (size_t)gp.nwpr + ccol] |= (uint8_t)(1u << j);
#else
                uint8_t *cell = tmap.data() + (size_t)(cr >> 12) * 7 * kSparseGroupBins;
                for (const uint16_t b : tab_bins[cr & 4095u]) cell[b] |= (uint8_t)(1u << j);
#endif
                umap[cr >> 12] |= tab_bits[cr & 4095u];
            }
        }
        gp.n_iv = (int)ivs.size();
        gp.umap_words = (int)n_cells;
        if (multi_lds_bytes(1024, gp.n_iv, gp.umap_words) > ms->ctx->lds_per_block / 2) gp.umap_words = 0;  // keep two workgroups per CU
        gp.ok = true;
    };
    {
        const unsigned hc = std::thread::hardware_concurrency();
        const int n_threads = std::max(1, std::min<int>((int)n_groups, (int)std::min<unsigned>(hc ? hc : 1u, 16u)));
        std::atomic<size_t> next{0};
        auto worker = [&]() {
            for (size_t g = next.fetch_add(1); g < n_groups; g = next.fetch_add(1)) plan_group(g);
        };
        std::vector<std::thread> pool;
        for (int i = 1; i < n_threads; i++) pool.emplace_back(worker);
        worker();
        for (auto &th : pool) th.join();
    }
    size_t planes_total = 0, tmap_total = 0, iv_total = 0;
    for (size_t g = 0; g < n_groups; g++) {
        SparseGroupPlan &gp = ms->sg_plans[g];
        if (!gp.ok) continue;
        gp.planes_off = planes_total, gp.tmap_off = tmap_total, gp.iv_off = iv_total;
        planes_total += planes_of[g].size();
        tmap_total += tmap_of[g].size();
        iv_total += iv_of[g].size();
    }
    if (planes_total == 0) {
        ms->sg_plans.clear();
        return CDS_OK;
    }
    hipStream_t up = ms->ctx->upload_stream ? ms->ctx->upload_stream : ms->ctx->stream;
    hipError_t e = cache_alloc(ms->ctx, (void **)&ms->d_sg_planes, planes_total);
    if (e == hipSuccess) e = cache_alloc(ms->ctx, (void **)&ms->d_sg_tmap, tmap_total);
    if (e == hipSuccess) e = cache_alloc(ms->ctx, (void **)&ms->d_sg_iv, iv_total * sizeof(uint4));
    for (size_t g = 0; g < n_groups && e == hipSuccess; g++) {
        const SparseGroupPlan &gp = ms->sg_plans[g];
        if (!gp.ok) continue;
        // (on the context's upload stream: these copies neither wait for the scans of earlier mask sets nor stall them)
        e = hipMemcpyAsync(ms->d_sg_planes + gp.planes_off, planes_of[g].data(), planes_of[g].size(), hipMemcpyHostToDevice, up);
        if (e == hipSuccess) e = hipMemcpyAsync(ms->d_sg_tmap + gp.tmap_off, tmap_of[g].data(), tmap_of[g].size(), hipMemcpyHostToDevice, up);
        if (e == hipSuccess) e = hipMemcpyAsync(ms->d_sg_iv + gp.iv_off, iv_of[g].data(), iv_of[g].size() * sizeof(uint4), hipMemcpyHostToDevice, up);
    }
    {
        const hipError_t es = hipStreamSynchronize(up);   // (the host vectors go out of scope; later searches run on another stream)
        if (e == hipSuccess) e = es;
    }
    if (e != hipSuccess) return hip_fail(e, "sparse group plans");
#endif
    return CDS_OK;
}

// the single-mask plans' slab images and interval tables -> device (once; the caller holds sp_mu or is the creating thread)
static int upload_single_plans(const cds_maskset *ms) {
    if (ms->sp_uploaded) return CDS_OK;
    cds_context *ctx = ms->ctx;
    cds_maskset *w = const_cast<cds_maskset *>(ms);
    hipSetDevice(ctx->device);
    hipStream_t up = ctx->upload_stream ? ctx->upload_stream : ctx->stream;   // (neither behind the scans of `stream` nor stalling them)
    hipError_t e = cache_alloc(ctx, (void **)&w->d_sp_slabs, std::max<size_t>(ms->sp_slab_total, 16));
    if (e == hipSuccess) e = cache_alloc(ctx, (void **)&w->d_sp_iv, std::max<size_t>(ms->sp_iv_total, 1) * sizeof(uint4));
    for (size_t m = 0; m < ms->sp_host_slabs.size() && e == hipSuccess; m++) {
        if (ms->sp_host_slabs[m].empty()) continue;
        e = hipMemcpyAsync(w->d_sp_slabs + ms->sp_plans[m].slabs_off, ms->sp_host_slabs[m].data(), ms->sp_host_slabs[m].size(), hipMemcpyHostToDevice, up);
        if (e == hipSuccess)
            e = hipMemcpyAsync(w->d_sp_iv + ms->sp_plans[m].iv_off, ms->sp_host_iv[m].data(), ms->sp_host_iv[m].size() * sizeof(uint4), hipMemcpyHostToDevice, up);
    }
    const hipError_t es = hipStreamSynchronize(up);
    if (e == hipSuccess) e = es;
    if (e != hipSuccess) {
        cache_free(ctx, w->d_sp_slabs);
        cache_free(ctx, w->d_sp_iv);
        w->d_sp_slabs = nullptr;
        w->d_sp_iv = nullptr;
        return hip_fail(e, "sparse mask plans (upload)");
    }
    std::vector<std::vector<uint8_t>>().swap(w->sp_host_slabs);
    std::vector<std::vector<uint4>>().swap(w->sp_host_iv);
    w->sp_uploaded = true;
    return CDS_OK;
}

int build_sparse_plans(cds_maskset *ms, const std::vector<std::vector<MaskPx>> &per_mask) {
    cds_context *ctx = ms->ctx;
    const bool xy2 = ms->params.xy_shift == 2;
    const int threads = sparse_threads();
    size_t budget = (size_t)80 << 10;  // two workgroups of 1024 threads per CU
    if (const char *e = getenv("CDS_SPARSE_LDS_KB")) budget = (size_t)std::max(8, atoi(e)) << 10;
    budget = std::min(budget, ctx->lds_per_block);
    const size_t n_masks = per_mask.size();
    PrepTimer pt;
    ms->sp_plans.assign(n_masks, SparseMaskPlan());
    std::vector<std::vector<uint8_t>> slab_of(n_masks);
    std::vector<std::vector<uint4>> iv_of(n_masks);
    auto plan_one = [&](size_t m) {
        const std::vector<MaskPx> &L = per_mask[m];
        SparseMaskPlan &pl = ms->sp_plans[m];
        if (L.empty()) return;  // n_slabs == 0: nothing can match
        int xlo = ms->W, xhi = -1, ylo = ms->H, yhi = -1;
        for (const MaskPx &q : L) {
            xlo = std::min(xlo, q.x), xhi = std::max(xhi, q.x);
            ylo = std::min(ylo, q.y), yhi = std::max(yhi, q.y);
        }
        pl.xlo = xlo, pl.cols = xhi - xlo + 1, pl.ylo = ylo, pl.nrows = yhi - ylo + 1;
#if SP_BAND
        // columns are clamped to G = the first column of an always-empty guard cell; a pixel between the window's end and G
        // can pass the cell test, so the plane is wide enough (and zero there) for its nine reads: columns up to G + 3
        const int G = round_up(pl.cols + 4, 8);
        pl.pitch = round_up(G + 4, 8);
        pl.nwpr = G / 8 + 1;                    // band-map words per cell row, with the guard cell
#else
        pl.pitch = round_up(pl.cols + 2 * SP_PAD, 8);
        pl.nwpr = (pl.cols + 4 + 1 + 31) / 32;  // + the guard bit behind the last column
#endif
        // distinct interval records of this mask; index 0 = the empty record
        std::vector<uint4> &tab = iv_of[m];
        std::vector<uint16_t> pidx;
        if (!mask_dictionary(L, 65535, tab, pidx)) {
            pl.n_slabs = -1;  // (a mask with more than 65,534 distinct interval records: use the dense scan)
            tab.clear();
            return;
        }
        pl.n_iv = (int)tab.size();
        pl.iv_in_lds = pl.n_iv <= SP_MAX_IV_LDS;
        pl.plane8 = pl.n_iv <= 256 && !getenv("CDS_SPARSE_PLANE16");
        const int esize = pl.plane8 ? 1 : 2;
        const size_t fixed = (pl.iv_in_lds ? (size_t)pl.n_iv * 16 : 0) + (size_t)(threads / 64) * SP_QBYTES;
        auto fits = [&](int R, size_t bud) {
            return fixed + slab_image_bytes(R, pl.pitch, pl.nwpr, esize, nullptr) <= bud;
        };
        // 80 KB = two workgroups of 1024 threads per CU (8 waves per SIMD, 64 VGPRs); the whole LDS = one workgroup per CU
        // (4 waves per SIMD, 128 VGPRs: more LDS requests in flight per wave).  Fewer slabs win: every slab re-reads 4 halo rows
        // of records, ends every target with partial rounds and adds a partial-count pass.  Measured per 262,144 targets:
        // the benchmark mask 1 slab / 160 KB 1.9 ms vs 3 slabs / 80 KB 2.1 ms; masks with 2-byte planes (> 255 interval
        // records) 2 slabs / 160 KB vs 7 slabs / 80 KB 3.5-4.2 ms.  So: the larger budget whenever it needs fewer slabs.
        auto rows_for = [&](size_t bud) {
            int R = 0;
            while (R < pl.nrows && fits(R + 1, bud)) R++;
            return R;
        };
        int R = rows_for(budget);
        if (!getenv("CDS_SPARSE_LDS_KB") || R == 0) {
            const int R2 = rows_for(ctx->lds_per_block);
            if (R2 > 0 && (R == 0 || (pl.nrows + R2 - 1) / R2 < (pl.nrows + R - 1) / R)) R = R2;
        }
        if (R == 0) {
            pl.n_slabs = -1;
            tab.clear();
            return;
        }
        pl.n_slabs = (pl.nrows + R - 1) / R;
        if (pl.n_slabs > SP_MAX_SLABS) {
            pl.n_slabs = -1;
            tab.clear();
            return;
        }
        R = (pl.nrows + pl.n_slabs - 1) / pl.n_slabs;
        pl.slab_rows = R;
        pl.slab_bytes = (int)slab_image_bytes(R, pl.pitch, pl.nwpr, esize, &pl.near_off);
        pl.lds_bytes = fixed + (size_t)pl.slab_bytes;
#if SP_BAND
        {   // the two-orientation map of sparse_scan16_kernel: columns relative to the union of the mask's window and its mirror image
            const int cx0 = xlo - 2, cm0 = ms->W - 1 - cx0;
            int ulo = cx0, uhi = cx0 + pl.cols + 3;
            if (ms->params.mirror) ulo = std::min(ulo, cm0 - (pl.cols + 3)), uhi = std::max(uhi, cm0);
            pl.ux0 = ulo;
            pl.g64 = round_up(uhi - ulo + 1, 8);
            pl.rowb64 = (pl.g64 / 8 + 1) * 8;
            const size_t mb = (size_t)round_up(((R + 8) / 4) * pl.rowb64, 16);
            pl.map64_bytes = mb <= 65528 ? (int)mb : 0;   // (cell addresses are 16-bit offsets; a wider or taller map: the 4-byte records)
        }
#endif
        std::vector<uint8_t> &slabs = slab_of[m];
        slabs.assign((size_t)pl.n_slabs * ((size_t)pl.slab_bytes + (size_t)pl.map64_bytes), 0);
#if SP_BAND
        std::vector<uint32_t> tab_bits(tab.size());
        for (size_t k = 0; k < tab.size(); k++) tab_bits[k] = band_bits_of(tab[k].x, tab[k].y) | band_bits_of(tab[k].z, tab[k].w);
        // the stream's band field is the record's XOR 3 (sparse_px16_kernel) and the kernel tests it by shifting the word LEFT by the field and
        // looking at the sign: bit 31 - ((b ^ 3) & 31) = (b ^ 28) & 31 for band b -- and bit 31 for the field 0 that everything beyond a run
        // decodes to, which is band 3's: never set
        std::vector<uint32_t> tab_bits16(pl.map64_bytes ? tab.size() : 0);
        for (size_t k = 0; k < tab_bits16.size(); k++) {
            uint32_t w = 0;
            for (uint32_t b = 0; b < 32; b++)
                if (tab_bits[k] >> b & 1u) w |= 1u << (b ^ 28u);
            tab_bits16[k] = w;
        }
#endif
        for (int s = 0; s < pl.n_slabs; s++) {
            uint8_t *img = slabs.data() + (size_t)s * pl.slab_bytes;
            uint16_t *plane16 = reinterpret_cast<uint16_t *>(img);
            uint32_t *nearb = reinterpret_cast<uint32_t *>(img + pl.near_off);
            const int ys0 = ylo + s * R;
#if SP_BAND
            uint32_t *map64 = pl.map64_bytes ? reinterpret_cast<uint32_t *>(slabs.data() + (size_t)pl.n_slabs * pl.slab_bytes + (size_t)s * pl.map64_bytes) : nullptr;
            const int w64 = pl.rowb64 / 4;   // dwords per cell row
#endif
            for (size_t i = 0; i < L.size(); i++) {
                const int rr = L[i].y - ys0;
                if (rr < 0 || rr >= R) continue;
                const size_t cell = (size_t)(rr + SP_PAD) * pl.pitch + (L[i].x - xlo) + SP_PAD;
                if (pl.plane8) img[cell] = (uint8_t)pidx[i];
                else plane16[cell] = pidx[i];
                // the target pixels that reach this mask pixel under some shift: (x + dx, y + dy)
                for (int dy = xy2 ? -2 : 0; dy <= (xy2 ? 2 : 0); dy += 2)
                    for (int dx = xy2 ? -2 : 0; dx <= (xy2 ? 2 : 0); dx += 2) {
                        const int ry = rr + 2 + dy, cx = (L[i].x - xlo) + 2 + dx;  // bitmap origin = (xlo - 2, ys0 - 2)
#if SP_BAND
                        nearb[(size_t)(ry >> 2) * pl.nwpr + (cx >> 3)] |= tab_bits[pidx[i]];
                        if (map64) {   // target column x + dx as it is, and where the mirrored mask has this pixel: W - 1 - x, shifted
                            const int up = L[i].x + dx - pl.ux0, um = ms->W - 1 - L[i].x + dx - pl.ux0;
                            map64[(size_t)(ry >> 2) * w64 + (size_t)(up >> 3) * 2] |= tab_bits16[pidx[i]];
                            if (ms->params.mirror) map64[(size_t)(ry >> 2) * w64 + (size_t)(um >> 3) * 2 + 1] |= tab_bits16[pidx[i]];
                        }
#else
                        nearb[(size_t)ry * pl.nwpr + (cx >> 5)] |= 1u << (cx & 31);
#endif
                    }
            }
                    }
    };
    {
        const unsigned hc = std::thread::hardware_concurrency();
        const int n_threads = std::max(1, std::min<int>((int)n_masks, (int)std::min<unsigned>(hc ? hc : 1u, 16u)));
        std::atomic<size_t> next{0};
        auto worker = [&]() {
            for (size_t m = next.fetch_add(1); m < n_masks; m = next.fetch_add(1)) plan_one(m);
        };
        std::vector<std::thread> pool;
        for (int i = 1; i < n_threads; i++) pool.emplace_back(worker);
        worker();
        for (auto &th : pool) th.join();
    }
    pt.lap("  single-mask plans (host)");
    size_t slab_total = 0, iv_total = 0;
    for (size_t m = 0; m < n_masks; m++) {
        ms->sp_plans[m].slabs_off = slab_total;
        ms->sp_plans[m].iv_off = iv_total;
        slab_total += slab_of[m].size();
        iv_total += iv_of[m].size();
    }
    hipSetDevice(ctx->device);
    hipError_t e = hipSuccess;
    if (slab_total > 0) e = cache_alloc(ctx, (void **)&ms->d_sp_work, (size_t)SP_MAX_SLABS * SP_MAX_PARTS * 32 * sizeof(uint32_t));
    if (e != hipSuccess) {
        free_sparse_plans(ms);
        return hip_fail(e, "sparse mask plans");
    }
    ms->sp_host_slabs = std::move(slab_of);
    ms->sp_host_iv = std::move(iv_of);
    ms->sp_slab_total = slab_total;
    ms->sp_iv_total = iv_total;
    ms->sp_uploaded = slab_total == 0;
    if (n_masks == 1 || getenv("CDS_SPARSE_EAGER_UPLOAD")) {   // a single mask is scanned by the single-mask kernel: upload now
        const int rc = upload_single_plans(ms);
        if (rc) {
            free_sparse_plans(ms);
            return rc;
        }
    }
    pt.lap("  single-mask plans (upload)");
    if (slab_total > 0) {
        const int rc = build_group_plans(ms, per_mask);
        if (rc) {
            free_sparse_plans(ms);
            return rc;
        }
    }
    pt.lap("  group plans");
    return CDS_OK;
}

// can a library go through sparse_scan16_kernel at all?  (It holds the 16-bit stream.)
static bool stream16_possible(const SparseScanParams &p) { return p.px16 != nullptr && p.map64_bytes > 0; }

template <bool XY2, bool MIRROR>
static int launch_sparse_one(cds_context *ctx, const SparseScanParams &p, size_t lds_bytes, int threads, hipStream_t st) {
    const bool wide = lds_bytes > ctx->lds_per_block / 2 && !getenv("CDS_SPARSE_WPE8");  // one workgroup per CU: 4 waves per SIMD, 128 VGPRs
    const bool s16 = p.use16 != 0;   // (decided by the caller: sparse_search_impl)
#define SP_PICK(K) (wide ? (p.plane8 ? K<XY2, MIRROR, true, true, 4>   /* (<= 255 records: the table is in LDS) */ \
                         : p.iv_in_lds ? K<XY2, MIRROR, true, false, 4> : K<XY2, MIRROR, false, false, 4>)       \
                      : (p.plane8 ? K<XY2, MIRROR, true, true, 8>                                              \
                         : p.iv_in_lds ? K<XY2, MIRROR, true, false, 8> : K<XY2, MIRROR, false, false, 8>))
    auto kern = s16 ? SP_PICK(sparse_scan16_kernel) : SP_PICK(sparse_scan_kernel);
#undef SP_PICK
    if (s16 && p.plane_global) {   // (the plane in L2: the LDS image is the band map alone, two workgroups per CU)
        kern = p.plane8 ? sparse_scan16_kernel<XY2, MIRROR, true, true, 8, false, true>
             : p.iv_in_lds ? sparse_scan16_kernel<XY2, MIRROR, true, false, 8, false, true> : sparse_scan16_kernel<XY2, MIRROR, false, false, 8, false, true>;
    }
    if (ctx->sparse_load_only.load()) {   // (a diagnostic of THIS context only: cds_debug_sparse_load_only)
        // the twin runs in the layout of the scan it stands for: the stream kernel's with the plane in L2 (two workgroups per CU) or in LDS,
        // the 4-byte-record kernel's with one workgroup per CU; one-byte planes only
        const bool twin = p.plane8 && (s16 ? (p.plane_global || wide) : wide);
        if (!twin) return fail(CDS_ERR_UNSUPPORTED, "the load-only diagnostic exists for one-byte-plane layouts only (4-byte records: one workgroup per CU only)");
        if (s16 && p.plane_global) kern = sparse_scan16_kernel<XY2, MIRROR, true, true, 8, true, true>;
        else kern = s16 ? sparse_scan16_kernel<XY2, MIRROR, true, true, 4, true> : sparse_scan_kernel<XY2, MIRROR, true, true, 4, true>;
    }
    if (s16) {   // the stream kernel addresses its map from LDS address 0: true as long as the kernel has no static LDS in front of the dynamic one
        static std::mutex mu;
        static std::vector<const void *> checked;   // (one driver query per kernel variant and process)
        std::lock_guard<std::mutex> lk(mu);
        const void *kp = reinterpret_cast<const void *>(kern);
        if (std::find(checked.begin(), checked.end(), kp) == checked.end()) {
            hipFuncAttributes fa{};
            CDS_HIP(hipFuncGetAttributes(&fa, kp));
            if (fa.sharedSizeBytes != 0) return fail(CDS_ERR_HIP, "sparse_scan16_kernel was built with static LDS: its map would not start at LDS address 0");
            checked.push_back(kp);
        }
    }
    CDS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    // persistent grid: as many workgroups as fit (LDS and the 2048 threads of a CU), every one visits every slab
    int per_cu = (int)std::min<size_t>((size_t)(2048 / threads), ctx->lds_per_block / std::max<size_t>(lds_bytes, 1));
    per_cu = std::max(1, per_cu);
    if (const char *e = getenv("CDS_SPARSE_WGS_PER_CU")) per_cu = std::max(1, atoi(e));
    const int64_t waves = threads / 64;
    const int64_t need = (p.count * p.n_slabs + (int64_t)p.group * waves - 1) / ((int64_t)p.group * waves);
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((int64_t)ctx->n_cus * per_cu, need));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds_bytes, st, p);
    CDS_HIP(hipGetLastError());
    return CDS_OK;
}

static int launch_sparse(cds_context *ctx, const SparseScanParams &p, size_t lds_bytes, int threads, bool xy2, bool mirror,
                         hipStream_t st) {
    if (xy2) return mirror ? launch_sparse_one<true, true>(ctx, p, lds_bytes, threads, st) : launch_sparse_one<true, false>(ctx, p, lds_bytes, threads, st);
    return mirror ? launch_sparse_one<false, true>(ctx, p, lds_bytes, threads, st) : launch_sparse_one<false, false>(ctx, p, lds_bytes, threads, st);
}


static int launch_sparse_multi(cds_context *ctx, const SparseMultiParams &p, int threads, bool xy2, bool mirror, bool plane8, hipStream_t st) {
    const bool pf = p.umap_words > 0;
#define SM_PICK(P8V, PFV) (xy2 ? (mirror ? sparse_multi_kernel<true, true, P8V, PFV> : sparse_multi_kernel<true, false, P8V, PFV>) \
                               : (mirror ? sparse_multi_kernel<false, true, P8V, PFV> : sparse_multi_kernel<false, false, P8V, PFV>))
    auto kern = plane8 ? (pf ? SM_PICK(true, true) : SM_PICK(true, false)) : (pf ? SM_PICK(false, true) : SM_PICK(false, false));
#undef SM_PICK
    const size_t lds_bytes = multi_lds_bytes(threads, p.n_iv, p.umap_words);
    CDS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    int per_cu = (int)std::min<size_t>((size_t)(2048 / threads), ctx->lds_per_block / lds_bytes);
    per_cu = std::max(1, per_cu);
    if (const char *e = getenv("CDS_SPARSE_WGS_PER_CU")) per_cu = std::max(1, atoi(e));
    const int64_t waves = threads / 64;
    const int64_t need = (p.count + (int64_t)p.group * waves - 1) / ((int64_t)p.group * waves);
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((int64_t)ctx->n_cus * per_cu, need));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds_bytes, st, p);
    CDS_HIP(hipGetLastError());
    return CDS_OK;
}

static int sparse_check_range(const cds_sparse_library *sl, int64_t first, int64_t count) {
    if (!sl) return fail(CDS_ERR_INVALID_ARGUMENT, "sparse library is NULL");
    if (first < 0 || count < 0 || first + count > sl->n_images) return fail(CDS_ERR_INVALID_ARGUMENT, "image range outside the sparse library");
    return CDS_OK;
}

// masks [m_begin, m_end) of the set (m_end < 0: all; a range must not cut a multi-mask group, see mask_chunks):
// d_scores[(m - m_begin) * count + t]; d_counts optional [m - m_begin][t][18]
static int sparse_search_impl(const cds_maskset *ms, const cds_sparse_library *sl, int64_t first, int64_t count, int32_t *d_scores,
                              int32_t *d_counts, bool timed, int m_begin = 0, int m_end = -1) {
    if (!ms || !sl) return fail(CDS_ERR_INVALID_ARGUMENT, "NULL argument");
    int rc = sparse_check_range(sl, first, count);
    if (rc) return rc;
    if (sl->W != ms->W || sl->H != ms->H) {
        char buf[200];
        snprintf(buf, sizeof buf, "Invalid image size - target's image size (%d, %d) must match query's image size: (%d, %d)",
                 ms->W, ms->H, sl->W, sl->H);  // PixelMatchColorDepthSearchAlgorithm.java:171-175
        return fail(CDS_ERR_INVALID_ARGUMENT, buf);
    }
    if (ms->params.data_threshold < sl->min_threshold) {
        char buf[240];
        snprintf(buf, sizeof buf, "the sparse library holds no pixel at or below its minimum data threshold %d; a search with data threshold %d "
                 "would miss the pixels in between (cds_sparse_library_set_min_threshold)", sl->min_threshold, ms->params.data_threshold);
        return fail(CDS_ERR_INVALID_ARGUMENT, buf);
    }
    if (sl->ctx != ms->ctx) return fail(CDS_ERR_INVALID_ARGUMENT, "mask set and library belong to different contexts");
    if (count == 0) return CDS_OK;
    if (count >= ((int64_t)1 << 31)) return fail(CDS_ERR_UNSUPPORTED, "more than 2^31 targets in one call");
    for (const SparseMaskPlan &pl : ms->sp_plans)
        if (pl.n_slabs < 0) return fail(CDS_ERR_UNSUPPORTED, "a mask of this set cannot be laid out for the sparse scan (use the dense library)");
    cds_context *ctx = ms->ctx;
    CDS_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    std::lock_guard<std::mutex> lk(ms->sp_mu);
    const int threads = sparse_threads();
    int group = count >= 262144 ? 4 : count >= 4096 ? 2 : 1;  // (grabs are cheap since the counters are partitioned; small grabs balance the tail; 4 per grab at 1 M targets: +2.4 %)
    if (const char *e = getenv("CDS_SPARSE_GROUP")) group = std::max(1, atoi(e));
    group = std::min(group, 64 / sl->nstrips);  // a grab's run headers (targets x visited strips) live in the lanes of one wave
    int nparts = (int)std::max<int64_t>(1, std::min<int64_t>(SP_MAX_PARTS, count / 2048));
    if (const char *e = getenv("CDS_SPARSE_PARTS")) nparts = std::max(1, std::min(SP_MAX_PARTS, atoi(e)));
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int32_t *partial = nullptr;
    auto cleanup = [&]() {
        if (ev0) hipEventDestroy(ev0);
        if (ev1) hipEventDestroy(ev1);
        if (partial) hipFreeAsync(partial, st);
        ev0 = ev1 = nullptr;
        partial = nullptr;
    };
#define SP_HIP(call)                         \
    do {                                     \
        hipError_t _e = (call);              \
        if (_e != hipSuccess) {              \
            cleanup();                       \
            return cds::hip_fail(_e, #call); \
        }                                    \
    } while (0)
    int max_slabs = 0;
    for (const SparseMaskPlan &pl : ms->sp_plans) max_slabs = std::max(max_slabs, pl.n_slabs);
    const bool want_partial = d_counts != nullptr || max_slabs > 1;
    if (want_partial) SP_HIP(hipMallocAsync((void **)&partial, (size_t)std::max(1, max_slabs) * count * CDS_N_TRANSFORMS * sizeof(int32_t), st));
    if (timed) {
        SP_HIP(hipEventCreate(&ev0));
        SP_HIP(hipEventCreate(&ev1));
        SP_HIP(hipEventRecord(ev0, st));
    }
    int launches = 0;
    if (m_end < 0) m_end = ms->n_masks;
    auto scan_one_mask = [&](int m) -> int {
        const SparseMaskPlan &pl = ms->sp_plans[(size_t)m];
        int32_t *out = d_scores + (size_t)(m - m_begin) * count;
        int32_t *cnt = d_counts ? d_counts + (size_t)(m - m_begin) * count * CDS_N_TRANSFORMS : nullptr;
        if (pl.n_slabs == 0) {  // no mask pixel can ever match
            SP_HIP(hipMemsetAsync(out, 0, (size_t)count * sizeof(int32_t), st));
            if (cnt) SP_HIP(hipMemsetAsync(cnt, 0, (size_t)count * CDS_N_TRANSFORMS * sizeof(int32_t), st));
            return CDS_OK;
        }
        if (!ms->sp_uploaded) {   // (first single-mask launch of a set prepared for the multi-mask kernel)
            const int urc = upload_single_plans(ms);
            if (urc) {
                cleanup();
                return urc;
            }
        }
        SparseScanParams p{};
        p.px = sl->d_px; p.img_off = sl->d_img_off; p.rowptr = sl->d_rowptr;
        p.px16 = sl->d_px16;
        p.first = first; p.count = count; p.W = sl->W; p.H = sl->H;
        p.nstrips = sl->nstrips; p.strip_w = sl->strip_w;
        p.slabs = ms->d_sp_slabs + pl.slabs_off;
        p.iv = ms->d_sp_iv + pl.iv_off;
        p.n_iv = pl.n_iv; p.iv_in_lds = pl.iv_in_lds ? 1 : 0; p.plane8 = pl.plane8 ? 1 : 0;
        p.xlo = pl.xlo; p.cols = pl.cols; p.ylo = pl.ylo; p.nrows = pl.nrows; p.pitch = pl.pitch; p.nwpr = pl.nwpr;
        p.slab_rows = pl.slab_rows; p.n_slabs = pl.n_slabs; p.near_off = pl.near_off; p.slab_bytes = pl.slab_bytes;
        p.ux0 = pl.ux0; p.g64 = pl.g64; p.rowb64 = pl.rowb64; p.map64_bytes = pl.map64_bytes;
        p.thr = std::max(0, ms->params.data_threshold);  // a pixel whose maximum is 0 is black and has no hue class anyway
        p.group = group;
        p.nparts = nparts;
        p.psize = (uint32_t)((count + nparts - 1) / nparts);
        p.work = ms->d_sp_work;
        p.out = out;
        const bool use_partial = cnt != nullptr || pl.n_slabs > 1;
        p.partial = use_partial ? partial : nullptr;
        // Which kernel.  A library with the 16-bit stage-1 stream can be scanned through it (half the bytes per visited record; a round
        // then gathers its 64 records) or through the 4-byte records as before.  The stream wins when few visited records reach a round
        // (a library whose colours are unrelated to the mask's: 1.5 % on the synthetic one, or a mask much narrower than the strips) and
        // loses when many do (the reference's LM images against the EM masks that overlap them: a third to a half).  That share is a
        // property of the (mask, library) pair, so the first large single-slab search MEASURES it on a pilot of its first targets
        // (p.stat) and the choice is remembered for later searches of the same library.  CDS_SPARSE_SCAN16=1 / CDS_SPARSE_SCAN32=1
        // force a kernel (tests, A/B runs).  Small searches and multi-slab masks take the 4-byte records.
        int use16 = 0;
        int64_t pilot = 0;
        if (stream16_possible(p)) {
            if (getenv("CDS_SPARSE_SCAN32")) use16 = 0;
            else if (getenv("CDS_SPARSE_SCAN16")) use16 = 1;
            else {
                if (ms->s16_lib != (const void *)sl || ms->s16_lib_images != sl->n_images || ms->s16_choice.size() != (size_t)ms->n_masks) {
                    ms->s16_choice.assign((size_t)ms->n_masks, (int8_t)-1);
                    ms->s16_lib = (const void *)sl;
                    ms->s16_lib_images = sl->n_images;
                }
                const int known = ms->s16_choice[(size_t)m];
                if (known >= 0) use16 = known;
                else if (!use_partial && count >= 16384 && !ctx->sparse_load_only.load()) pilot = 4096;
            }
        }
        if (pilot > 0) {
            if (!ms->d_s16_stat) SP_HIP(cache_alloc(ctx, (void **)&ms->d_s16_stat, 2 * sizeof(unsigned long long)));
            SP_HIP(hipMemsetAsync(ms->d_s16_stat, 0, 2 * sizeof(unsigned long long), st));
            SparseScanParams q = p;
            q.count = pilot;
            q.nparts = 1;
            q.psize = (uint32_t)pilot;
            q.group = 1;
            q.use16 = 1;
            q.stat = ms->d_s16_stat;
            SP_HIP(hipMemsetAsync(ms->d_sp_work, 0, (size_t)pl.n_slabs * 32 * sizeof(uint32_t), st));
            q.plane_global = 1;
            const int prc = launch_sparse(ctx, q, pl.lds_bytes - (size_t)pl.slab_bytes + (size_t)pl.map64_bytes, threads, ms->params.xy_shift == 2, ms->params.mirror != 0, st);
            if (prc) {
                cleanup();
                return prc;
            }
            unsigned long long stat[2] = {0, 0};
            SP_HIP(hipMemcpyAsync(stat, ms->d_s16_stat, sizeof stat, hipMemcpyDeviceToHost, st));
            SP_HIP(hipStreamSynchronize(st));
            const double ratio = stat[0] ? (double)stat[1] / ((double)stat[0] * (ms->params.mirror ? 2 : 1)) : 0.0;
            double limit = 0.05;
            if (const char *e = getenv("CDS_SPARSE_PILOT_LIMIT")) limit = atof(e);
            use16 = ratio <= limit ? 1 : 0;
            ms->s16_choice[(size_t)m] = (int8_t)use16;
            ms->last_pilot_ratio = ratio;
            launches++;
            // the rest of the range with the chosen kernel (the pilot's scores stand: both kernels give the same)
            p.first = first + pilot;
            p.count = count - pilot;
            p.out = out + pilot;
            p.psize = (uint32_t)((p.count + nparts - 1) / nparts);
        }
        p.use16 = use16;
        p.stat = nullptr;
        ms->last_scan_kernel = use16;
        size_t lds_launch = pl.lds_bytes;
        // the stream kernel runs where rounds are rare: its plane stays in L2 and two workgroups share a CU (+4 % on the 1 M-target scan;
        // CDS_SPARSE_PLANE_LDS=1: the plane in LDS, one workgroup per CU, as the 4-byte-record kernel has it)
        if (use16) {   // LDS of the stream kernel: interval table, queues, its own map -- and the plane unless that stays in L2
            const size_t fixed16 = pl.lds_bytes - (size_t)pl.slab_bytes + (size_t)pl.map64_bytes;
            p.plane_global = getenv("CDS_SPARSE_PLANE_LDS") && fixed16 + (size_t)pl.near_off <= ctx->lds_per_block ? 0 : 1;
            lds_launch = fixed16 + (p.plane_global ? 0 : (size_t)pl.near_off);
        }
        SP_HIP(hipMemsetAsync(ms->d_sp_work, 0, (size_t)pl.n_slabs * nparts * 32 * sizeof(uint32_t), st));
        const int lrc = launch_sparse(ctx, p, lds_launch, threads, ms->params.xy_shift == 2, ms->params.mirror != 0, st);
        if (lrc) {
            cleanup();
            return lrc;
        }
        if (use_partial) {
            hipLaunchKernelGGL(sparse_finalize_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, partial, pl.n_slabs, count, out, cnt);
            SP_HIP(hipGetLastError());
        }
        launches++;
        return CDS_OK;
    };
    // groups of masks share one pass (sparse_multi_kernel) when the set was laid out for it and only scores are wanted
    const char *multi_env = getenv("CDS_SPARSE_MULTI");
    const bool use_multi = !d_counts && !ms->sg_plans.empty() && threads == 1024 && !(multi_env && atoi(multi_env) == 0);
    if (!use_multi) {
        for (int m = m_begin; m < m_end; m++) {
            rc = scan_one_mask(m);
            if (rc) return rc;
        }
    } else {
        for (const SparseGroupPlan &gp : ms->sg_plans) {
            if (gp.first_mask < m_begin || gp.first_mask + gp.n_masks > m_end) continue;
            if (!gp.ok) {
                for (int m = gp.first_mask; m < gp.first_mask + gp.n_masks; m++) {
                    rc = scan_one_mask(m);
                    if (rc) return rc;
                }
                continue;
            }
            SparseMultiParams p{};
            p.px = sl->d_px; p.img_off = sl->d_img_off; p.rowptr = sl->d_rowptr;
            p.first = first; p.count = count; p.W = sl->W; p.H = sl->H;
            p.nstrips = sl->nstrips; p.strip_w = sl->strip_w;
            p.planes = ms->d_sg_planes + gp.planes_off;
            p.tmap = ms->d_sg_tmap + gp.tmap_off;
            p.tmap_bytes = (int)gp.tmap_bytes;
            p.umap = reinterpret_cast<const uint32_t *>(ms->d_sg_tmap + gp.tmap_off + gp.umap_off);
            p.umap_words = gp.umap_words;
            p.iv = ms->d_sg_iv + gp.iv_off;
            p.n_masks = gp.n_masks; p.plane_cells = gp.plane_cells;
            p.n_iv = gp.n_iv;
            for (int j = 0; j < kSparseGroupMasks; j++) p.iv_base[j] = gp.iv_base[j];
            p.xlo = gp.xlo; p.cols = gp.cols; p.ylo = gp.ylo; p.nrows = gp.nrows; p.pitch = gp.pitch; p.nwpr = gp.nwpr;
            p.thr = std::max(0, ms->params.data_threshold);
            p.group = std::min(group, 2); p.nparts = nparts;   // (a grab of the multi-mask kernel is 8 masks' worth of work already)
            p.psize = (uint32_t)((count + nparts - 1) / nparts);
            p.work = ms->d_sp_work;
            p.out = d_scores + (size_t)(gp.first_mask - m_begin) * count;
            p.out_stride = count;
            SP_HIP(hipMemsetAsync(ms->d_sp_work, 0, (size_t)nparts * 32 * sizeof(uint32_t), st));
            rc = launch_sparse_multi(ctx, p, threads, ms->params.xy_shift == 2, ms->params.mirror != 0, gp.plane8, st);
            if (rc) {
                cleanup();
                return rc;
            }
            launches++;
        }
    }
    if (launches > 0) {   // (free_sparse_plans waits for this before the set's blocks go back to the context's cache)
        if (!ms->sp_last_use) SP_HIP(hipEventCreateWithFlags(&ms->sp_last_use, hipEventDisableTiming));
        SP_HIP(hipEventRecord(ms->sp_last_use, st));
    }
    if (timed) {
        SP_HIP(hipEventRecord(ev1, st));
        SP_HIP(hipEventSynchronize(ev1));
        float t = 0;
        SP_HIP(hipEventElapsedTime(&t, ev0, ev1));
        std::lock_guard<std::mutex> lk2(ms->mu);
        ms->last_ms = launches ? t / launches : 0;
        ms->last_launches = launches;
    }
    cleanup();
#undef SP_HIP
    return CDS_OK;
}


// =================================================================================================
//            match list: ColorMIPSearch.isMatch + stream compaction on the device
// =================================================================================================
// The reference keeps a (mask, target) pair when ColorMIPSearch.isMatch holds (API/cds/ColorMIPSearch.java:42-45: score > 0
// and (float) score / querySize > pctPositivePixels / 100) and drops the rest (tools cmd/cdsprocess/
// AbstractColorMIPSearchProcessor.java:60-85).  A 1,024-mask x 1 M-target search produces 4 GB of scores of which a handful
// per mask pass; here the filter and the compaction run where the scores are, and only the match list crosses PCIe.
// isMatch is monotone in the score, so the host finds, per mask, the smallest passing score with the reference's own
// arithmetic (cds_is_match) and the device test is an integer compare.
struct MatchParams {
    const int32_t *scores;     // [n_masks][count] score | mirrored << 31
    const int32_t *min_score;  // [n_masks] smallest score that is a match (INT32_MAX: none)
    int n_masks, mask0;
    int64_t count;
    int nblk;                  // blocks of 1024 targets per mask
    uint32_t *blockcnt;        // [n_masks * nblk] matches per block -> exclusive offsets inside the chunk
    unsigned long long *total; // [0] matches before this chunk, [1] matches of this chunk
    cds_match *out;
    int64_t capacity;
};

__device__ __forceinline__ int match_flags(const MatchParams &p, int m, int64_t t0, int32_t (&v)[4]) {
    const int32_t thr = p.min_score[m];
    int f = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int64_t t = t0 + k;
        v[k] = t < p.count ? p.scores[(size_t)m * p.count + t] : 0;
        if (t < p.count && (v[k] & 0x7fffffff) >= thr) f |= 1 << k;
    }
    return f;
}

__global__ __launch_bounds__(256) void match_count_kernel(MatchParams p) {
    __shared__ int wsum[4];
    const int m = blockIdx.x / p.nblk, b = blockIdx.x - m * p.nblk;
    int32_t v[4];
    const int f = match_flags(p, m, (int64_t)b * 1024 + threadIdx.x * 4, v);
    int c = __popc(f);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) p.blockcnt[blockIdx.x] = (uint32_t)(wsum[0] + wsum[1] + wsum[2] + wsum[3]);
}

// one workgroup: blockcnt[0..n) -> exclusive offsets in place, chunk total -> total[1]
__global__ __launch_bounds__(1024) void match_scan_kernel(MatchParams p) {
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t carry_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t n = (int64_t)p.n_masks * p.nblk;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int64_t base = 0; base < n; base += 1024) {
        const int64_t i = base + threadIdx.x;
        const uint32_t v = i < n ? p.blockcnt[i] : 0u;
        const uint32_t inc = wave_incl_scan(v, lane);
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        uint32_t off = carry_s;
        for (int w = 0; w < wave; w++) off += wsum[w];
        __syncthreads();
        if (i < n) p.blockcnt[i] = off + inc - v;
        if (threadIdx.x == 1023) carry_s = off + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) p.total[1] = carry_s;
}

__global__ __launch_bounds__(256) void match_fill_kernel(MatchParams p) {
    __shared__ int wsum[4];
    const int m = blockIdx.x / p.nblk, b = blockIdx.x - m * p.nblk;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t t0 = (int64_t)b * 1024 + threadIdx.x * 4;
    int32_t v[4];
    const int f = match_flags(p, m, t0, v);
    const int c = __popc(f);
    int inc = c;  // inclusive scan over the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int u = __shfl_up(inc, o, 64);
        if (lane >= o) inc += u;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int before = inc - c;
    for (int w = 0; w < wave; w++) before += wsum[w];
    unsigned long long pos = p.total[0] + p.blockcnt[blockIdx.x] + (unsigned long long)before;
#pragma unroll
    for (int k = 0; k < 4; k++)
        if (f & (1 << k)) {
            if (pos < (unsigned long long)p.capacity) {
                cds_match r;
                r.mask = p.mask0 + m;
                r.target = (int32_t)(t0 + k);
                r.score = v[k];
                p.out[pos] = r;
            }
            pos++;
        }
}

__global__ void match_advance_kernel(unsigned long long *total) { total[0] += total[1]; }


// =================================================================================================
//            per-mask top-k on the device (the "top-500 hits per mask" of BASELINE configs[4])
// =================================================================================================
// One workgroup per mask: (1) histogram of the scores (4,096 bins of score >> shift; scores never exceed the query size) ->
// the bin the k-th best score falls in, (2) histogram of that bin's low bits -> the exact threshold score T and how many
// entries with score == T are still needed, (3) an ordered pass that writes the entries with score > T, and the first needed
// ones with score == T, in target order, (4) a bitonic sort of the k entries by (score descending, target ascending).
struct TopkParams {
    const int32_t *scores;   // [n_masks][count] score | mirrored << 31
    const int32_t *qsize;    // [n_masks] query sizes (upper bound of a mask's scores)
    int64_t count;
    int k, kp;               // kp = k rounded up to a power of two (<= 1024)
    int32_t *out_target;     // [n_masks][k]  (-1 behind the last entry when count < k)
    int32_t *out_score;      // [n_masks][k]  packed
};

__global__ __launch_bounds__(1024) void topk_kernel(TopkParams p) {
    __shared__ uint32_t hist[4096];
    __shared__ unsigned long long key[1024];
    __shared__ uint32_t val[1024];
    __shared__ uint32_t wsum[2][16];
    __shared__ int sh_bin, sh_above, sh_T, sh_need, sh_baseA, sh_baseE;
    const int m = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int32_t *sc = p.scores + (size_t)m * p.count;
    const int k = (int)min((int64_t)p.k, p.count);
    int shift = 0;
    while (((int64_t)p.qsize[m] >> shift) >= 4096) shift++;
    // (1) coarse histogram
    for (int i = tid; i < 4096; i += 1024) hist[i] = 0;
    __syncthreads();
    for (int64_t t = tid; t < p.count; t += 1024) atomicAdd(&hist[min((sc[t] & 0x7fffffff) >> shift, 4095)], 1u);
    __syncthreads();
    if (tid == 0) {
        int above = 0, b = 4095;
        while (b > 0 && above + (int)hist[b] < k) above += (int)hist[b--];
        sh_bin = b, sh_above = above;
    }
    __syncthreads();
    const int tb = sh_bin;
    // (2) exact threshold inside the bin
    if (shift > 0) {
        const int nfine = 1 << shift;
        __syncthreads();
        for (int i = tid; i < nfine; i += 1024) hist[i] = 0;
        __syncthreads();
        for (int64_t t = tid; t < p.count; t += 1024) {
            const int v = sc[t] & 0x7fffffff;
            if ((v >> shift) == tb) atomicAdd(&hist[v & (nfine - 1)], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            int above = sh_above, f = nfine - 1;
            while (f > 0 && above + (int)hist[f] < k) above += (int)hist[f--];
            sh_T = (tb << shift) | f, sh_above = above;
        }
    } else if (tid == 0) {
        sh_T = tb;
    }
    __syncthreads();
    if (tid == 0) sh_need = k - sh_above, sh_baseA = 0, sh_baseE = 0;
    for (int i = tid; i < 1024; i += 1024) key[i] = ~0ull, val[i] = 0;
    __syncthreads();
    const int T = sh_T, above = sh_above, need = sh_need;
    // (3) ordered pass: entries above T to [0, above), the first `need` entries equal to T to [above, k)
    for (int64_t base = 0; base < p.count; base += 1024) {
        const int64_t t = base + tid;
        const int32_t pk = t < p.count ? sc[t] : 0;
        const int v = pk & 0x7fffffff;
        const bool isA = t < p.count && v > T, isE = t < p.count && v == T;
        const unsigned long long ba = __ballot(isA), be = __ballot(isE);
        const int ra = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(ba >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ba, 0));
        const int re = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(be >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)be, 0));
        if (lane == 0) wsum[0][wave] = (uint32_t)__popcll(ba), wsum[1][wave] = (uint32_t)__popcll(be);
        __syncthreads();
        int offA = sh_baseA, offE = sh_baseE;
        for (int w = 0; w < wave; w++) offA += (int)wsum[0][w], offE += (int)wsum[1][w];
        int pos = -1;
        if (isA) pos = offA + ra;
        else if (isE && offE + re < need) pos = above + offE + re;
        if (pos >= 0 && pos < 1024) {
            key[pos] = ((unsigned long long)(0x7fffffffu - (uint32_t)v) << 32) | (uint32_t)t;
            val[pos] = (uint32_t)pk;
        }
        __syncthreads();
        if (tid == 0) {
            int a = 0, e = 0;
            for (int w = 0; w < 16; w++) a += (int)wsum[0][w], e += (int)wsum[1][w];
            sh_baseA += a, sh_baseE += e;
        }
        __syncthreads();
        if (sh_baseA >= above && sh_baseE >= need) break;   // (uniform: everything is placed)
    }
    __syncthreads();
    // (4) bitonic sort of the kp entries by (score descending, target ascending); the padding sorts to the end
    for (int size = 2; size <= p.kp; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            if (tid < p.kp) {
                const int partner = tid ^ stride;
                if (partner > tid) {
                    const bool up = (tid & size) == 0;
                    const unsigned long long a = key[tid], b = key[partner];
                    if ((a > b) == up) {
                        key[tid] = b, key[partner] = a;
                        const uint32_t va = val[tid];
                        val[tid] = val[partner], val[partner] = va;
                    }
                }
            }
            __syncthreads();
        }
    if (tid < p.k) {
        const bool have = tid < k;
        p.out_target[(size_t)m * p.k + tid] = have ? (int32_t)(uint32_t)key[tid] : -1;
        p.out_score[(size_t)m * p.k + tid] = have ? (int32_t)val[tid] : 0;
    }
}

}  // namespace cds

using namespace cds;

// The 16-bit stream is worth its +50 % of record memory only while records AND stream stay resident: by policy it is left out when
// the two together would take more than CDS_SPARSE_STREAM16_MAX_FRACTION (default 0.6) of the device's memory -- a million real LM
// MIPs (36 K records each) are 144 GB of records; their stream would leave no room for anything else, and the pilot routes such
// libraries to the 4-byte records anyway.
static bool px16_fits(int64_t capacity_pixels) {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || total_b == 0) return true;
    double frac = 0.6;
    if (const char *e = getenv("CDS_SPARSE_STREAM16_MAX_FRACTION")) frac = atof(e);
    return (double)capacity_pixels * 6.0 <= frac * (double)total_b;
}

// Enlarge a growable library's arrays to hold at least need_images images and need_px records: new arrays, device-to-device copies
// of what is held, the old ones freed after the copies.  The stream is derived data: when its larger copy does not fit (or the policy
// above says no) the library goes on without it.  On failure the library is what it was.
static int sparse_grow(cds_sparse_library *sl, int64_t need_images, int64_t need_px) {
    cds_context *ctx = sl->ctx;
    hipStream_t st = ctx->stream;
    auto grown = [](int64_t cap, int64_t need) { return std::max(need, cap + cap / 2 + 1); };
    if (need_images > sl->cap_images) {
        const int64_t cap = grown(sl->cap_images, need_images);
        const size_t per = (size_t)sl->nstrips * (sl->H + 1);
        uint64_t *off = nullptr;
        uint32_t *rp = nullptr;
        hipError_t e = dev_malloc(&off, (size_t)(cap + 1) * sizeof(uint64_t));
        if (e == hipSuccess) e = dev_malloc(&rp, (size_t)cap * per * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMemcpyAsync(off, sl->d_img_off, (size_t)(sl->n_images + 1) * sizeof(uint64_t), hipMemcpyDeviceToDevice, st);
        if (e == hipSuccess && sl->n_images > 0)
            e = hipMemcpyAsync(rp, sl->d_rowptr, (size_t)sl->n_images * per * sizeof(uint32_t), hipMemcpyDeviceToDevice, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) {
            if (off) hipFree(off);
            if (rp) hipFree(rp);
            return hip_fail(e, "growing the sparse library (images)");
        }
        hipFree(sl->d_img_off);
        hipFree(sl->d_rowptr);
        sl->d_img_off = off, sl->d_rowptr = rp, sl->cap_images = cap;
    }
    if (need_px > sl->cap_px) {
        const int64_t cap = grown(sl->cap_px, need_px);
        uint32_t *px = nullptr;
        hipError_t e = dev_malloc(&px, (size_t)cap * sizeof(uint32_t));
        if (e == hipSuccess && sl->n_px > 0) e = hipMemcpyAsync(px, sl->d_px, (size_t)sl->n_px * sizeof(uint32_t), hipMemcpyDeviceToDevice, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) {
            if (px) hipFree(px);
            return hip_fail(e, "growing the sparse library (records)");
        }
        hipFree(sl->d_px);
        sl->d_px = px;
        if (sl->d_px16) {   // the stream follows, or goes
            uint16_t *p16 = nullptr;
            hipError_t e16 = px16_fits(cap) ? dev_malloc(&p16, ((size_t)cap + 2) * sizeof(uint16_t)) : hipErrorOutOfMemory;
            if (e16 == hipSuccess) e16 = hipMemcpyAsync(p16, sl->d_px16, ((size_t)sl->n_px + 1) * sizeof(uint16_t), hipMemcpyDeviceToDevice, st);   // (+ the sentinel)
            if (e16 == hipSuccess) e16 = hipStreamSynchronize(st);
            if (e16 != hipSuccess) {
                (void)hipGetLastError();
                if (p16) hipFree(p16);
                p16 = nullptr;
            }
            hipFree(sl->d_px16);
            sl->d_px16 = p16;
        }
        sl->cap_px = cap;
    }
    return CDS_OK;
}

// =================================================================================================
//                                            C ABI
// =================================================================================================
extern "C" {

int cds_sparse_library_set_growable(cds_sparse_library *sl, int32_t on) {
    if (!sl) return fail(CDS_ERR_INVALID_ARGUMENT, "sparse library is NULL");
    std::lock_guard<std::mutex> lk(sl->mu);
    sl->growable = on != 0;
    return CDS_OK;
}

int cds_sparse_library_reserve(cds_sparse_library *sl, int64_t capacity_images, int64_t capacity_pixels) {
    if (!sl) return fail(CDS_ERR_INVALID_ARGUMENT, "sparse library is NULL");
    std::lock_guard<std::mutex> lk(sl->mu);
    CDS_HIP(hipSetDevice(sl->ctx->device));
    return sparse_grow(sl, capacity_images, capacity_pixels);
}

int cds_sparse_library_capacity(const cds_sparse_library *sl, int64_t *capacity_images, int64_t *capacity_pixels, int32_t *has_stream16) {
    if (!sl) return fail(CDS_ERR_INVALID_ARGUMENT, "sparse library is NULL");
    if (capacity_images) *capacity_images = sl->cap_images;
    if (capacity_pixels) *capacity_pixels = sl->cap_px;
    if (has_stream16) *has_stream16 = sl->d_px16 ? 1 : 0;
    return CDS_OK;
}

int cds_sparse_library_create(cds_context *ctx, int32_t W, int32_t H, int64_t capacity_images, int64_t capacity_pixels,
                              cds_sparse_library **out) {
    if (!ctx || !out) return fail(CDS_ERR_INVALID_ARGUMENT, "NULL argument");
    *out = nullptr;
    if (!image_size_ok(W, H)) return fail(CDS_ERR_UNSUPPORTED, kImageSizeMessage);
    if (capacity_images <= 0 || capacity_pixels <= 0) return fail(CDS_ERR_INVALID_ARGUMENT, "capacities must be positive");
    CDS_HIP(hipSetDevice(ctx->device));
    cds_sparse_library *sl = new cds_sparse_library();
    sl->ctx = ctx; sl->W = W; sl->H = H; sl->cap_images = capacity_images; sl->cap_px = capacity_pixels;
    // up to 4 column strips, each a multiple of 4 pixels and at least 64 wide (a scan reads only the strips its mask's
    // column window -- and the window's mirror image -- can touch)
    if (const char *e = getenv("CDS_SPARSE_DENSITY_LIMIT")) {   // (default of cds_sparse_library_set_density_limit; the test suite lifts it)
        const double v = atof(e);
        if (v > 0) sl->density_limit = v;
    }
    sl->strip_w = std::max(64, round_up((W + 3) / 4, 4));
    if (const char *e = getenv("CDS_SPARSE_STRIP_W")) sl->strip_w = std::max(4, round_up(atoi(e), 4));
    sl->nstrips = (W + sl->strip_w - 1) / sl->strip_w;
    if (sl->nstrips > 16 || sl->strip_w > kMaxStripW) {
        delete sl;
        return fail(CDS_ERR_INVALID_ARGUMENT, "CDS_SPARSE_STRIP_W must give at most 16 strips of at most 512 pixels (a record holds its column relative to its strip in 9 bits)");
    }
    hipError_t e = dev_malloc(&sl->d_px, (size_t)capacity_pixels * sizeof(uint32_t));
    if (e == hipSuccess) e = dev_malloc(&sl->d_img_off, (size_t)(capacity_images + 1) * sizeof(uint64_t));
    if (e == hipSuccess) e = dev_malloc(&sl->d_rowptr, (size_t)capacity_images * sl->nstrips * (H + 1) * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemsetAsync(sl->d_img_off, 0, sizeof(uint64_t), ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess) {
        // the 16-bit stage-1 stream (+50 % of record memory): optional -- without it the scan reads the 4-byte records as before
        const char *env = getenv("CDS_SPARSE_STREAM16");
        sl->want_px16 = !(env && atoi(env) == 0);
        if (sl->want_px16 && px16_fits(capacity_pixels)) {
            if (dev_malloc(&sl->d_px16, ((size_t)capacity_pixels + 2) * sizeof(uint16_t)) != hipSuccess) {
                sl->d_px16 = nullptr;
                (void)hipGetLastError();   // (not an error of this call: the library works without the stream)
            }
        }
    }
    if (e != hipSuccess) {
        if (sl->d_px) hipFree(sl->d_px);
        if (sl->d_img_off) hipFree(sl->d_img_off);
        if (sl->d_rowptr) hipFree(sl->d_rowptr);
        delete sl;
        return hip_fail(e, "hipMalloc(sparse library)");
    }
    *out = sl;
    return CDS_OK;
}

int cds_sparse_library_dims(const cds_sparse_library *sl, int32_t *W, int32_t *H) {
    if (!sl) return fail(CDS_ERR_INVALID_ARGUMENT, "NULL argument");
    if (W) *W = sl->W;
    if (H) *H = sl->H;
    return CDS_OK;
}

void cds_sparse_library_destroy(cds_sparse_library *sl) {
    if (!sl) return;
    hipSetDevice(sl->ctx->device);
    hipStreamSynchronize(sl->ctx->stream);
    if (sl->stage) cds_library_destroy(sl->stage);
    if (sl->d_px16) hipFree(sl->d_px16);
    if (sl->d_px) hipFree(sl->d_px);
    if (sl->d_img_off) hipFree(sl->d_img_off);
    if (sl->d_rowptr) hipFree(sl->d_rowptr);
    if (sl->d_tmp) hipFree(sl->d_tmp);
    delete sl;
}

int cds_sparse_library_layout(const cds_sparse_library *sl, int32_t *n_strips, int32_t *strip_width) {
    if (!sl) return fail(CDS_ERR_INVALID_ARGUMENT, "sparse library is NULL");
    if (n_strips) *n_strips = sl->nstrips;
    if (strip_width) *strip_width = sl->strip_w;
    return CDS_OK;
}

int cds_sparse_library_set_density_limit(cds_sparse_library *sl, double max_signal_fraction) {
    if (!sl) return fail(CDS_ERR_INVALID_ARGUMENT, "sparse library is NULL");
    if (!(max_signal_fraction > 0)) return fail(CDS_ERR_INVALID_ARGUMENT, "the density limit must be positive (>= 1.1 accepts everything)");
    std::lock_guard<std::mutex> lk(sl->mu);
    sl->density_limit = max_signal_fraction;
    return CDS_OK;
}

int cds_sparse_library_set_min_threshold(cds_sparse_library *sl, int32_t min_data_threshold) {
    if (!sl) return fail(CDS_ERR_INVALID_ARGUMENT, "sparse library is NULL");
    if (min_data_threshold < 0 || min_data_threshold > 254) return fail(CDS_ERR_INVALID_ARGUMENT, "the minimum data threshold must be in 0..254");
    std::lock_guard<std::mutex> lk(sl->mu);
    if (sl->n_images > 0 && min_data_threshold < sl->min_threshold)
        return fail(CDS_ERR_INVALID_ARGUMENT, "the minimum data threshold of a library that holds images can only be raised (its records below the old one are gone)");
    sl->min_threshold = min_data_threshold;
    return CDS_OK;
}

int cds_sparse_library_min_threshold(const cds_sparse_library *sl, int32_t *min_data_threshold) {
    if (!sl || !min_data_threshold) return fail(CDS_ERR_INVALID_ARGUMENT, "NULL argument");
    *min_data_threshold = sl->min_threshold;
    return CDS_OK;
}

int64_t cds_sparse_library_size(const cds_sparse_library *sl) { return sl ? sl->n_images : 0; }
int64_t cds_sparse_library_pixels(const cds_sparse_library *sl) { return sl ? sl->n_px : 0; }

int cds_sparse_library_clear(cds_sparse_library *sl) {
    if (!sl) return fail(CDS_ERR_INVALID_ARGUMENT, "sparse library is NULL");
    std::lock_guard<std::mutex> lk(sl->mu);
    sl->n_images = 0;
    sl->n_px = 0;
    return CDS_OK;
}

// The sparse scan's cost grows with the records it visits: above ~1/4 of the frame (183 K records at 1210 x 566) the dense
// layout's kernel is the faster one (measured: 2.1 M comparisons/s against 6.0 M/s on dense-uniform images, 254 against 12 on
// 2.6 %-dense synthetic ones).  Appends whose images are denser than the library's limit are refused with this message instead
// of silently running slower than the dense scan; cds_sparse_library_set_density_limit raises or removes the limit.
static thread_local bool g_density_refusal = false;   // set by the refusal below, taken by cds_sparse_library_append_files' roll-back
}  // extern "C"
bool cds::take_density_refusal() {
    const bool v = g_density_refusal;
    g_density_refusal = false;
    return v;
}
extern "C" {
static int density_refused(double density, double limit) {
    g_density_refusal = true;
    char buf[320];
    snprintf(buf, sizeof buf, "images are %.1f %% signal pixels, above this sparse library's density limit of %.1f %%: the dense-layout scan "
             "(cds_library + cds_search) is faster there -- keep them in a cds_library, or raise the limit with cds_sparse_library_set_density_limit",
             100.0 * density, 100.0 * limit);
    return fail(CDS_ERR_UNSUPPORTED, buf);
}

// the derived 16-bit stream of records [first, first + count) (+ the sentinel behind them), on the library's stream
static int refresh_px16(cds_sparse_library *sl, int64_t first, int64_t count) {
    if (!sl->d_px16 || count <= 0) return CDS_OK;
    const unsigned blocks = (unsigned)std::min<int64_t>((count + 1 + 255) / 256, (int64_t)sl->ctx->n_cus * 32);
    hipLaunchKernelGGL(sparse_px16_kernel, dim3(blocks), dim3(256), 0, sl->ctx->stream, sl->d_px, sl->d_px16, (uint64_t)first, (uint64_t)count);
    CDS_HIP(hipGetLastError());
    return CDS_OK;
}

// One APPEND CALL is a transaction: the density limit is tested on the call's average (AppendTally sums its sub-batches; a
// sub-batch is refused early only when the records seen so far already put the whole call above the limit), and a call that fails
// -- density, capacity, a HIP error -- puts n_images / n_px back to their values at entry (the records written behind the
// logical end are simply overwritten by the next append), so "nothing is added" holds for the call, not per sub-batch.
struct AppendTally {
    int64_t images_total = 0;   // images of the whole call
    int64_t images = 0;         // ... converted so far
    double records = 0;         // ... and their records
    int64_t n_images0 = 0, n_px0 = 0;   // the library at entry
};
static void append_begin(const cds_sparse_library *sl, int64_t images_total, AppendTally &t) {
    t = AppendTally();
    t.images_total = images_total;
    t.n_images0 = sl->n_images;
    t.n_px0 = sl->n_px;
}
static int append_end(cds_sparse_library *sl, const AppendTally &t, int rc) {
    if (rc == CDS_OK && t.images > 0) {
        const double density = t.records / ((double)t.images * sl->W * sl->H);
        if (density > sl->density_limit) rc = density_refused(density, sl->density_limit);
    }
    if (rc != CDS_OK) {
        sl->n_images = t.n_images0;
        sl->n_px = t.n_px0;
    }
    return rc;
}

static int sparse_append_locked(cds_sparse_library *sl, const cds_library *src, int64_t first, int64_t count, AppendTally &tally) {
    cds_context *ctx = sl->ctx;
    hipStream_t st = ctx->stream;
    // one launch covers at most 2^32 / 64 image rows (one wave each)
    const int64_t max_batch = std::max<int64_t>(1, (int64_t)(0xffffffffu / 256u) * 4 / sl->H);
    if (sl->tmp_images < std::min(count, max_batch)) {
        if (sl->d_tmp) hipFree(sl->d_tmp);
        sl->d_tmp = nullptr;
        sl->tmp_images = 0;
        const int64_t n = std::min(count, max_batch);
        CDS_HIP(dev_malloc(&sl->d_tmp, (size_t)n * sizeof(uint32_t) + sizeof(uint64_t)));
        sl->tmp_images = n;
    }
    for (int64_t done = 0; done < count;) {
        const int64_t n = std::min(max_batch, count - done);
        BuildParams p{};
        p.rgb = src->d_rgb + (size_t)(first + done) * src->img_stride;
        p.img_stride = src->img_stride; p.pitch = src->pitch; p.W = src->W; p.H = src->H;
        p.nstrips = sl->nstrips; p.strip_w = sl->strip_w;
        p.count = n; p.dst_first = sl->n_images;
        p.px = sl->d_px; p.img_off = sl->d_img_off; p.rowptr = sl->d_rowptr;
        p.tmp = sl->d_tmp;
        p.total = reinterpret_cast<uint64_t *>(sl->d_tmp + sl->tmp_images);
        p.min_thr = (uint32_t)sl->min_threshold;
        const unsigned row_blocks = (unsigned)((n * sl->H + 3) / 4);
        hipLaunchKernelGGL(sparse_rows_kernel<false>, dim3(row_blocks), dim3(256), 0, st, p);
        CDS_HIP(hipGetLastError());
        hipLaunchKernelGGL(sparse_rowscan_kernel, dim3((unsigned)n), dim3(256), 0, st, p);
        CDS_HIP(hipGetLastError());
        hipLaunchKernelGGL(sparse_imgscan_kernel, dim3(1), dim3(1024), 0, st, p);
        CDS_HIP(hipGetLastError());
        uint64_t total = 0;
        CDS_HIP(hipMemcpyAsync(&total, p.total, sizeof total, hipMemcpyDeviceToHost, st));
        CDS_HIP(hipStreamSynchronize(st));
        tally.images += n;
        tally.records += (double)total;
        {   // already certain: even if every remaining image were black, the call's average would be above the limit
            const double floor_density = tally.records / ((double)std::max(tally.images_total, tally.images) * sl->W * sl->H);
            if (floor_density > sl->density_limit) return density_refused(tally.records / ((double)tally.images * sl->W * sl->H), sl->density_limit);
        }
        if (sl->n_px + (int64_t)total > sl->cap_px && sl->growable) {
            const int rcg = sparse_grow(sl, 0, sl->n_px + (int64_t)total);
            if (rcg) return rcg;
            p.px = sl->d_px;   // (the fill below writes into the new arena)
        }
        if (sl->n_px + (int64_t)total > sl->cap_px) {
            char buf[200];
            snprintf(buf, sizeof buf, "sparse library is full: %lld records held, %lld more needed, capacity %lld", (long long)sl->n_px,
                     (long long)total, (long long)sl->cap_px);
            return fail(CDS_ERR_OUT_OF_MEMORY, buf);
        }
        hipLaunchKernelGGL(sparse_rows_kernel<true>, dim3(row_blocks), dim3(256), 0, st, p);
        CDS_HIP(hipGetLastError());
        {
            const int rc16 = refresh_px16(sl, sl->n_px, (int64_t)total);
            if (rc16) return rc16;
        }
        sl->n_images += n;
        sl->n_px += (int64_t)total;
        done += n;
    }
    CDS_HIP(hipStreamSynchronize(st));
    return CDS_OK;
}

int cds_sparse_library_append(cds_sparse_library *sl, const cds_library *src, int64_t first, int64_t count) {
    if (!sl || !src) return fail(CDS_ERR_INVALID_ARGUMENT, "NULL argument");
    if (first < 0 || count < 0 || first + count > src->capacity) return fail(CDS_ERR_INVALID_ARGUMENT, "slot range outside the source library");
    if (src->W != sl->W || src->H != sl->H) return fail(CDS_ERR_INVALID_ARGUMENT, "image size of the source library differs");
    if (src->ctx != sl->ctx) return fail(CDS_ERR_INVALID_ARGUMENT, "libraries belong to different contexts");
    if (count == 0) return CDS_OK;
    std::lock_guard<std::mutex> lk(sl->mu);
    CDS_HIP(hipSetDevice(sl->ctx->device));
    if (sl->n_images + count > sl->cap_images && sl->growable) {
        const int rcg = sparse_grow(sl, sl->n_images + count, 0);
        if (rcg) return rcg;
    }
    if (sl->n_images + count > sl->cap_images) return fail(CDS_ERR_OUT_OF_MEMORY, "sparse library is full (images)");
    AppendTally tally;
    append_begin(sl, count, tally);
    return append_end(sl, tally, sparse_append_locked(sl, src, first, count, tally));
}

int cds_sparse_library_append_host(cds_sparse_library *sl, int64_t count, const uint8_t *rgb) {
    if (!sl) return fail(CDS_ERR_INVALID_ARGUMENT, "sparse library is NULL");
    if (count < 0) return fail(CDS_ERR_INVALID_ARGUMENT, "count is negative");
    if (count == 0) return CDS_OK;
    if (!rgb) return fail(CDS_ERR_INVALID_ARGUMENT, "rgb is NULL");
    std::lock_guard<std::mutex> lk(sl->mu);
    CDS_HIP(hipSetDevice(sl->ctx->device));
    if (sl->n_images + count > sl->cap_images && sl->growable) {
        const int rcg = sparse_grow(sl, sl->n_images + count, 0);
        if (rcg) return rcg;
    }
    if (sl->n_images + count > sl->cap_images) return fail(CDS_ERR_OUT_OF_MEMORY, "sparse library is full (images)");
    const int64_t stage_n = 64;  // dense staging slots (132 MB at 1210 x 566)
    if (!sl->stage) {
        int rc = cds_library_create(sl->ctx, sl->W, sl->H, stage_n, &sl->stage);
        if (rc) return rc;
    }
    const size_t img_bytes = (size_t)3 * sl->W * sl->H;
    AppendTally tally;
    append_begin(sl, count, tally);
    int rc = CDS_OK;
    for (int64_t done = 0; done < count && rc == CDS_OK; done += stage_n) {
        const int64_t n = std::min(stage_n, count - done);
        rc = cds_library_upload(sl->stage, 0, n, rgb + (size_t)done * img_bytes);
        if (rc == CDS_OK) rc = sparse_append_locked(sl, sl->stage, 0, n, tally);
    }
    return append_end(sl, tally, rc);
}

// cds_sparse_library_append_files commits batch by batch (a file that cannot be read leaves the earlier batches in place, as
// documented); a density refusal, however, must leave nothing of the call behind: cds_imageio.hip rolls back with this.
}  // extern "C"
int cds::sparse_library_rollback(cds_sparse_library *sl, int64_t n_images, int64_t n_px) {
    if (!sl) return CDS_OK;
    std::lock_guard<std::mutex> lk(sl->mu);
    if (n_images <= sl->n_images && n_px <= sl->n_px) {
        sl->n_images = n_images;
        sl->n_px = n_px;
    }
    return CDS_OK;
}
extern "C" {

int cds_sparse_library_image_records(const cds_sparse_library *sl, int64_t index, uint32_t *records, int64_t capacity,
                                     int64_t *n_records, uint32_t *row_offsets) {
    if (!sl || !n_records) return fail(CDS_ERR_INVALID_ARGUMENT, "NULL argument");
    if (index < 0 || index >= sl->n_images) return fail(CDS_ERR_INVALID_ARGUMENT, "image index outside the sparse library");
    CDS_HIP(hipSetDevice(sl->ctx->device));
    uint64_t off[2];
    CDS_HIP(hipMemcpy(off, sl->d_img_off + index, sizeof off, hipMemcpyDeviceToHost));
    const int64_t n = (int64_t)(off[1] - off[0]);
    *n_records = n;
    if (records) {
        if (capacity < n) return fail(CDS_ERR_INVALID_ARGUMENT, "record buffer too small");
        CDS_HIP(hipMemcpy(records, sl->d_px + off[0], (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    }
    if (row_offsets)
        CDS_HIP(hipMemcpy(row_offsets, sl->d_rowptr + (size_t)index * sl->nstrips * (sl->H + 1), (size_t)sl->nstrips * (sl->H + 1) * sizeof(uint32_t),
                          hipMemcpyDeviceToHost));
    return CDS_OK;
}

int cds_sparse_library_image_offsets(const cds_sparse_library *sl, int64_t first, int64_t count, uint64_t *offsets) {
    int rc = sparse_check_range(sl, first, count);
    if (rc) return rc;
    if (!offsets) return fail(CDS_ERR_INVALID_ARGUMENT, "offsets is NULL");
    CDS_HIP(hipSetDevice(sl->ctx->device));
    CDS_HIP(hipStreamSynchronize(sl->ctx->stream));
    CDS_HIP(hipMemcpy(offsets, sl->d_img_off + first, (size_t)(count + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return CDS_OK;
}

int cds_maskset_sparse_groups(const cds_maskset *ms, int32_t *n_groups, int32_t *n_multi_groups, int32_t *n_multi_masks) {
    if (!ms) return fail(CDS_ERR_INVALID_ARGUMENT, "NULL argument");
    int g = 0, gm = 0, mm = 0;
    for (const SparseGroupPlan &gp : ms->sg_plans) {
        g++;
        if (gp.ok) gm++, mm += gp.n_masks;
    }
    if (n_groups) *n_groups = g;
    if (n_multi_groups) *n_multi_groups = gm;
    if (n_multi_masks) *n_multi_masks = mm;
    return CDS_OK;
}

int cds_maskset_sparse_layout(const cds_maskset *ms, int32_t mask_index, int32_t *out8) {
    if (!ms || !out8) return fail(CDS_ERR_INVALID_ARGUMENT, "NULL argument");
    if (mask_index < 0 || mask_index >= ms->n_masks || (size_t)mask_index >= ms->sp_plans.size()) return fail(CDS_ERR_INVALID_ARGUMENT, "mask_index out of range");
    const SparseMaskPlan &pl = ms->sp_plans[(size_t)mask_index];
    out8[0] = pl.n_slabs; out8[1] = pl.slab_rows; out8[2] = (int32_t)pl.lds_bytes; out8[3] = pl.cols;
    out8[4] = pl.nrows; out8[5] = pl.n_iv; out8[6] = pl.plane8 ? 1 : 2; out8[7] = pl.iv_in_lds ? 1 : 0;
    return CDS_OK;
}

// ---- export / import of the record arrays (a library converted once is reloaded at disk speed) -------------------------
int cds_sparse_library_export(const cds_sparse_library *sl, int64_t first, int64_t count, uint64_t *image_offsets, uint32_t *cell_offsets,
                              uint32_t *records, int64_t records_capacity) {
    int rc = sparse_check_range(sl, first, count);
    if (rc) return rc;
    if (!image_offsets) return fail(CDS_ERR_INVALID_ARGUMENT, "image_offsets is NULL");
    CDS_HIP(hipSetDevice(sl->ctx->device));
    CDS_HIP(hipStreamSynchronize(sl->ctx->stream));
    CDS_HIP(hipMemcpy(image_offsets, sl->d_img_off + first, (size_t)(count + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost));
    const uint64_t base = image_offsets[0], n_rec = image_offsets[count] - base;
    for (int64_t i = 0; i <= count; i++) image_offsets[i] -= base;   // relative to the first exported image
    if (cell_offsets && count > 0)
        CDS_HIP(hipMemcpy(cell_offsets, sl->d_rowptr + (size_t)first * sl->nstrips * (sl->H + 1), (size_t)count * sl->nstrips * (sl->H + 1) * sizeof(uint32_t),
                          hipMemcpyDeviceToHost));
    if (records) {
        if ((uint64_t)records_capacity < n_rec) return fail(CDS_ERR_INVALID_ARGUMENT, "record buffer too small");
        if (n_rec) CDS_HIP(hipMemcpy(records, sl->d_px + base, (size_t)n_rec * sizeof(uint32_t), hipMemcpyDeviceToHost));
    }
    return CDS_OK;
}

int cds_sparse_library_import(cds_sparse_library *sl, int64_t count, const uint64_t *image_offsets, const uint32_t *cell_offsets,
                              const uint32_t *records, int32_t n_strips, int32_t strip_width, int32_t min_data_threshold) {
    if (!sl) return fail(CDS_ERR_INVALID_ARGUMENT, "sparse library is NULL");
    if (count < 0) return fail(CDS_ERR_INVALID_ARGUMENT, "count is negative");
    if (min_data_threshold < 0 || min_data_threshold > 255) return fail(CDS_ERR_INVALID_ARGUMENT, "minimum data threshold must be 0..255");
    if (count > 0 && min_data_threshold > sl->min_threshold) {
        char buf[260];
        snprintf(buf, sizeof buf, "the records were made with minimum data threshold %d (pixels at or below it are missing), this library promises searches "
                 "down to data threshold %d: raise it first (cds_sparse_library_set_min_threshold) or convert the images again", min_data_threshold, sl->min_threshold);
        return fail(CDS_ERR_INVALID_ARGUMENT, buf);
    }
    if (count == 0) return CDS_OK;
    if (!image_offsets || !cell_offsets || !records) return fail(CDS_ERR_INVALID_ARGUMENT, "NULL argument");
    if (n_strips != sl->nstrips || strip_width != sl->strip_w) return fail(CDS_ERR_INVALID_ARGUMENT, "record layout (strips) differs from this library's");
    std::lock_guard<std::mutex> lk(sl->mu);
    const uint64_t n_rec = image_offsets[count] - image_offsets[0];
    if (sl->growable && (sl->n_images + count > sl->cap_images || sl->n_px + (int64_t)n_rec > sl->cap_px)) {
        CDS_HIP(hipSetDevice(sl->ctx->device));
        const int rcg = sparse_grow(sl, sl->n_images + count, sl->n_px + (int64_t)n_rec);
        if (rcg) return rcg;
    }
    if (sl->n_images + count > sl->cap_images) return fail(CDS_ERR_OUT_OF_MEMORY, "sparse library is full (images)");
    if (sl->n_px + (int64_t)n_rec > sl->cap_px) return fail(CDS_ERR_OUT_OF_MEMORY, "sparse library is full (records)");
    // consistency of the untrusted tables: monotone image offsets, every image's last cell offset == its record count
    const size_t cells = (size_t)sl->nstrips * (sl->H + 1);
    for (int64_t i = 0; i < count; i++) {
        if (image_offsets[i + 1] < image_offsets[i]) return fail(CDS_ERR_INVALID_ARGUMENT, "image offsets are not monotone");
        const uint32_t *co = cell_offsets + (size_t)i * cells;
        if (co[0] != 0 || co[cells - 1] != image_offsets[i + 1] - image_offsets[i]) return fail(CDS_ERR_INVALID_ARGUMENT, "cell offsets do not match the image's record count");
        for (int k = 0; k < sl->nstrips; k++) {   // [strip][H + 1]: strictly increasing inside a strip (no empty cell), strips back to back
            const uint32_t *ck = co + (size_t)k * (sl->H + 1);
            for (int r = 0; r < sl->H; r++)
                if (ck[r + 1] <= ck[r]) return fail(CDS_ERR_INVALID_ARGUMENT, "cell offsets are not strictly increasing (every cell holds at least one record)");
            if (k + 1 < sl->nstrips && ck[sl->H] != ck[sl->H + 1]) return fail(CDS_ERR_INVALID_ARGUMENT, "cell offsets of consecutive strips do not meet");
        }
    }
    {
        const double density = (double)n_rec / ((double)count * sl->W * sl->H);
        if (density > sl->density_limit) return density_refused(density, sl->density_limit);
    }
    CDS_HIP(hipSetDevice(sl->ctx->device));
    hipStream_t st = sl->ctx->stream;
    std::vector<uint64_t> off((size_t)count + 1);
    for (int64_t i = 0; i <= count; i++) off[(size_t)i] = (uint64_t)sl->n_px + (image_offsets[i] - image_offsets[0]);
    CDS_HIP(hipMemcpyAsync(sl->d_img_off + sl->n_images, off.data(), off.size() * sizeof(uint64_t), hipMemcpyHostToDevice, st));
    CDS_HIP(hipMemcpyAsync(sl->d_rowptr + (size_t)sl->n_images * cells, cell_offsets, (size_t)count * cells * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    if (n_rec) CDS_HIP(hipMemcpyAsync(sl->d_px + sl->n_px, records + image_offsets[0], (size_t)n_rec * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    {   // the records themselves, on the device (nothing is committed before they pass)
        unsigned long long *d_err = nullptr, err[2] = {0, 0};
        CDS_HIP(dev_malloc(&d_err, sizeof err));
        hipError_t e = hipMemsetAsync(d_err, 0, sizeof err, st);
        ValidateParams vp{sl->d_px, sl->d_img_off, sl->d_rowptr, sl->n_images, count, sl->W, sl->H, sl->nstrips, sl->strip_w, d_err};
        const int64_t n_cells = count * sl->nstrips * sl->H;
        if (e == hipSuccess) {
            hipLaunchKernelGGL(sparse_validate_kernel, dim3((unsigned)((n_cells + 255) / 256)), dim3(256), 0, st, vp);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(err, d_err, sizeof err, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        hipFree(d_err);
        if (e != hipSuccess) return hip_fail(e, "validate imported records");
        if (err[0]) {
            char buf[200];
            snprintf(buf, sizeof buf, "imported records are inconsistent: %llu bad (strip, row) cells, e.g. in image %llu of the batch (row flags / x order / class / filler rules)",
                     err[0], err[1] - 1);
            return fail(CDS_ERR_INVALID_ARGUMENT, buf);
        }
    }
    {
        const int rc16 = refresh_px16(sl, sl->n_px, (int64_t)n_rec);
        if (rc16) return rc16;
    }
    CDS_HIP(hipStreamSynchronize(st));
    sl->n_images += count;
    sl->n_px += (int64_t)n_rec;
    return CDS_OK;
}

int cds_search_sparse_device(const cds_maskset *ms, const cds_sparse_library *sl, int64_t first, int64_t count, int32_t *d_scores) {
    if (!d_scores && count > 0) return fail(CDS_ERR_INVALID_ARGUMENT, "d_scores is NULL");
    return sparse_search_impl(ms, sl, first, count, d_scores, nullptr, getenv("CDS_TIME_KERNELS") != nullptr);
}

static int sparse_search_host(const cds_maskset *ms, const cds_sparse_library *sl, int64_t first, int64_t count, int32_t *scores,
                              uint8_t *mirrored, int32_t *counts) {
    if (!ms || !sl) return fail(CDS_ERR_INVALID_ARGUMENT, "NULL argument");
    int rc = sparse_check_range(sl, first, count);
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
    rc = sparse_search_impl(ms, sl, first, count, d_scores, d_counts, true);
    if (rc == CDS_OK) {
        std::vector<int32_t> packed(n);
        hipError_t e = hipMemcpyAsync(packed.data(), d_scores, n * sizeof(int32_t), hipMemcpyDeviceToHost, ms->ctx->stream);
        if (e == hipSuccess && counts)
            e = hipMemcpyAsync(counts, d_counts, n * CDS_N_TRANSFORMS * sizeof(int32_t), hipMemcpyDeviceToHost, ms->ctx->stream);
        if (e == hipSuccess) e = sync_here(ms->ctx->stream);
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

// The match list stays on the device until it is fetched: its length is not known before the scan, and a caller that guessed
// too small a host buffer must not pay for a second scan.
struct cds_match_list {
    cds_context *ctx = nullptr;
    cds_match *d = nullptr;
    int64_t n = 0, cap = 0;
};

int cds_search_sparse_match_list(const cds_maskset *ms, const cds_sparse_library *sl, int64_t first, int64_t count, double pct_positive_pixels,
                                 cds_match_list **out, int64_t *n_matches) {
    if (!ms || !sl || !out) return fail(CDS_ERR_INVALID_ARGUMENT, "NULL argument");
    *out = nullptr;
    if (n_matches) *n_matches = 0;
    int rc = sparse_check_range(sl, first, count);
    if (rc) return rc;
    if (count >= ((int64_t)1 << 31)) return fail(CDS_ERR_UNSUPPORTED, "more than 2^31 targets in one call");
    cds_context *ctx = ms->ctx;
    CDS_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    cds_match_list *ml = new cds_match_list();
    ml->ctx = ctx;
    if (count == 0) {
        *out = ml;
        return CDS_OK;
    }
    // smallest passing score per mask (isMatch is monotone in the score); an empty mask is skipped as the reference does
    // ("No computation created ... because it is empty", LocalColorMIPSearchProcessor.java:88-92)
    std::vector<int32_t> min_score((size_t)ms->n_masks, INT32_MAX);
    for (int m = 0; m < ms->n_masks; m++) {
        const int32_t P = ms->sizes[(size_t)m];
        if (P <= 0 || !cds_is_match(P, P, pct_positive_pixels)) continue;
        int32_t lo = 1, hi = P;
        while (lo < hi) {
            const int32_t mid = lo + (hi - lo) / 2;
            if (cds_is_match(mid, P, pct_positive_pixels)) hi = mid; else lo = mid + 1;
        }
        min_score[(size_t)m] = lo;
    }
    // chunks of masks: at most 128 masks and fewer than 2^32 pairs each, never cutting a multi-mask group
    int max_chunk = (int)std::max<int64_t>(1, std::min<int64_t>(128, (((int64_t)1 << 32) - 1) / count));
    if (const char *e = getenv("CDS_MATCH_CHUNK_MASKS")) max_chunk = std::max(1, atoi(e));
    std::vector<std::pair<int, int>> chunks;
    if (ms->sg_plans.empty()) {
        for (int m = 0; m < ms->n_masks; m += max_chunk) chunks.push_back({m, std::min(ms->n_masks, m + max_chunk)});
    } else {
        int b = 0, e = 0;
        for (const SparseGroupPlan &gp : ms->sg_plans) {
            if (e > b && gp.first_mask + gp.n_masks - b > max_chunk) {
                chunks.push_back({b, e});
                b = e;
            }
            e = gp.first_mask + gp.n_masks;
        }
        if (e > b) chunks.push_back({b, e});
    }
    int widest = 0;
    for (auto &c : chunks) widest = std::max(widest, c.second - c.first);
    const int nblk = (int)((count + 1023) / 1024);
    int32_t *d_scores = nullptr, *d_min = nullptr;
    uint32_t *d_blockcnt = nullptr;
    unsigned long long *d_total = nullptr;
    auto cleanup = [&](bool in_flight = true) {   // (error paths: the launches that use these blocks may still be running)
        if (in_flight) hipStreamSynchronize(st);
        cache_free(ctx, d_scores);
        cache_free(ctx, d_min);
        cache_free(ctx, d_blockcnt);
        cache_free(ctx, d_total);
    };
#define MT_HIP(call)                         \
    do {                                     \
        hipError_t _e = (call);              \
        if (_e != hipSuccess) {              \
            cleanup();                       \
            cds_match_list_destroy(ml);      \
            return cds::hip_fail(_e, #call); \
        }                                    \
    } while (0)
    MT_HIP(cache_alloc(ctx, (void **)&d_scores, (size_t)widest * count * sizeof(int32_t)));
    MT_HIP(cache_alloc(ctx, (void **)&d_min, (size_t)ms->n_masks * sizeof(int32_t)));
    MT_HIP(cache_alloc(ctx, (void **)&d_blockcnt, (size_t)widest * nblk * sizeof(uint32_t)));
    MT_HIP(cache_alloc(ctx, (void **)&d_total, 2 * sizeof(unsigned long long)));
    MT_HIP(hipMemcpyAsync(d_min, min_score.data(), min_score.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
    MT_HIP(hipMemsetAsync(d_total, 0, 2 * sizeof(unsigned long long), st));
    // ONE chunk (up to 128 masks: what a pipelining caller sends): scan, count, offsets and fill are enqueued back to back into a list
    // sized by the previous call's length (or an eighth of the pairs) and the host waits once.  With the length read back BETWEEN count
    // and fill, the fill of this set queued behind the scan that another host thread had enqueued meanwhile for the NEXT set -- 45 ms of
    // idle device after every second set of the processor's pipeline.  A list that turns out too short is filled again at its exact size
    // (the scores are still there); CDS_MATCH_GUESS_ROWS sets the first guess (tests).
    if (chunks.size() == 1 && !getenv("CDS_MATCH_TWO_PHASE")) {
        const auto &c = chunks[0];
        const int64_t pairs = (int64_t)(c.second - c.first) * count;
        const int64_t hint = (int64_t)ctx->match_rows_hint.load();
        int64_t guess = hint > 0 ? hint : pairs / 8;
        if (const char *e = getenv("CDS_MATCH_GUESS_ROWS")) guess = atoll(e);
        guess = std::max<int64_t>(1, std::min<int64_t>(pairs, std::max<int64_t>(guess, 1 << 16)));
        if (const char *e = getenv("CDS_MATCH_GUESS_ROWS")) guess = std::max<int64_t>(1, atoll(e));
        MT_HIP(cache_alloc(ctx, (void **)&ml->d, (size_t)guess * sizeof(cds_match)));
        ml->cap = guess;
        rc = sparse_search_impl(ms, sl, first, count, d_scores, nullptr, false, c.first, c.second);
        if (rc) {
            cleanup();
            cds_match_list_destroy(ml);
            return rc;
        }
        MatchParams p{};
        p.scores = d_scores; p.min_score = d_min + c.first; p.n_masks = c.second - c.first; p.mask0 = c.first;
        p.count = count; p.nblk = nblk; p.blockcnt = d_blockcnt; p.total = d_total;
        p.out = ml->d; p.capacity = ml->cap;
        const unsigned grid = (unsigned)((int64_t)p.n_masks * nblk);
        hipLaunchKernelGGL(match_count_kernel, dim3(grid), dim3(256), 0, st, p);
        hipLaunchKernelGGL(match_scan_kernel, dim3(1), dim3(1024), 0, st, p);
        hipLaunchKernelGGL(match_fill_kernel, dim3(grid), dim3(256), 0, st, p);
        MT_HIP(hipGetLastError());
        unsigned long long total[2] = {0, 0};
        MT_HIP(hipMemcpyAsync(total, d_total, sizeof total, hipMemcpyDeviceToHost, st));
        MT_HIP(sync_here(st));
        const int64_t need = (int64_t)total[1];
        if (need > ml->cap) {   // (the guess was too small: the rows beyond it were not written -- once more, at the exact size)
            cds_match *nd = nullptr;
            MT_HIP(cache_alloc(ctx, (void **)&nd, (size_t)need * sizeof(cds_match)));
            cache_free(ctx, ml->d);
            ml->d = nd;
            ml->cap = need;
            p.out = ml->d; p.capacity = ml->cap;
            hipLaunchKernelGGL(match_fill_kernel, dim3(grid), dim3(256), 0, st, p);
            MT_HIP(hipGetLastError());
            MT_HIP(sync_here(st));
        }
        ml->n = need;
        ctx->match_rows_hint.store((long long)(need + need / 4 + 65536));
        cleanup(false);
        if (n_matches) *n_matches = ml->n;
        *out = ml;
        return CDS_OK;
    }
    for (auto &c : chunks) {
        rc = sparse_search_impl(ms, sl, first, count, d_scores, nullptr, false, c.first, c.second);
        if (rc) {
            cleanup();
            cds_match_list_destroy(ml);
            return rc;
        }
        MatchParams p{};
        p.scores = d_scores; p.min_score = d_min + c.first; p.n_masks = c.second - c.first; p.mask0 = c.first;
        p.count = count; p.nblk = nblk; p.blockcnt = d_blockcnt; p.total = d_total;
        const unsigned grid = (unsigned)((int64_t)p.n_masks * nblk);
        hipLaunchKernelGGL(match_count_kernel, dim3(grid), dim3(256), 0, st, p);
        hipLaunchKernelGGL(match_scan_kernel, dim3(1), dim3(1024), 0, st, p);
        MT_HIP(hipGetLastError());
        unsigned long long total[2] = {0, 0};   // the chunk's length: the list grows to hold it before it is written
        MT_HIP(hipMemcpyAsync(total, d_total, sizeof total, hipMemcpyDeviceToHost, st));
        MT_HIP(sync_here(st));
        const int64_t need = (int64_t)(total[0] + total[1]);
        if (need > ml->cap) {
            const int64_t cap = std::max<int64_t>(need, std::max<int64_t>(1 << 16, ml->cap * 2));
            cds_match *nd = nullptr;
            MT_HIP(cache_alloc(ctx, (void **)&nd, (size_t)cap * sizeof(cds_match)));
            if (ml->n) MT_HIP(hipMemcpyAsync(nd, ml->d, (size_t)ml->n * sizeof(cds_match), hipMemcpyDeviceToDevice, st));
            MT_HIP(sync_here(st));
            cache_free(ctx, ml->d);
            ml->d = nd;
            ml->cap = cap;
        }
        p.out = ml->d; p.capacity = ml->cap;
        hipLaunchKernelGGL(match_fill_kernel, dim3(grid), dim3(256), 0, st, p);
        hipLaunchKernelGGL(match_advance_kernel, dim3(1), dim3(1), 0, st, d_total);
        MT_HIP(hipGetLastError());
        ml->n = need;
    }
    MT_HIP(sync_here(st));
    cleanup(false);
#undef MT_HIP
    if (n_matches) *n_matches = ml->n;
    *out = ml;
    return CDS_OK;
}

int64_t cds_match_list_size(const cds_match_list *ml) { return ml ? ml->n : 0; }

int cds_match_list_fetch(const cds_match_list *ml, int64_t first, int64_t count, cds_match *matches) {
    if (!ml) return fail(CDS_ERR_INVALID_ARGUMENT, "match list is NULL");
    if (first < 0 || count < 0 || first + count > ml->n) return fail(CDS_ERR_INVALID_ARGUMENT, "range outside the match list");
    if (count == 0) return CDS_OK;
    if (!matches) return fail(CDS_ERR_INVALID_ARGUMENT, "matches is NULL");
    CDS_HIP(hipSetDevice(ml->ctx->device));
    CDS_HIP(hipMemcpy(matches, ml->d + first, (size_t)count * sizeof(cds_match), hipMemcpyDeviceToHost));
    return CDS_OK;
}

int cds_match_list_fetch_columns(const cds_match_list *ml, int64_t first, int64_t count, const int32_t *query_sizes, int32_t n_masks,
                                 int32_t mask_offset, int32_t *mask, int64_t *target, int32_t *score, float *ratio, uint8_t *mirrored) {
    if (!ml) return fail(CDS_ERR_INVALID_ARGUMENT, "match list is NULL");
    if (first < 0 || count < 0 || first + count > ml->n) return fail(CDS_ERR_INVALID_ARGUMENT, "range outside the match list");
    if (ratio && (!query_sizes || n_masks <= 0)) return fail(CDS_ERR_INVALID_ARGUMENT, "ratio wanted without the query sizes");
    if (count == 0) return CDS_OK;
    CDS_HIP(hipSetDevice(ml->ctx->device));
    // the rows come down in pieces of 4 M (48 MB) and are split into the columns by a few host threads while the next piece is copied.
    // No C++ exception may cross the C ABI (ctypes, JNI), and none may escape a std::thread (std::terminate): the buffers' bad_alloc is
    // caught here, a refused thread (std::system_error under a container's thread limit) falls back to the calling thread, and the
    // splitter is joined by a scope guard on every way out.
    const int64_t piece = (int64_t)4 << 20;
    std::atomic<int> bad{0};
    auto work = [&](const cds_match *src, int64_t at, int64_t n, int k, int nt) {
        const int64_t a = n * k / nt, b = n * (k + 1) / nt;
        for (int64_t i = a; i < b; i++) {
            const cds_match r = src[i];
            const int32_t sc = r.score & 0x7fffffff;
            if (mask) mask[at + i] = r.mask + mask_offset;
            if (target) target[at + i] = r.target;
            if (score) score[at + i] = sc;
            if (mirrored) mirrored[at + i] = (uint8_t)((uint32_t)r.score >> 31);
            if (ratio) {
                if (r.mask < 0 || r.mask >= n_masks) { bad.store(1); continue; }
                ratio[at + i] = (float)((double)sc / std::max((double)query_sizes[r.mask], 1.0));
            }
        }
    };
    auto split = [&](const cds_match *src, int64_t at, int64_t n) noexcept {   // (nothing in here throws past its own catch)
        const unsigned hc = std::thread::hardware_concurrency();
        const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<unsigned>(hc ? hc : 1u, 8u), n / 65536 + 1));
        std::vector<std::thread> pool;
        int started = 1;   // share 0 is the calling thread's
        try {
            pool.reserve((size_t)nt);
            for (; started < nt; started++) pool.emplace_back(work, src, at, n, started, nt);
        } catch (...) {   // no more threads (or no memory for the handle): the shares not handed out are done here
        }
        work(src, at, n, 0, nt);
        for (int k = started; k < nt; k++) work(src, at, n, k, nt);
        for (auto &t : pool) t.join();
    };
    std::vector<cds_match> buf[2];   // (declared before the splitter: destroyed after it has been joined, on every way out)
    struct Joiner {
        std::thread t;
        ~Joiner() { if (t.joinable()) t.join(); }
    } splitter;
    try {
        int which = 0;
        hipError_t e = hipSuccess;
        for (int64_t at = 0; at < count && e == hipSuccess; at += piece, which ^= 1) {
            const int64_t n = std::min(piece, count - at);
            buf[which].resize((size_t)n);
            e = hipMemcpy(buf[which].data(), ml->d + first + at, (size_t)n * sizeof(cds_match), hipMemcpyDeviceToHost);
            if (splitter.t.joinable()) splitter.t.join();   // (the other buffer is free again before the next copy goes into it)
            if (e != hipSuccess) break;
            const cds_match *src = buf[which].data();
            try {
                splitter.t = std::thread(split, src, at, n);
            } catch (const std::system_error &) {
                split(src, at, n);   // no thread to be had: split this piece before the next copy
            }
        }
        if (splitter.t.joinable()) splitter.t.join();
        if (e != hipSuccess) return hip_fail(e, "hipMemcpy(match list)");
    } catch (const std::bad_alloc &) {
        if (splitter.t.joinable()) splitter.t.join();
        return fail(CDS_ERR_OUT_OF_MEMORY, "out of host memory for the match list's staging buffers");
    } catch (const std::exception &ex) {
        if (splitter.t.joinable()) splitter.t.join();
        return fail(CDS_ERR_HIP, std::string("cds_match_list_fetch_columns: ") + ex.what());
    }
    if (bad.load()) return fail(CDS_ERR_INVALID_ARGUMENT, "the match list names a mask beyond n_masks: query_sizes of another mask set?");
    return CDS_OK;
}

void cds_match_list_destroy(cds_match_list *ml) {
    if (!ml) return;
    if (ml->d) {
        hipSetDevice(ml->ctx->device);
        cache_free(ml->ctx, ml->d);
    }
    delete ml;
}

int cds_search_sparse_matches(const cds_maskset *ms, const cds_sparse_library *sl, int64_t first, int64_t count, double pct_positive_pixels,
                              cds_match *matches, int64_t capacity, int64_t *n_matches) {
    if (!n_matches) return fail(CDS_ERR_INVALID_ARGUMENT, "NULL argument");
    *n_matches = 0;
    if (capacity < 0 || (capacity > 0 && !matches)) return fail(CDS_ERR_INVALID_ARGUMENT, "bad match buffer");
    cds_match_list *ml = nullptr;
    int rc = cds_search_sparse_match_list(ms, sl, first, count, pct_positive_pixels, &ml, n_matches);
    if (rc) return rc;
    if (*n_matches > capacity) {
        cds_match_list_destroy(ml);
        char buf[200];
        snprintf(buf, sizeof buf, "match buffer too small: %lld matches, capacity %lld (n_matches holds the count; cds_search_sparse_match_list avoids the second scan)",
                 (long long)*n_matches, (long long)capacity);
        return fail(CDS_ERR_OUT_OF_MEMORY, buf);
    }
    rc = cds_match_list_fetch(ml, 0, *n_matches, matches);
    cds_match_list_destroy(ml);
    return rc;
}

int cds_scores_topk(const cds_maskset *ms, const int32_t *d_scores, int64_t count, int32_t k, int32_t *targets, int32_t *scores) {
    if (!ms || !d_scores || !targets || !scores) return fail(CDS_ERR_INVALID_ARGUMENT, "NULL argument");
    if (k < 1 || k > 1024) return fail(CDS_ERR_INVALID_ARGUMENT, "k must be 1..1024");
    if (count <= 0 || count >= ((int64_t)1 << 31)) return fail(CDS_ERR_INVALID_ARGUMENT, "count must be 1 .. 2^31 - 1");
    cds_context *ctx = ms->ctx;
    CDS_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const size_t n_out = (size_t)ms->n_masks * (size_t)k;
    int32_t *d_q = nullptr, *d_t = nullptr, *d_s = nullptr;
    auto cleanup = [&]() {
        for (void *q : {(void *)d_q, (void *)d_t, (void *)d_s})
            if (q) hipFree(q);
    };
#define TK_HIP(call)                         \
    do {                                     \
        hipError_t _e = (call);              \
        if (_e != hipSuccess) {              \
            cleanup();                       \
            return cds::hip_fail(_e, #call); \
        }                                    \
    } while (0)
    TK_HIP(dev_malloc(&d_q, (size_t)ms->n_masks * sizeof(int32_t)));
    TK_HIP(dev_malloc(&d_t, n_out * sizeof(int32_t)));
    TK_HIP(dev_malloc(&d_s, n_out * sizeof(int32_t)));
    TK_HIP(hipMemcpyAsync(d_q, ms->sizes.data(), (size_t)ms->n_masks * sizeof(int32_t), hipMemcpyHostToDevice, st));
    TopkParams p{};
    p.scores = d_scores; p.qsize = d_q; p.count = count; p.k = k; p.kp = 1;
    while (p.kp < k) p.kp <<= 1;
    p.out_target = d_t; p.out_score = d_s;
    hipLaunchKernelGGL(topk_kernel, dim3((unsigned)ms->n_masks), dim3(1024), 0, st, p);
    TK_HIP(hipGetLastError());
    TK_HIP(hipMemcpyAsync(targets, d_t, n_out * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    TK_HIP(hipMemcpyAsync(scores, d_s, n_out * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    TK_HIP(hipStreamSynchronize(st));
    cleanup();
#undef TK_HIP
    return CDS_OK;
}

int cds_search_sparse_topk(const cds_maskset *ms, const cds_sparse_library *sl, int64_t first, int64_t count, int32_t k, int32_t *targets,
                           int32_t *scores) {
    if (!ms || !sl || !targets || !scores) return fail(CDS_ERR_INVALID_ARGUMENT, "NULL argument");
    if (k < 1 || k > 1024) return fail(CDS_ERR_INVALID_ARGUMENT, "k must be 1..1024");
    int rc = sparse_check_range(sl, first, count);
    if (rc) return rc;
    if (count >= ((int64_t)1 << 31)) return fail(CDS_ERR_UNSUPPORTED, "more than 2^31 targets in one call");
    const size_t n_out = (size_t)ms->n_masks * (size_t)k;
    if (count == 0) {
        for (size_t i = 0; i < n_out; i++) targets[i] = -1, scores[i] = 0;
        return CDS_OK;
    }
    cds_context *ctx = ms->ctx;
    CDS_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    int max_chunk = (int)std::max<int64_t>(1, std::min<int64_t>(128, (((int64_t)1 << 32) - 1) / count));
    if (const char *e = getenv("CDS_MATCH_CHUNK_MASKS")) max_chunk = std::max(1, atoi(e));
    std::vector<std::pair<int, int>> chunks;
    if (ms->sg_plans.empty()) {
        for (int m = 0; m < ms->n_masks; m += max_chunk) chunks.push_back({m, std::min(ms->n_masks, m + max_chunk)});
    } else {
        int b = 0, e = 0;
        for (const SparseGroupPlan &gp : ms->sg_plans) {
            if (e > b && gp.first_mask + gp.n_masks - b > max_chunk) {
                chunks.push_back({b, e});
                b = e;
            }
            e = gp.first_mask + gp.n_masks;
        }
        if (e > b) chunks.push_back({b, e});
    }
    int widest = 0;
    for (auto &c : chunks) widest = std::max(widest, c.second - c.first);
    int32_t *d_scores = nullptr, *d_q = nullptr, *d_t = nullptr, *d_s = nullptr;
    auto cleanup = [&]() {
        for (void *q : {(void *)d_scores, (void *)d_q, (void *)d_t, (void *)d_s})
            if (q) hipFree(q);
    };
#define TK_HIP(call)                         \
    do {                                     \
        hipError_t _e = (call);              \
        if (_e != hipSuccess) {              \
            cleanup();                       \
            return cds::hip_fail(_e, #call); \
        }                                    \
    } while (0)
    TK_HIP(dev_malloc(&d_scores, (size_t)widest * count * sizeof(int32_t)));
    TK_HIP(dev_malloc(&d_q, (size_t)ms->n_masks * sizeof(int32_t)));
    TK_HIP(dev_malloc(&d_t, n_out * sizeof(int32_t)));
    TK_HIP(dev_malloc(&d_s, n_out * sizeof(int32_t)));
    TK_HIP(hipMemcpyAsync(d_q, ms->sizes.data(), (size_t)ms->n_masks * sizeof(int32_t), hipMemcpyHostToDevice, st));
    int kp = 1;
    while (kp < k) kp <<= 1;
    for (auto &c : chunks) {
        rc = sparse_search_impl(ms, sl, first, count, d_scores, nullptr, false, c.first, c.second);
        if (rc) {
            cleanup();
            return rc;
        }
        TopkParams p{};
        p.scores = d_scores; p.qsize = d_q + c.first; p.count = count; p.k = k; p.kp = kp;
        p.out_target = d_t + (size_t)c.first * k; p.out_score = d_s + (size_t)c.first * k;
        hipLaunchKernelGGL(topk_kernel, dim3((unsigned)(c.second - c.first)), dim3(1024), 0, st, p);
        TK_HIP(hipGetLastError());
    }
    TK_HIP(hipMemcpyAsync(targets, d_t, n_out * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    TK_HIP(hipMemcpyAsync(scores, d_s, n_out * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    TK_HIP(hipStreamSynchronize(st));
    cleanup();
#undef TK_HIP
    return CDS_OK;
}

int cds_search_sparse(const cds_maskset *ms, const cds_sparse_library *sl, int64_t first, int64_t count, int32_t *scores,
                      uint8_t *mirrored) {
    return sparse_search_host(ms, sl, first, count, scores, mirrored, nullptr);
}

int cds_search_sparse_counts(const cds_maskset *ms, const cds_sparse_library *sl, int64_t first, int64_t count, int32_t *counts) {
    if (!counts && count > 0) return fail(CDS_ERR_INVALID_ARGUMENT, "counts is NULL");
    return sparse_search_host(ms, sl, first, count, nullptr, nullptr, counts);
}


// which single-mask kernel the last sparse search of this mask set launched (0: 4-byte records, 1: the 16-bit stage-1 stream, -1: none yet)
// and the share of visited records that reached an evaluation round in its last pilot (-1: no pilot yet)
int cds_debug_maskset_scan_kernel(const cds_maskset *ms, int32_t *kernel, double *pilot_ratio) {
    if (!ms) return fail(CDS_ERR_INVALID_ARGUMENT, "NULL argument");
    std::lock_guard<std::mutex> lk(ms->sp_mu);
    if (kernel) *kernel = ms->last_scan_kernel;
    if (pilot_ratio) *pilot_ratio = ms->last_pilot_ratio;
    return CDS_OK;
}

// per CONTEXT: another context's searches (another thread's, another GPU's) are never affected
int cds_debug_sparse_load_only(cds_context *ctx, int32_t on) {
    if (!ctx) return fail(CDS_ERR_INVALID_ARGUMENT, "NULL context");
    ctx->sparse_load_only.store(on ? 1 : 0);
    return CDS_OK;
}
int32_t cds_debug_sparse_load_only_state(const cds_context *ctx) { return ctx ? ctx->sparse_load_only.load() : 0; }

#if SP_BAND
// diagnostics of the stage-1 hue-band map (include/cds_hip_debug.h; host arithmetic identical to the kernel's)
int cds_debug_band_bits(uint32_t lo, uint32_t len, uint32_t *bits) {
    if (!bits) return fail(CDS_ERR_INVALID_ARGUMENT, "NULL argument");
    *bits = band_bits_of(lo, len);
    return CDS_OK;
}
int32_t cds_debug_band_of(const float *rcp_table, int32_t cls, int32_t second, int32_t max) {
    if (!rcp_table || max < 0 || max > 255 || second < 0 || second > 255) return -1;
    const float hf = (float)second * rcp_table[max];
    return (int32_t)(((uint32_t)cls * 5u + (uint32_t)(hf * 5.0f)) & 31u);
}
#endif

}  // extern "C"
