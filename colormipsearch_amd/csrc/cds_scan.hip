// cds_scan.hip -- kernel 1: the mask-vs-library pixel-match scan for gfx950 (MI355X).
//
// What it computes (reference: API/cds/PixelMatchColorDepthSearchAlgorithm.java:166-263 with
// calculatePixelGap API/cds/AbstractColorDepthSearchAlgorithm.java:157-390): for every (mask, target)
// pair and every transform (9 xy-shifts x {plain, mirrored}) the number of mask pixels whose
// transformed target pixel passes the data threshold and lies within zTolerance in hue-ratio space;
// then max over shifts, mirrored only if strictly greater.
//
// How (MI355X-first, not the reference's gather loops) -- see DESIGN.md section 4 for the measurements behind it:
//   * one workgroup streams ONE target at a time: only the rows the mask can touch, and of those only the
//     column window the mask (and its mirror image) can touch, as 12-byte (4 pixel) non-temporal loads, 768
//     contiguous bytes per wave-instruction;
//   * the mask's rows are cut into bands of 12; per band, two barriers:
//     stage: raw 12-byte groups into a 16-row LDS ring as (a,b,c,0) (an all-black group is already its all-zero
//        code), the segment ballots become occupancy words, and the staging wave converts the non-black groups it
//        just wrote (its own small queue, no atomics) to 32-bit "hue codes" = class<<29 | bits(med * v_rcp_f32(max))>>1
//        (0 = below threshold / no hue class).  The host builds its acceptance intervals from this device's own
//        v_rcp_f32 table, so the device predicate is an integer range test that is bit-exact with the reference's
//        double arithmetic by construction (cds_host.hip: build_ratio_table / intervals_for).  The loads of band b+1
//        are issued at the top of band b by asm with manual s_waitcnt, so HBM requests stay in flight during the probes;
//     probe: mask pixels are pre-sorted into spatially compact chunks of 64 per band; a wave tests the neighbourhoods of
//        its chunks against the band's occupancy with one ballot and skips an orientation's 9 probes where all is
//        black; otherwise every lane probes its 9 (18) transformed positions in LDS against its two code intervals
//        (own hue class + at most one adjacent class);
//   * mask batches (n_masks > 1) share the staging: chunk records band-major, blocks of chunks drawn from an LDS counter
//     by 16 waves, records loaded only for occupied chunks, matches counted by s_bcnt1 into SGPR counters, a ring that
//     never wraps (4 mirrored rows) with host-precomputed LDS offsets -- see the comment at the batch branch below;
//   * no MFMA: byte compare / integer count; the kernel is instruction-issue-, synchronisation- and HBM-bound.
//   * CDS_SCAN_DEBUG bits (timing-only ablations, results are wrong with them): 1 skip probes, 2 skip conversion,
//     4 never skip by occupancy, 8 no queue, 16 skip the commit of prefetched rows, 64 every workgroup re-reads the
//     first 8 images (rows come from L2 instead of HBM).
#include "cds_internal.hpp"

namespace cds {

typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
typedef u32x3 u32x3_a4 __attribute__((aligned(4)));  // 12-byte groups are only 4-byte aligned
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// Asynchronous loads with MANUAL waits.  hipcc's own s_waitcnt placement treats vmcnt conservatively (vector-memory
// operations retire in order, and it waits with vmcnt(0) as soon as any register with a pending load is touched),
// which would make the probes wait for the HBM rows of the NEXT band.  The band loop therefore issues its loads --
// first the (L2-resident) mask data of this band, then the rows of the next band -- through these asm statements
// and waits explicitly: vmcnt(KMAX) before the probes (mask data arrived, rows still in flight), vmcnt(0) before
// the rows are committed.  No asm-loaded value is loop-carried or read before its wait statement, which names it
// as an in/out operand (cdna_hip_programming.md section 5.7).
__device__ __forceinline__ u32x4 async_ld16(const void *ptr) {
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(ptr) : "memory");
    return v;
}
template <int KMAX>
__device__ __forceinline__ void wait_rows(u32x3 (&w)[KMAX]) {  // all outstanding loads have landed; names w as in/out
    if (KMAX == 2) asm volatile("s_waitcnt vmcnt(0)" : "+v"(w[0]), "+v"(w[KMAX - 1]) :: "memory");
    else if (KMAX == 3) asm volatile("s_waitcnt vmcnt(0)" : "+v"(w[0]), "+v"(w[1]), "+v"(w[KMAX - 1]) :: "memory");
    else if (KMAX == 5) asm volatile("s_waitcnt vmcnt(0)" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[KMAX - 1]) :: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[KMAX - 1]) :: "memory");
}
template <int N>
__device__ __forceinline__ void wait_mask(u32x4 &ch, uint32_t &pos, u32x4 &iv) {  // at most N loads still outstanding
    asm volatile("s_waitcnt vmcnt(%3)" : "+v"(ch), "+v"(pos), "+v"(iv) : "n"(N) : "memory");
}
// ... with a wave-uniform (SGPR) base, a per-lane 32-bit offset and an immediate offset (-4096..4095): four chunks at a fixed
// stride cost one base computation instead of four 64-bit per-lane addresses
template <int OFF>
__device__ __forceinline__ u32x4 async_ld16_s(uint32_t voff, const void *sbase) {
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(v) : "v"(voff), "s"(sbase), "n"(OFF) : "memory");
    return v;
}
template <int OFF>
__device__ __forceinline__ uint32_t async_ld4_s(uint32_t voff, const void *sbase) {
    uint32_t v;
    asm volatile("global_load_dword %0, %1, %2 offset:%3" : "=v"(v) : "v"(voff), "s"(sbase), "n"(OFF) : "memory");
    return v;
}
__device__ __forceinline__ uint32_t async_ld4(const void *ptr) {
    uint32_t v;
    asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(ptr) : "memory");
    return v;
}

constexpr int BH = kScanBandRows;  // mask rows per band
constexpr int RR = 16;             // code ring rows = BH + 4: exactly what one band's probes touch
constexpr int RO = 16;             // occupancy ring rows
constexpr int NT = 256;     // threads of the single-mask kernel = the waves that stage rows
constexpr int NW = NT / 64;
constexpr int NT_BATCH = 1024;  // the batch kernel adds 4 waves that only probe (its LDS footprint allows 2-3 workgroups per CU)
constexpr int QCAP = 512;  // queue entries (one per non-black group); overflow falls back to in-place conversion
constexpr int QW = QCAP / (NT / 64);  // ... per wave
constexpr int MAXC = 4;    // chunks per wave and band whose mask data is fetched asynchronously at the top of the band
constexpr int RPW = BH / NW;  // new rows per wave and band
constexpr int RDUP = 4;      // mask batches: ring slots 0..3 are mirrored in slots RR..RR+3, so that a probe's three rows
                             // (slot, slot+2, slot+4) never wrap and the host can precompute LDS offsets per mask pixel
constexpr int BLKC = 32;     // mask batches: chunks per block of work drawn from the LDS counter (<= 32)
static_assert(RR >= BH + 4 && (RR & (RR - 1)) == 0 && RO >= RR && BH % NW == 0, "ring geometry");

// class lookup by the three strict comparisons: idx = 4*(r>g) + 8*(g>b) + 16*(b>r) selects a nibble.
// 1 BR, 2 BG, 3 GB, 4 GR, 5 RG, 6 RB (AbstractColorDepthSearchAlgorithm.java:195-225); 0 = none.
__device__ __forceinline__ uint32_t hue_code(float r, float g, float b, float thr) {
    float mx = __builtin_fmaxf(__builtin_fmaxf(r, g), b);
    float md = __builtin_amdgcn_fmed3f(r, g, b);
    float x = md * __builtin_amdgcn_rcpf(mx);
    uint32_t idx = (r > g ? 4u : 0u) | (g > b ? 8u : 0u) | (b > r ? 16u : 0u);
    uint32_t cls = (0x03125460u >> idx) & 7u;
    uint32_t c = (cls << 29) | (__float_as_uint(x) >> 1);
    return (mx > thr && mx > md) ? c : 0u;
}

__device__ __forceinline__ uint4 convert4(uint32_t a, uint32_t b, uint32_t c, float thr) {
    uint4 o;
    o.x = hue_code((float)(a & 255), (float)((a >> 8) & 255), (float)((a >> 16) & 255), thr);
    o.y = hue_code((float)(a >> 24), (float)(b & 255), (float)((b >> 8) & 255), thr);
    o.z = hue_code((float)((b >> 16) & 255), (float)(b >> 24), (float)(c & 255), thr);
    o.w = hue_code((float)((c >> 8) & 255), (float)((c >> 16) & 255), (float)(c >> 24), thr);
    return o;
}

struct Lds {
    uint32_t *ring;   // [RR][rp]   codes (raw groups between a row's commit and its conversion); row pitch rp = wc + 4, see ScanParams
    uint32_t *occ;    // [RR][occw] one bit per 4-pixel group (32 groups per word): group non-black
    uint16_t *queue;  // [NW][QW]   per wave: ring index / 4 of the non-black groups it staged and still has to convert
    int *next;        // [2]        mask batches: next block of chunks to hand out, double-buffered by band parity
    int *cnt;         // counters
};

// Staging of target rows (rows relative to m0-halo).  A row is K = ceil(gpr/64) segments of 64 groups (12 bytes each):
// lane l of segment k loads group 64k+l, so every load instruction reads 768 contiguous bytes from a wave-uniform row
// pointer; all K loads of a row are in flight together (the lane offsets are clamped, not branched, for the lanes beyond
// the window).  The raw group goes to the ring as (a,b,c,0) -- a black group is already its all-zero code --, the
// segment's ballot IS its two occupancy words, and every non-black group is queued for the conversion that follows.
// Per-lane constants of the staging, computed once per kernel: for segment k the byte offset of this lane's group inside
// an image row, the dword offset of its slot inside a ring row, and whether the lane's data is real.
// Lanes beyond the window (or beyond the image columns) re-read the window's first in-image group: no extra traffic;
// lanes beyond the window also re-write that group's slot with the same value (a benign duplicate), lanes inside the
// window but outside the image columns write zeros.
template <int KMAX>
struct LaneConst {
    uint32_t ldoff[KMAX];  // byte offset inside the image row
    int wr4[KMAX];         // dword offset inside the ring row
    uint32_t own;          // bit k: group 64k+lane is inside window and image (data counts)
    uint32_t zero;         // bit k: inside the window but outside the image columns (write zeros)
    unsigned long long ownmask[KMAX];  // wave-uniform: ballot of the own bits of segment k
    bool anyzero;                      // wave-uniform: some lane of some segment has its zero bit set
};

template <int KMAX>
__device__ __forceinline__ LaneConst<KMAX> make_lane_const(const ScanParams &p, int lane) {
    LaneConst<KMAX> lc;
    lc.own = lc.zero = 0;
    const int gfirst = p.cx0 < 0 ? (-p.cx0) >> 2 : 0;  // first window group inside the image
#pragma unroll
    for (int k = 0; k < KMAX; k++) {
        const int gi = 64 * k + lane;
        const int x0 = p.cx0 + 4 * gi;
        const bool inwin = gi < p.gpr;
        const bool incol = inwin && x0 >= 0 && x0 + 4 <= p.wpad;
        lc.ldoff[k] = (uint32_t)((incol ? x0 : p.cx0 + 4 * gfirst) * 3);
        lc.wr4[k] = 4 * (inwin ? gi : gfirst);
        lc.own |= incol ? (1u << k) : 0u;
        lc.zero |= (inwin && !incol) ? (1u << k) : 0u;
    }
#pragma unroll
    for (int k = 0; k < KMAX; k++) lc.ownmask[k] = __ballot((lc.own >> k) & 1);
    lc.anyzero = __ballot(lc.zero != 0) != 0;
    return lc;
}

// a_issue: the KMAX loads of ring row r (asm: the caller waits with wait_rows before reading w).
template <int KMAX>
__device__ __forceinline__ void a_issue(const ScanParams &p, const uint8_t *img, const LaneConst<KMAX> &lc, int r, u32x3 (&w)[KMAX]) {
    int row = r + p.m0 - p.halo;
    row = row < 0 ? 0 : (row >= p.H ? p.H - 1 : row);  // rows outside the image are loaded (clamped) and ignored
    const uint8_t *rowp = img + (uint32_t)(row * p.pitch);  // wave-uniform
#pragma unroll
    for (int k = 0; k < KMAX; k++)
        asm volatile("global_load_dwordx3 %0, %1, %2 nt" : "=v"(w[k]) : "v"(lc.ldoff[k]), "s"(rowp) : "memory");
}

// a_commit: registers -> ring (raw) / occupancy words / this wave's queue of non-black groups (qn = entries so far).
template <int KMAX, bool DUP>
__device__ __forceinline__ void a_commit(const ScanParams &p, const Lds &L, const LaneConst<KMAX> &lc, int r, int wave, int lane,
                                         const u32x3 (&w)[KMAX], int &qn) {
    constexpr int OCCW = 2 * KMAX;  // occupancy words per ring row (segments beyond the window stay zero)
    const int row = r + p.m0 - p.halo;
    const bool rowok = row >= 0 && row < p.H;  // wave-uniform
    const int slot = r & (RR - 1);
    uint32_t *ringrow = L.ring + slot * p.rp;
    const bool dup = DUP && slot < RDUP;  // wave-uniform: this row is also kept in slot RR + slot
    const int dupoff = RR * p.rp;
    unsigned long long bal[KMAX];
    int total = 0;
#pragma unroll
    for (int k = 0; k < KMAX; k++) {  // (a segment wholly beyond the window re-writes the window's first group: benign)
        uint32_t a = w[k].x, b = w[k].y, c = w[k].z;
        if (!rowok || lc.anyzero) {  // wave-uniform and rare: rows / columns outside the image are black
            const bool keep = rowok && !((lc.zero >> k) & 1);
            a = keep ? a : 0u, b = keep ? b : 0u, c = keep ? c : 0u;
        }
        *reinterpret_cast<uint4 *>(ringrow + lc.wr4[k]) = make_uint4(a, b, c, 0);  // black group == zero codes
        if (dup) *reinterpret_cast<uint4 *>(ringrow + dupoff + lc.wr4[k]) = make_uint4(a, b, c, 0);
        bal[k] = __ballot((a | b | c) != 0) & lc.ownmask[k];
        total += __popcll(bal[k]);
    }
    // the segments' ballots ARE the occupancy words of the row: lane k stores segment k's
    {
        unsigned long long mine = bal[0];
#pragma unroll
        for (int k = 1; k < KMAX; k++) mine = lane == k ? bal[k] : mine;
        if (lane < KMAX) *reinterpret_cast<uint2 *>(L.occ + (r & (RO - 1)) * OCCW + 2 * lane) = make_uint2((uint32_t)mine, (uint32_t)(mine >> 32));
    }
    if (total != 0 && !(p.debug & 8)) {  // wave-uniform
        uint16_t *wq = L.queue + wave * QW;
        int base = qn;
#pragma unroll
        for (int k = 0; k < KMAX; k++) {
            if (bal[k] != 0) {  // wave-uniform
                if ((bal[k] >> lane) & 1) {
                    const int my = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal[k] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal[k], 0));
                    const int dst = slot * p.rp + lc.wr4[k];
                    if (my < QW) {
                        wq[my] = (uint16_t)((dst >> 2) | (dup ? 0x8000 : 0));
                    } else {  // queue full (dense target): convert in place now
                        const uint4 cv = convert4(w[k].x, w[k].y, w[k].z, p.data_thr);
                        *reinterpret_cast<uint4 *>(L.ring + dst) = cv;
                        if (dup) *reinterpret_cast<uint4 *>(L.ring + dst + dupoff) = cv;
                    }
                }
                base += __popcll(bal[k]);
            }
        }
        qn = base;
    }
}

// convert_own: the wave converts, in place, the groups it queued itself (its LDS writes and reads execute in order, so no
// barrier is needed between a_commit and this; the workgroup barrier that follows publishes the codes to the probes).
__device__ __forceinline__ void convert_own(const ScanParams &p, const Lds &L, int wave, int lane, int qn) {
    const uint16_t *wq = L.queue + wave * QW;
    const int n = qn < QW ? qn : QW;
    for (int e = lane; e < n; e += 64) {
        const uint32_t q = wq[e];
        const int idx = (int)(q & 0x7fffu) << 2;
        const uint4 raw = *reinterpret_cast<const uint4 *>(L.ring + idx);
        const uint4 cv = convert4(raw.x, raw.y, raw.z, p.data_thr);
        *reinterpret_cast<uint4 *>(L.ring + idx) = cv;
        if (q & 0x8000u) *reinterpret_cast<uint4 *>(L.ring + idx + RR * p.rp) = cv;  // mirrored slot (mask batches)
    }
}

// Staging without prefetch (band 0): wave w loads, commits and converts rows ra+w, ra+w+NW, ...; two rows' loads
// are in flight together (band 0 is BH + 4 = 16 rows = 4 per wave; four together would spill).
template <int KMAX, bool DUP>
__device__ __forceinline__ void stage_a(const ScanParams &p, const uint8_t *img, const Lds &L, const LaneConst<KMAX> &lc, int ra, int n,
                                        int wave, int lane) {
    int qn = 0;
    for (int r0 = ra + wave; r0 < ra + n; r0 += 2 * NW) {  // wave-uniform
        u32x3 w[2][KMAX];
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int r = r0 + j * NW;
            a_issue<KMAX>(p, img, lc, r < ra + n ? r : r0, w[j]);
        }
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int r = r0 + j * NW;
            wait_rows<KMAX>(w[j]);
            if (r < ra + n) a_commit<KMAX, DUP>(p, L, lc, r, wave, lane, w[j], qn);
        }
    }
    if (!(p.debug & 2)) convert_own(p, L, wave, lane, qn);
}

__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// XY2: xyShift == 2 (9 shifts) else 0 (1 shift).  MIRROR: also the mirrored orientation.
// MULTI: several masks share the staged rows (counts by wave ballot into LDS) else one mask with
// per-lane counters kept in registers for the whole target.
//
// Per band of BH mask rows: the band's new target rows are staged (the staging wave converts what it wrote), barrier,
// probe, barrier.  The code ring holds exactly the BH + 4 rows one band needs (38 KB for the benchmark window): four
// workgroups share a CU and cover each other's memory latency and barrier waits.
template <bool XY2, bool MIRROR, bool MULTI, int KMAX>
__global__ __launch_bounds__(MULTI ? NT_BATCH : NT, KMAX <= 3 ? (MULTI ? 8 : 4) : 1) void scan_kernel(ScanParams p) {  // one mask: 4 waves/SIMD, <= 128 VGPRs
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    constexpr int NS = XY2 ? 9 : 1;
    constexpr int NSX = XY2 ? 3 : 1;
    constexpr int NO = MIRROR ? 2 : 1;
    constexpr int NC = NS * NO;
    Lds L;
    L.ring = lds;
    L.occ = L.ring + (RR + (MULTI ? RDUP : 0)) * p.rp;
    constexpr int OCCW = 2 * KMAX;
    constexpr int NTH = MULTI ? NT_BATCH : NT;
    L.queue = reinterpret_cast<uint16_t *>(L.occ + RO * OCCW);
    L.next = reinterpret_cast<int *>(L.queue + QCAP);
    L.cnt = L.next + 2;  // MULTI: [n_masks][NC]; single: [NC]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int halo = XY2 ? 2 : 0;
    const int m1rel = p.nrows;  // mask rows are [0, nrows) relative to m0
    const LaneConst<KMAX> lc = make_lane_const<KMAX>(p, lane);

    for (int64_t t = blockIdx.x; t < p.count; t += gridDim.x) {
        const uint8_t *img = p.lib + (size_t)((p.debug & 64) ? (t & 7) : t) * p.img_stride;  // debug 64: timing with L2-resident rows
        const int ncnt = MULTI ? p.n_masks * NC : NC;
        for (int i = threadIdx.x; i < ncnt; i += NTH) L.cnt[i] = 0;
        if (threadIdx.x < 2) L.next[threadIdx.x] = 0;
        for (int i = threadIdx.x; i < RO * OCCW; i += NTH) L.occ[i] = 0;
        int acc[NC];
#pragma unroll
        for (int k = 0; k < NC; k++) acc[k] = 0;
        __syncthreads();

        // band b = mask rows [b*BH, (b+1)*BH) needs target rows [b*BH, (b+1)*BH + 2*halo) (relative to m0-halo).
        // Band 0 is staged directly.  Every later band adds BH = NW new rows, one per wave: at the top of band b the
        // wave issues (asynchronously) the mask data of band b and then the loads of ITS row of band b+1, which stay
        // in flight during the probes of band b; after the band's last barrier the row is committed to the ring.
        // A launch may cover only the bands [band_begin, band_end) of the masks (band slices of a mask batch: the records of
        // one slice stay L2-resident while every workgroup of the launch works on it; the per-transform counts are carried
        // from slice to slice in p.counts).
        int band = p.band_begin;
        const int yb0 = band * BH;
        int staged = ((yb0 + BH < m1rel) ? yb0 + BH : m1rel) + 2 * halo;
        const bool stager = !MULTI || wave < NW;  // wave-uniform: the batch kernel's extra waves only probe
        if (stager) stage_a<KMAX, MULTI>(p, img, L, lc, yb0, staged - yb0, wave, lane);
        for (int y0 = yb0; y0 < m1rel && band < p.band_end; y0 += BH, band++) {
            const int y1 = (y0 + BH < m1rel && band + 1 < p.band_end) ? y0 + BH : m1rel;
            const int qsel = band & 1;
            // mask data of this wave's first MAXC chunks of the band (chunks are padded to 64 pixels: pixel index =
            // chunk * 64 + lane); a wave without a chunk re-reads the band's first chunk and ignores it
            const uint32_t cbeg0 = p.bandptr[band], cend0 = p.bandptr[band + 1];
            u32x4 pch[MAXC], piv[MAXC];
            uint32_t ppos[MAXC];
            if (!MULTI) {
                // chunks cbeg0 + wave + k*NW, k < MAXC, without clamping (the arrays have slack): one scalar base per array
                // (two for the 4 KB-strided interval records) and immediate offsets
                static_assert(MAXC == 4 && NW == 4, "immediate offsets below");
                const uint32_t c0 = cbeg0 + wave;
                const uint4 *dbase = p.chunks + c0;
                const uint32_t *pbase = p.px_pos + (size_t)c0 * 64;
                const uint4 *ibase1 = p.px_iv + (size_t)(c0 + NW) * 64, *ibase3 = p.px_iv + (size_t)(c0 + 3 * NW) * 64;
                const uint32_t zero = 0, l4 = (uint32_t)lane * 4u, l16 = (uint32_t)lane * 16u;
                pch[0] = async_ld16_s<0>(zero, dbase);
                ppos[0] = async_ld4_s<0>(l4, pbase);
                piv[0] = async_ld16_s<-4096>(l16, ibase1);
                pch[1] = async_ld16_s<64>(zero, dbase);
                ppos[1] = async_ld4_s<1024>(l4, pbase);
                piv[1] = async_ld16_s<0>(l16, ibase1);
                pch[2] = async_ld16_s<128>(zero, dbase);
                ppos[2] = async_ld4_s<2048>(l4, pbase);
                piv[2] = async_ld16_s<-4096>(l16, ibase3);
                pch[3] = async_ld16_s<192>(zero, dbase);
                ppos[3] = async_ld4_s<3072>(l4, pbase);
                piv[3] = async_ld16_s<0>(l16, ibase3);
            }
            // this wave's RPW rows of the next band (clamped and ignored where there is none)
            const int next_end = (y1 < m1rel) ? ((y1 + BH < m1rel) ? y1 + BH : m1rel) + 2 * halo : staged;
            u32x3 pw[RPW][KMAX];
            if (!MULTI) {  // (the batch kernel stages after the probes instead: its staging is amortised over the masks)
                const int r0 = staged + wave, row0 = r0 + p.m0 - p.halo;  // this wave's first new row: ring row / image row
                if (r0 + (RPW - 1) * NW < next_end && row0 >= 0 && row0 + (RPW - 1) * NW < p.H) {
                    // the usual case (all rows exist and lie inside the image): one row pointer, constant strides
                    const uint8_t *rowp = img + (uint32_t)(row0 * p.pitch);
                    const uint32_t step = (uint32_t)(NW * p.pitch);
#pragma unroll
                    for (int j = 0; j < RPW; j++) {
#pragma unroll
                        for (int k = 0; k < KMAX; k++)
                            asm volatile("global_load_dwordx3 %0, %1, %2 nt" : "=v"(pw[j][k]) : "v"(lc.ldoff[k]), "s"(rowp) : "memory");
                        rowp += step;
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < RPW; j++) {
                        const int my_row = r0 + j * NW;
                        a_issue<KMAX>(p, img, lc, my_row < next_end ? my_row : staged - 1, pw[j]);
                    }
                }
            }
            __syncthreads();  // the band's rows are staged and converted (by the waves that loaded them)
            if (MULTI && threadIdx.x == 0) L.next[qsel ^ 1] = 0;
            // ---- phase P: the mask data has arrived when at most the RPW*KMAX row loads are outstanding
            if (!MULTI) {
#pragma unroll
                for (int k = 0; k < MAXC; k++) wait_mask<RPW * KMAX>(pch[k], ppos[k], piv[k]);
            }
            // occ2: bit 0 = the chunk's plain neighbourhood holds a non-black group (or cannot be tested), bit 1 = mirrored
            auto probe_chunk = [&](const uint32_t occ2, const uint32_t pos, const uint4 iv, int *ctr) {
                const bool do_plain = occ2 & 1, do_mirror = MIRROR && (occ2 & 2);
                if (!(do_plain || do_mirror)) return;
                const int cpl = pos & 2047, cmi = (pos >> 11) & 2047, yrel = pos >> 22;
                // all LDS reads of one orientation are issued before the first compare (one wait instead of one per row)
                auto probe9 = [&](int col, int sgn, int *c9) {
                    uint32_t cv[NS];
#pragma unroll
                    for (int iy = 0; iy < NSX; iy++) {
                        const int rel = yrel + (XY2 ? 2 * iy : 0);  // = yrel + halo + dy
                        const uint32_t *rowp = L.ring + __umul24((uint32_t)(rel & (RR - 1)), (uint32_t)p.rp) + col;  // v_mul_u32_u24: full rate
#pragma unroll
                        for (int ix = 0; ix < NSX; ix++) cv[ix * NSX + iy] = rowp[XY2 ? sgn * 2 * (ix - 1) : 0];
                    }
#pragma unroll
                    for (int k = 0; k < NS; k++) c9[k] += (((cv[k] - iv.x) <= iv.y) | ((cv[k] - iv.z) <= iv.w)) ? 1 : 0;
                };
                if (do_plain) probe9(cpl, 1, ctr);
                if (do_mirror) probe9(cmi, -1, ctr + NS);
            };
            if (!MULTI) {
                // one mask: the band's chunks are dealt to the waves; counters stay in registers for the whole target
                if (!(p.debug & 1)) {
                    // occupancy of this wave's (up to MAXC) prefetched chunks in ONE test: lane k looks at the plain region of
                    // chunk k, lane 32+k at its mirrored region, against the band's occupancy = the OR of the ring rows'
                    // occupancy words (lane w holds word w; the two words a region spans come by ds_bpermute)
                    uint32_t bo = 0;
                    if (lane < OCCW) {
#pragma unroll
                        for (int r = 0; r < RO; r++) bo |= L.occ[r * OCCW + lane];
                    }
                    const int kk = lane & 31;
                    // rounds of MAXC chunks per wave: the first round's records were prefetched at the top of the band; a mask
                    // with more chunks than that requests the next MAXC together and waits for them (the band's rows have long
                    // landed by then), instead of crawling through them one load latency at a time
                    for (uint32_t cbase = cbeg0 + wave; cbase < cend0; cbase += MAXC * NW) {  // wave-uniform
                        if (cbase != cbeg0 + wave) {
#pragma unroll
                            for (int k = 0; k < MAXC; k++) {
                                uint32_t ci = cbase + k * NW;
                                ci = ci < cend0 ? ci : cbase;
                                pch[k] = async_ld16(p.chunks + ci);
                                ppos[k] = async_ld4(p.px_pos + (size_t)ci * 64 + lane);
                                piv[k] = async_ld16(p.px_iv + (size_t)ci * 64 + lane);
                            }
#pragma unroll
                            for (int k = 0; k < MAXC; k++) wait_mask<0>(pch[k], ppos[k], piv[k]);
                        }
                        uint32_t dz = pch[0].z, dw = pch[0].w;
#pragma unroll
                        for (int k = 1; k < MAXC; k++) {
                            dz = kk == k ? pch[k].z : dz;
                            dw = kk == k ? pch[k].w : dw;
                        }
                        const uint32_t desc = (lane < 32 || !MIRROR) ? dz : dw;
                        const int g0 = desc & 511, g1 = (desc >> 9) & 511;
                        const uint32_t wlo = __shfl(bo, g0 >> 5, 64), whi = __shfl(bo, (g0 >> 5) + 1, 64);
                        const unsigned long long bits = (((unsigned long long)whi << 32) | wlo) >> (g0 & 31);
                        const int width = g1 - g0 + 1;
                        const unsigned long long wm = width >= 64 ? ~0ull : ((1ull << width) - 1ull);
                        const bool hit = kk < MAXC && (MIRROR || lane < 32) && (!(desc >> 31) || (bits & wm) != 0 || (p.debug & 4));
                        const unsigned long long bal = __ballot(hit);
#pragma unroll
                        for (int k = 0; k < MAXC; k++) {
                            if (cbase + k * NW < cend0)
                                probe_chunk((uint32_t)((bal >> k) & 1) | (uint32_t)(((bal >> (32 + k)) & 1) << 1), ppos[k],
                                            make_uint4(piv[k].x, piv[k].y, piv[k].z, piv[k].w), acc);
                        }
                    }
                }
            } else if (!(p.debug & 1)) {
                // Mask batch.  The chunks of ALL masks of the launch are stored band-major (band, mask, chunk), so one
                // band's work is the contiguous chunk range [cbeg0, cend0).  Waves draw blocks of BLKC chunks from an
                // LDS counter; lane l tests chunk l's plain region, lane 32+l its mirrored region, against the BAND's
                // occupancy (the OR of the 16 ring rows' occupancy words, one word per lane, fetched by ds_bpermute)
                // -- one ballot tells which chunks and orientations need probing at all.  Only those chunks' pixel
                // records are loaded (the next occupied chunk's while the current one is probed), matches are
                // counted with s_bcnt1 of the compare masks into wave-uniform (SGPR) counters, which are flushed to the
                // mask's LDS counters when the mask changes.
                uint32_t bo = 0;
                if (lane < OCCW) {
#pragma unroll
                    for (int r = 0; r < RO; r++) bo |= L.occ[r * OCCW + lane];
                }
                int sacc[NC];
#pragma unroll
                for (int k = 0; k < NC; k++) sacc[k] = 0;
                int curmask = -1;
                auto flush = [&]() {
                    if (curmask >= 0) {
                        int v = 0;
#pragma unroll
                        for (int k = 0; k < NC; k++) asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(v) : "s"(__builtin_amdgcn_readfirstlane(sacc[k])), "n"(k));
                        if (lane < NC && v) atomicAdd(&L.cnt[curmask * NC + lane], v);
#pragma unroll
                        for (int k = 0; k < NC; k++) sacc[k] = 0;
                    }
                };
                const int l32 = lane & 31;  // lanes 0..31 test plain regions, lanes 32..63 mirrored ones
                for (;;) {
                    int blk = 0;
                    if (lane == 0) blk = atomicAdd(&L.next[qsel], 1);
                    blk = __builtin_amdgcn_readfirstlane(blk);
                    const uint32_t base = cbeg0 + (uint32_t)BLKC * (uint32_t)blk;
                    if (base >= cend0) break;
                    const uint32_t ci = base + l32;
                    const bool valid = l32 < BLKC && ci < cend0;
                    const uint4 d = p.chunks[valid ? ci : cend0 - 1];
                    // region [g0, g1] of the band: at most 2 occupancy words (else the host cleared bit 31: never skip)
                    const uint32_t desc = (lane < 32 || !MIRROR) ? d.z : d.w;
                    const int g0 = desc & 511, g1 = (desc >> 9) & 511;
                    const uint32_t wlo = __shfl(bo, g0 >> 5, 64), whi = __shfl(bo, (g0 >> 5) + 1, 64);
                    const unsigned long long bits = (((unsigned long long)whi << 32) | wlo) >> (g0 & 31);
                    const int width = g1 - g0 + 1;
                    const unsigned long long wm = width >= 64 ? ~0ull : ((1ull << width) - 1ull);
                    const bool hit = valid && (MIRROR || lane < 32) && (!(desc >> 31) || (bits & wm) != 0 || (p.debug & 4));
                    const unsigned long long bal = __ballot(hit);
                    const uint32_t bpl = (uint32_t)bal, bmi = (uint32_t)(bal >> 32);
                    uint32_t todo = bpl | bmi;
                    if (!todo) continue;
                    // two register sets (A/B) alternate so that the next occupied chunk's records are in flight while the
                    // current chunk is probed, without a register copy (which would wait for the load just issued)
                    auto process = [&](int i, uint32_t pos, const uint4 &iv) {
                        const uint32_t dx = (uint32_t)__builtin_amdgcn_readlane((int)d.x, i);
                        const int g = (dx >> 8) & 255;
                        if (g != curmask) {
                            flush();
                            curmask = g;
                        }
                        // pos = two 16-bit dword offsets into the ring (plain | mirrored << 16) of the lane's first probe:
                        // (slot of its first row) * rp + column - shift; the other probes are +2, +4 dwords and +2, +4 rows
                        // (no wrap: slots 0..3 are mirrored behind the ring)
                        auto probe9s = [&](uint32_t off, bool mirrored, int *sc) {
                            const uint32_t *b0 = L.ring + off;
                            uint32_t cv[NS];
#pragma unroll
                            for (int iy = 0; iy < NSX; iy++) {
#pragma unroll
                                for (int ix = 0; ix < NSX; ix++)
                                    cv[ix * NSX + iy] = b0[(XY2 ? 2 * iy : 0) * p.rp + (XY2 ? 2 * (mirrored ? 2 - ix : ix) : 0)];
                            }
                            // both intervals, always: a mask pixel whose hue class has a neighbour class usually has a second
                            // interval (negative gaps pass, so it is wide); the OR is taken on the wave masks (s_or_b64)
#pragma unroll
                            for (int k = 0; k < NS; k++)
                                sc[k] += __popcll(__ballot((cv[k] - iv.x) <= iv.y) | __ballot((cv[k] - iv.z) <= iv.w));
                        };
                        if ((bpl >> i) & 1) probe9s(pos & 0xffffu, false, sacc);
                        if (MIRROR && ((bmi >> i) & 1)) probe9s(pos >> 16, true, sacc + NS);
                    };
                    // wave-uniform chunk base (kept in SGPRs) + lane: global_load with a scalar base address
                    auto ld_pos = [&](int i) { return (p.px_pos + (size_t)(base + (uint32_t)i) * 64)[(uint32_t)lane]; };
                    auto ld_iv = [&](int i) { return (p.px_iv + (size_t)(base + (uint32_t)i) * 64)[(uint32_t)lane]; };
                    int iA = __builtin_ctz(todo), iB;
                    uint32_t posA = ld_pos(iA), posB;
                    uint4 ivA = ld_iv(iA), ivB;
                    for (;;) {
                        todo &= todo - 1;
                        iB = todo ? __builtin_ctz(todo) : iA;  // next occupied chunk (or this one again: no branch around the loads)
                        posB = ld_pos(iB);
                        ivB = ld_iv(iB);
                        process(iA, posA, ivA);
                        if (!todo) break;
                        todo &= todo - 1;
                        iA = todo ? __builtin_ctz(todo) : iB;
                        posA = ld_pos(iA);
                        ivA = ld_iv(iA);
                        process(iB, posB, ivB);
                        if (!todo) break;
                    }
                }
                flush();
            }
            __syncthreads();  // all probes of this band are done: the staging of the next band overwrites the oldest BH ring rows
            // ---- staging of the next band: commit the rows loaded at the top of this band, convert their non-black groups
            if (MULTI) {
                // batch kernel: no prefetch registers held across the probes (they would cost a third of the waves a SIMD can
                // hold); the 4 staging waves load, commit and convert the next band's rows here
                if (stager && next_end > staged) stage_a<KMAX, MULTI>(p, img, L, lc, staged, next_end - staged, wave, lane);
            } else {
                int qn = 0;
#pragma unroll
                for (int j = 0; j < RPW; j++) {
                    const int my_row = staged + wave + j * NW;
                    wait_rows<KMAX>(pw[j]);
                    if (my_row < next_end && !(p.debug & 16)) a_commit<KMAX, MULTI>(p, L, lc, my_row, wave, lane, pw[j], qn);
                }
                if (!(p.debug & 2)) convert_own(p, L, wave, lane, qn);
            }
            staged = next_end;
        }
        if (!MULTI) {
#pragma unroll
            for (int k = 0; k < NC; k++) {
                int v = wave_sum(acc[k]);
                if (lane == 0 && v) atomicAdd(&L.cnt[k], v);
            }
        }
        __syncthreads();
        const int nm = MULTI ? p.n_masks : 1;
        const bool first_slice = p.band_begin == 0, last_slice = p.band_end >= p.nbands;
        for (int g = threadIdx.x; g < nm; g += NTH) {
            int c[NC];
#pragma unroll
            for (int k = 0; k < NC; k++) c[k] = L.cnt[g * NC + k];
            int32_t *co = p.counts ? p.counts + ((size_t)g * p.out_stride + p.out_offset + t) * CDS_N_TRANSFORMS : nullptr;
            if (!first_slice) {  // add the counts of the earlier band slices (p.counts is mandatory when slicing)
#pragma unroll
                for (int k = 0; k < NS; k++) {
                    c[k] += co[k];
                    if (MIRROR) c[NS + k] += co[9 + k];
                }
            }
            if (last_slice) {
                int best = 0, bm = 0;
#pragma unroll
                for (int k = 0; k < NS; k++) best = c[k] > best ? c[k] : best;
                if (MIRROR) {
#pragma unroll
                    for (int k = 0; k < NS; k++) bm = c[NS + k] > bm ? c[NS + k] : bm;
                }
                int32_t packed = best;
                if (MIRROR && bm > best) packed = (int32_t)((uint32_t)bm | 0x80000000u);
                p.out[(size_t)g * p.out_stride + p.out_offset + t] = packed;
            }
            if (co) {
                if (first_slice) {
                    for (int k = 0; k < CDS_N_TRANSFORMS; k++) co[k] = 0;
                }
#pragma unroll
                for (int k = 0; k < NS; k++) {
                    co[k] = c[k];
                    if (MIRROR) co[9 + k] = c[NS + k];
                }
            }
        }
        __syncthreads();  // cnt, queue and ring are reused by the next target
    }
}

// kernel instantiations exist for 2, 3, 5 and 8 segments (64 groups = 256 pixels each) per staged row
static int scan_kmax(int nseg) { return nseg <= 2 ? 2 : nseg <= 3 ? 3 : nseg <= 5 ? 5 : 8; }

size_t scan_lds_bytes(int wc, int n_masks, bool batch, bool xy2, bool mirror) {
    const int nc = (xy2 ? 9 : 1) * (mirror ? 2 : 1);
    const int rp = ring_pitch(wc), occw = 2 * scan_kmax((wc / 4 + 63) / 64);
    return (size_t)(RR + (batch ? RDUP : 0)) * rp * 4 + (size_t)RO * occw * 4 + (size_t)QCAP * 2 + 2 * sizeof(int) +
           (size_t)(n_masks > 1 ? n_masks : 1) * nc * sizeof(int);
}

template <bool XY2, bool MIRROR, bool MULTI>
static int launch_one(cds_context *ctx, const ScanParams &p, hipStream_t stream) {
    const size_t lds_bytes = scan_lds_bytes(p.wc, p.n_masks, MULTI, XY2, MIRROR);
    if (lds_bytes > ctx->lds_per_block) return fail(CDS_ERR_UNSUPPORTED, "column window too wide for LDS");
    int per_cu = 64;  // many short workgroups: targets differ in sparsity, the tail balances itself
    if (const char *e = getenv("CDS_SCAN_WGS_PER_CU")) per_cu = atoi(e);
    const int64_t g64 = (int64_t)ctx->n_cus * per_cu;
    int grid = (int)(p.count < g64 ? p.count : g64);
    if (grid < 1) grid = 1;
    auto kern = p.nseg <= 2 ? scan_kernel<XY2, MIRROR, MULTI, 2> : p.nseg <= 3 ? scan_kernel<XY2, MIRROR, MULTI, 3>
                : p.nseg <= 5 ? scan_kernel<XY2, MIRROR, MULTI, 5> : scan_kernel<XY2, MIRROR, MULTI, 8>;
    CDS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(MULTI ? NT_BATCH : NT), lds_bytes, stream, p);
    CDS_HIP(hipGetLastError());
    return CDS_OK;
}

int launch_scan(cds_context *ctx, const ScanParams &p, bool xy2, bool mirror, hipStream_t stream) {
    const bool multi = p.batch != 0;  // the mask set's records are in the batch layout (even if this launch holds one mask)
    if (xy2) {
        if (mirror) return multi ? launch_one<true, true, true>(ctx, p, stream) : launch_one<true, true, false>(ctx, p, stream);
        return multi ? launch_one<true, false, true>(ctx, p, stream) : launch_one<true, false, false>(ctx, p, stream);
    }
    if (mirror) return multi ? launch_one<false, true, true>(ctx, p, stream) : launch_one<false, true, false>(ctx, p, stream);
    return multi ? launch_one<false, false, true>(ctx, p, stream) : launch_one<false, false, false>(ctx, p, stream);
}

// v_rcp_f32 of 0..255 as this device computes it -- the host's interval builder uses the same values.
__global__ void rcp_table_kernel(float *out) {
    int i = threadIdx.x;
    out[i] = __builtin_amdgcn_rcpf((float)i);
}

int init_rcp_table(cds_context *ctx) {
    if (ctx->layout_stamp != sizeof(cds_context))
        return failf(CDS_ERR_HIP, "libcds_hip.so is linked from objects built against different revisions of cds_internal.hpp "
                                 "(context layout %zu bytes here, %zu where it was created): rebuild every object", sizeof(cds_context), ctx->layout_stamp);
    float *d = nullptr;
    CDS_HIP(dev_malloc(&d, 256 * sizeof(float)));
    // the buffer starts as NaNs (0xff bytes): a kernel that never ran -- no code object for this device, a launch refused -- is told apart
    // from a table that came back wrong
    hipError_t e = hipMemsetAsync(d, 0xff, 256 * sizeof(float), ctx->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(rcp_table_kernel, dim3(1), dim3(256), 0, ctx->stream, d);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(ctx->rcp, d, 256 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    hipFree(d);
    if (e != hipSuccess) return hip_fail(e, "reciprocal table kernel (is libcds_hip.so built for this device's architecture?)");
    // v_rcp_f32 is exact for powers of two: 1, 0.5 and 1/128 come back as themselves on every device that ran the kernel
    if (!(ctx->rcp[1] == 1.0f && ctx->rcp[2] == 0.5f && ctx->rcp[128] == 0.0078125f))
        return failf(CDS_ERR_HIP, "the reciprocal table kernel did not write its table (rcp[1], rcp[2], rcp[3], rcp[128] read back as %g, %g, %g, %g; "
                                 "NaN = never written)", (double)ctx->rcp[1], (double)ctx->rcp[2], (double)ctx->rcp[3], (double)ctx->rcp[128]);
    return CDS_OK;
}

}  // namespace cds
