// cds_shape.hip -- kernel 2: the GradientAreaGap ("shape") scorer for gfx950, its per-mask
// preparation, and the circular max filter both need.
//
// Reference: API/cds/Shape2DMatchColorDepthSearchAlgorithm.java:28-42 (PIXEL_GAP_OP), :150-245;
// API/cds/GradientAreaGapUtils.java:18-197 (slice numbers); API/cds/ColorDepthSearchAlgorithmProviderFactory.java:91-125
// (mask preparation); API/imageprocessing/ImageTransformation.java:353-572 (max filter, makeLineRadii).
//
// Design: the reference evaluates ~2*W*H pixels per pair through nested boxed lambdas and searches the
// colour LUT twice per pixel.  Here the slice number S(colour) is a table lookup (S factorises:
// sliceGap(q,z) = |S(q) - S(z)|), the mask side is precomputed once per mask into one 16-bit plane per
// orientation (S(q) | queryMask<<9 | highExpression<<10, mirror and ROI already applied), and one
// workgroup streams the rows of ONE (mask,target) pair: zgap rows are converted to S(z) in LDS so the
// mirrored orientation can read them reversed; both orientations are reduced in the same pass.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "cds_internal.hpp"
#include "cds_lut.hpp"

namespace cds {

static bool in_rects_host(const cds_rect *r, int n, int x, int y) {
    for (int i = 0; i < n; i++)
        if (x >= r[i].x0 && x < r[i].x1 && y >= r[i].y0 && y < r[i].y1) return true;
    return false;
}

// ---------------------------------------------------------------------------------------------
// Slice-number table: S(first, second, sextant) - 1 for every colour with first > 0.
// sextant = 2*firstChannel + (second is the later of the two remaining channels), channels R,G,B = 0,1,2
// with the reference's ">=" precedence (GradientAreaGapUtils.java:31-91).
// ---------------------------------------------------------------------------------------------
static int slice_in_lut(int lo, int hi, double ratio) {  // GradientAreaGapUtils.java:131-197
    int slice = 0;
    double mingap = 1000;
    for (int i = lo; i <= hi; i++) {
        double R = kDepthLutHost[3 * i], G = kDepthLutHost[3 * i + 1], B = kDepthLutHost[3 * i + 2], lr = 0;
        if (B > R && B > G) {
            if (R > G) lr = R / B; else if (G > R) lr = G / B;
        } else if (G > R && G > B) {
            if (R > B) lr = R / G; else if (B > R) lr = B / G;
        } else if (R > G && R > B) {
            if (G > B) lr = G / R; else if (B > G) lr = B / R;
        }
        if (lr == ratio) return i + 1;
        double gp = std::fabs(ratio - lr);
        if (gp < mingap) { mingap = gp; slice = i + 1; }
    }
    return slice;
}
// sextant -> LUT index range (GradientAreaGapUtils.java:107-129)
static const int kSextLo[6] = {171, 213, 128, 86, 0, 30};   // R/G, R/B, G/R, G/B, B/R, B/G
static const int kSextHi[6] = {212, 255, 170, 127, 29, 85};

static inline int sextant_of(int r, int g, int b, int *first, int *second) {
    if (r >= g && r >= b) { *first = r; if (g >= b) { *second = g; return 0; } *second = b; return 1; }
    if (g >= r && g >= b) { *first = g; if (r >= b) { *second = r; return 2; } *second = b; return 3; }
    *first = b; if (r >= g) { *second = r; return 4; } *second = g; return 5;
}

static void build_slice_table(std::vector<uint8_t> &tab) {
    tab.assign((size_t)6 * 256 * 256, 0);
    for (int s = 0; s < 6; s++)
        for (int first = 1; first < 256; first++)
            for (int second = 0; second <= first; second++) {
                double ratio = (double)second / (double)first;
                tab[((size_t)s * 256 + first) * 256 + second] = (uint8_t)(slice_in_lut(kSextLo[s], kSextHi[s], ratio) - 1);
            }
}

static const std::vector<uint8_t> &host_slice_table() {
    static std::vector<uint8_t> t;
    static std::once_flag once;
    std::call_once(once, [] { build_slice_table(t); });
    return t;
}
static inline int host_slice_number(int r, int g, int b) {
    if ((r | g | b) == 0) return 0;
    int first, second;
    int s = sextant_of(r, g, b, &first, &second);
    return host_slice_table()[((size_t)s * 256 + first) * 256 + second] + 1;
}

static int ensure_slice_lut(cds_context *ctx) {
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (ctx->d_slice_lut) return CDS_OK;
    const std::vector<uint8_t> &t = host_slice_table();
    CDS_HIP(dev_malloc(&ctx->d_slice_lut, t.size()));
    ctx->owns_lut = true;   // (also for a pair-slot context cloned before its parent had the table)
    CDS_HIP(hipMemcpy(ctx->d_slice_lut, t.data(), t.size(), hipMemcpyHostToDevice));
    return CDS_OK;
}

// makeLineRadii -- ImageTransformation.java:549-572.  hw[dy + k] = half width of kernel row dy.
static int line_radii(double radius_arg, std::vector<int> &hw) {
    double radius = radius_arg;
    if (radius_arg >= 1.5 && radius_arg < 1.75) radius = 1.75;
    else if (radius_arg >= 2.5 && radius_arg < 2.85) radius = 2.85;
    int r2 = (int)(radius * radius) + 1;
    int k = (int)(std::sqrt(r2 + 1e-10));
    hw.assign((size_t)2 * k + 1, 0);
    hw[(size_t)k] = k;
    for (int y = 1; y <= k; y++) {
        int dx = (int)(std::sqrt(r2 - y * y + 1e-10));
        hw[(size_t)(k - y)] = dx;
        hw[(size_t)(k + y)] = dx;
    }
    return k;
}

// ---------------------------------------------------------------------------------------------
// Circular max filter (per channel, out-of-image ignored) of a pitched RGB image, restricted to the
// output box [bx0,bx1) x [by0,by1).  Optional pre-ops on the source: black out rectangles, black out
// pixels whose channels are all <= thr (ColorTransformation.mask).  One thread per output pixel; the
// source box + k fits L2, so this small helper is left un-tiled.
// ---------------------------------------------------------------------------------------------
struct MaxFilterParams {
    const uint8_t *src;
    uint8_t *dst;
    int pitch, W, H, k;
    int bx0, by0, bx1, by1;
    int thr;  // < 0: no threshold masking
    int nrects;
    cds_rect rects[CDS_MAX_RECTS];
    const int *hw;  // device, 2k+1
};

__device__ __forceinline__ bool dev_in_rects(const cds_rect *r, int n, int x, int y) {
    for (int i = 0; i < n; i++)
        if (x >= r[i].x0 && x < r[i].x1 && y >= r[i].y0 && y < r[i].y1) return true;
    return false;
}

__global__ __launch_bounds__(256) void prefilter_kernel(MaxFilterParams p) {
    // dst = src with rectangles and sub-threshold pixels blacked (whole image)
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= p.W) return;
    const uint8_t *s = p.src + (size_t)y * p.pitch + 3 * x;
    uint8_t *d = p.dst + (size_t)y * p.pitch + 3 * x;
    int r = s[0], g = s[1], b = s[2];
    bool kill = dev_in_rects(p.rects, p.nrects, x, y) || (p.thr >= 0 && r <= p.thr && g <= p.thr && b <= p.thr);
    d[0] = kill ? 0 : r; d[1] = kill ? 0 : g; d[2] = kill ? 0 : b;
}

__global__ __launch_bounds__(256) void max_filter_kernel(MaxFilterParams p) {
    int x = p.bx0 + blockIdx.x * blockDim.x + threadIdx.x, y = p.by0 + blockIdx.y;
    if (x >= p.bx1) return;
    int mr = 0, mg = 0, mb = 0;
    for (int dy = -p.k; dy <= p.k; dy++) {
        int yy = y + dy;
        if (yy < 0 || yy >= p.H) continue;
        int h = p.hw[dy + p.k];
        int xa = x - h < 0 ? 0 : x - h, xb = x + h >= p.W ? p.W - 1 : x + h;
        const uint8_t *s = p.src + (size_t)yy * p.pitch + 3 * xa;
        for (int xx = xa; xx <= xb; xx++, s += 3) {
            mr = max(mr, (int)s[0]); mg = max(mg, (int)s[1]); mb = max(mb, (int)s[2]);
        }
    }
    uint8_t *d = p.dst + (size_t)y * p.pitch + 3 * x;
    d[0] = mr; d[1] = mg; d[2] = mb;
}

// host image (pitch 3W) -> optional clear/threshold -> max filter -> host image.  Output outside the box
// that can be non-black (source bbox grown by k) is black by construction.
static int max_filter_host(cds_context *ctx, const uint8_t *src, int W, int H, double radius, int thr,
                           const cds_rect *rects, int nrects, uint8_t *dst) {
    CDS_HIP(hipSetDevice(ctx->device));
    std::vector<int> hw;
    const int k = line_radii(radius, hw);
    const int pitch = 3 * W;  // contiguous rows: the kernels below access bytes, and one plain transfer each way replaces
                              // a host-side pitched copy (one small transfer per row: ~4 ms per image)
    const size_t bytes = (size_t)pitch * H;
    uint8_t *d_a = nullptr, *d_b = nullptr;
    int *d_hw = nullptr;
    hipStream_t st = ctx->stream;
    int rc = CDS_OK;
    auto cleanup = [&]() { if (d_a) hipFree(d_a); if (d_b) hipFree(d_b); if (d_hw) hipFree(d_hw); };
#define MF_HIP(call) do { hipError_t _e = (call); if (_e != hipSuccess) { cleanup(); return hip_fail(_e, #call); } } while (0)
    MF_HIP(dev_malloc(&d_a, bytes));
    MF_HIP(dev_malloc(&d_b, bytes));
    MF_HIP(dev_malloc(&d_hw, hw.size() * sizeof(int)));
    MF_HIP(hipMemsetAsync(d_b, 0, bytes, st));
    MF_HIP(hipMemcpyAsync(d_a, src, bytes, hipMemcpyHostToDevice, st));
    MF_HIP(hipMemcpyAsync(d_hw, hw.data(), hw.size() * sizeof(int), hipMemcpyHostToDevice, st));
    MaxFilterParams p{};
    p.pitch = pitch; p.W = W; p.H = H; p.k = k; p.thr = thr; p.nrects = nrects; p.hw = d_hw;
    for (int i = 0; i < nrects; i++) p.rects[i] = rects[i];
    if (thr >= 0 || nrects > 0) {
        p.src = d_a; p.dst = d_a;  // in place, pixel-wise
        hipLaunchKernelGGL(prefilter_kernel, dim3((W + 255) / 256, H), dim3(256), 0, st, p);
        MF_HIP(hipGetLastError());
    }
    // bounding box of the (pre-filtered) source on the host copy is cheap enough
    int bx0 = W, by0 = H, bx1 = 0, by1 = 0;
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const uint8_t *q = src + 3 * ((size_t)y * W + x);
            if ((q[0] | q[1] | q[2]) == 0) continue;
            bx0 = std::min(bx0, x); bx1 = std::max(bx1, x + 1); by0 = std::min(by0, y); by1 = std::max(by1, y + 1);
        }
    if (bx1 > bx0) {
        p.bx0 = std::max(0, bx0 - k); p.bx1 = std::min(W, bx1 + k);
        p.by0 = std::max(0, by0 - k); p.by1 = std::min(H, by1 + k);
        p.src = d_a; p.dst = d_b;
        hipLaunchKernelGGL(max_filter_kernel, dim3((p.bx1 - p.bx0 + 255) / 256, p.by1 - p.by0), dim3(256), 0, st, p);
        MF_HIP(hipGetLastError());
    }
    MF_HIP(hipMemcpyAsync(dst, d_b, bytes, hipMemcpyDeviceToHost, st));
    MF_HIP(hipStreamSynchronize(st));
#undef MF_HIP
    cleanup();
    return rc;
}

// rgbToGrayNoGammaCorrection(pv, 255) -- ColorTransformation.java:40-54
static inline int gray_nogamma(int r, int g, int b) {
    if ((r | g | b) == 0) return 0;
    double rw = 1 / 3., gw = 1 / 3., bw = 1 / 3.;
    float maxGray = 255;
    volatile double s = r * rw;
    s = s + g * gw;
    s = s + b * bw;
    s = s + 0.5;
    return (int)((maxGray / 255) * s);
}

// ---------------------------------------------------------------------------------------------
// kernel 2
// ---------------------------------------------------------------------------------------------
constexpr int kPlaneS = 511, kPlaneQM = 1 << 9, kPlaneQH = 1 << 10;

struct ShapeParams {
    const uint8_t *rgb;
    const uint16_t *grad, *zslice;  // zslice: (slice number - 1) | max channel << 8 per zgap pixel (gradient layout)
    const uint8_t *zgap;            // the zgap images themselves (RGB layout) and the slice-number table: read instead of zslice by the
    const uint8_t *slice_lut;       // ZRGB variant of the list kernel (no conversion pass; a target that serves few pairs)
    size_t img_stride, grad_stride;
    int pitch, gpitch, W, H, wp;  // wp: plane row length (u16 elements), multiple of 4
    const uint16_t *planes;       // [n_masks][2][H][wp]
    const int *boxes;             // [n_masks][8]: ya, yb (rows with any plane bit), gap columns [gx0,gx1), QH columns [hx0,hx1)
    const uint8_t *rowflags;      // [n_masks][H]: bit0 row has gap bits (S(q) or queryMask), bit1 row has high-expression bits
    int thr, mirror, nrects;
    cds_rect rects[CDS_MAX_RECTS];
    const uint32_t *glist, *hlist;  // per mask: the 4-pixel groups with gap bits / with high-expression bits (either orientation)
    const ushort4 *gl_a, *gl_b;     // the two orientations' plane values of the glist groups (parallel to glist: coalesced reads)
    const uint32_t *list_off;       // [n_masks + 1][2]: first entry of mask m in glist / hlist
    const int32_t *pair_mask;
    const int64_t *pair_slot;     // < 0: pair has no variants, skipped (host reports -1,-1)
    int64_t n_pairs;
    int64_t *out;                 // [n_pairs][4]: gap0, high0, gap1, high1
};

__device__ __forceinline__ int dev_slice(const uint8_t *lut, int r, int g, int b, int thr) {
    // ColorTransformation.mask(thr): all channels <= thr -> black -> no slice
    if (r <= thr && g <= thr && b <= thr) return 0;
    if ((r | g | b) == 0) return 0;
    int first, second, s;
    if (r >= g && r >= b) { first = r; if (g >= b) { second = g; s = 0; } else { second = b; s = 1; } }
    else if (g >= r && g >= b) { first = g; if (r >= b) { second = r; s = 2; } else { second = b; s = 3; } }
    else { first = b; if (r >= g) { second = r; s = 4; } else { second = g; s = 5; } }
    return lut[((s * 256 + first) << 8) + second] + 1;
}

__device__ __forceinline__ int gap_of(int pl, int sz, int g) {
    const int sq = pl & kPlaneS;
    const int d = sq > sz ? sq - sz : sz - sq;
    int gap = (sq != 0 && sz != 0 && d >= 80) ? d - 40 : ((pl & kPlaneQM) ? g : 0);
    return gap > 3 ? gap : 0;
}

// zgap RGB -> zslice plane, once per uploaded / generated variant image: the colour-LUT search of
// GradientAreaGapUtils.calculateSliceGap becomes one table gather here and a byte compare in the scorer.
struct ZsliceParams {
    const uint8_t *zgap;        // slot 0 of the library
    uint16_t *zslice;
    const uint8_t *slice_lut;
    size_t img_stride, grad_stride;
    int pitch, gpitch, H, gpr;  // gpr: 4-pixel groups per row (covers the padded row)
    const int32_t *slots;       // the `count` slots to convert
    int64_t count;
};

__global__ __launch_bounds__(256) void zslice_kernel(ZsliceParams p) {
    const int64_t items = (int64_t)p.count * p.H * p.gpr;
    for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += (int64_t)gridDim.x * blockDim.x) {
        const int gi = (int)(it % p.gpr);
        const int64_t rowi = it / p.gpr;
        const int y = (int)(rowi % p.H);
        const int64_t img = p.slots[rowi / p.H];
        const uint32_t *src = reinterpret_cast<const uint32_t *>(p.zgap + (size_t)img * p.img_stride + (size_t)y * p.pitch + (size_t)gi * 12);
        const uint32_t a = src[0], b = src[1], c = src[2];
        const int px[4][3] = {{(int)(a & 255), (int)((a >> 8) & 255), (int)((a >> 16) & 255)},
                              {(int)(a >> 24), (int)(b & 255), (int)((b >> 8) & 255)},
                              {(int)((b >> 16) & 255), (int)(b >> 24), (int)(c & 255)},
                              {(int)((c >> 8) & 255), (int)((c >> 16) & 255), (int)(c >> 24)}};
        uint16_t o[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int mx = max(px[j][0], max(px[j][1], px[j][2]));
            const int s = mx ? dev_slice(p.slice_lut, px[j][0], px[j][1], px[j][2], -1) : 1;  // 1..256
            o[j] = (uint16_t)((s - 1) | (mx << 8));
        }
        uint16_t *dst = reinterpret_cast<uint16_t *>(reinterpret_cast<uint8_t *>(p.zslice) + (size_t)img * p.grad_stride + (size_t)y * p.gpitch) + 4 * gi;
        *reinterpret_cast<ushort4 *>(dst) = make_ushort4(o[0], o[1], o[2], o[3]);
    }
}

// S(z) of a zslice entry under ColorTransformation.mask(thr): black if all channels <= thr.
__device__ __forceinline__ int zs_of(int e, int thr) {
    const int mx = e >> 8;
    return (mx > thr && mx > 0) ? (e & 255) + 1 : 0;
}

__global__ __launch_bounds__(256) void gradscore_kernel(ShapeParams p) {
    // One workgroup scores one (mask, target) pair; one row per wave at a time, lanes over 4-pixel groups, no barriers in
    // the row loop.  The two orientation planes of a group decide what else is fetched: the target pixels when a
    // high-expression bit is set (the wide ring around the mask: most groups), gradient + zslice (plain and reversed) when
    // a gap bit is (the mask's own pixels: ~5 % of the groups).  The gap arithmetic is the expensive part and its groups
    // are scattered, so a wave does not run it in place with a few lanes active: it appends the group to its own LDS
    // queue and runs the gap code over 64 queued groups at a time, all lanes busy.
    __shared__ long long red[4][4];
    __shared__ uint32_t gq[4][128];  // per wave: row << 9 | group of the queued gap groups
    __shared__ uint32_t hq[4][128];  // ... and of the queued high-expression groups (QH bit set, target group not black)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool even_w = (p.W & 1) == 0;
    for (int64_t pi = blockIdx.x; pi < p.n_pairs; pi += gridDim.x) {
        const int64_t slot = p.pair_slot[pi];
        if (slot < 0) continue;  // uniform per block
        const int m = p.pair_mask[pi];
        const uint8_t *timg = p.rgb + (size_t)slot * p.img_stride;
        const uint8_t *zimg = reinterpret_cast<const uint8_t *>(p.zslice) + (size_t)slot * p.grad_stride;
        const uint8_t *gimg = reinterpret_cast<const uint8_t *>(p.grad) + (size_t)slot * p.grad_stride;
        const uint16_t *pl0 = p.planes + (size_t)m * 2 * p.H * p.wp;
        const uint16_t *pl1 = pl0 + (size_t)p.H * p.wp;
        const int *bx = p.boxes + 8 * m;
        const int ya = bx[0], yb = bx[1];
        const int gq0 = bx[2] >> 2, gq1 = (bx[3] + 3) >> 2;                             // gap window, in groups
        const int aq0 = min(bx[2], bx[4]) >> 2, aq1 = (max(bx[3], bx[5]) + 3) >> 2;     // gap U high-expression window
        const uint8_t *rf = p.rowflags + (size_t)m * p.H;
        unsigned gap0 = 0, gap1 = 0;  // per lane < 2^31 for H*W <= 2M pixels (gap <= 65535 per pixel)
        int high0 = 0, high1 = 0;
        int qn = 0, hn = 0;  // wave-uniform: queued gap / high-expression groups
        // the gap arithmetic of the first n (<= 64) queued groups, one per lane
        auto run_gaps = [&](int n) {
            if (lane < n) {
                const uint32_t e = gq[wave][lane];
                const int y = (int)(e >> 9), x0 = 4 * (int)(e & 511);
                const ushort4 qa = *reinterpret_cast<const ushort4 *>(pl0 + (size_t)y * p.wp + x0);
                ushort4 qb = make_ushort4(0, 0, 0, 0);
                if (p.mirror) qb = *reinterpret_cast<const ushort4 *>(pl1 + (size_t)y * p.wp + x0);
                const int pa[4] = {qa.x, qa.y, qa.z, qa.w}, pb[4] = {qb.x, qb.y, qb.z, qb.w};
                const uint16_t *zrow = reinterpret_cast<const uint16_t *>(zimg + (size_t)y * p.gpitch);
                const ushort4 g = *reinterpret_cast<const ushort4 *>(gimg + (size_t)y * p.gpitch + (size_t)x0 * 2);
                const ushort4 z = *reinterpret_cast<const ushort4 *>(zrow + x0);
                const int gg[4] = {g.x, g.y, g.z, g.w}, zz[4] = {z.x, z.y, z.z, z.w};
                int zm[4] = {0, 0, 0, 0};  // zslice at W-1-x for x = x0 .. x0+3
                if (p.mirror) {
                    const int xm = p.W - 4 - x0;  // element index of the reversed run (may be < 0 at the far right)
                    if (even_w && xm >= 0) {
                        const uint2 r = *reinterpret_cast<const uint2 *>(zrow + xm);  // xm even: 4-byte aligned
                        zm[3] = r.x & 0xffff; zm[2] = r.x >> 16; zm[1] = r.y & 0xffff; zm[0] = r.y >> 16;
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; j++)
                            if (p.W - 1 - x0 - j >= 0) zm[j] = zrow[p.W - 1 - x0 - j];
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if (x0 + j >= p.W) break;
                    gap0 += gap_of(pa[j], zs_of(zz[j], p.thr), gg[j]);
                    if (p.mirror) gap1 += gap_of(pb[j], zs_of(zm[j], p.thr), gg[j]);
                }
            }
        };
        // the high-expression count of the first n (<= 64) queued groups, one per lane
        auto run_high = [&](int n) {
            if (lane < n) {
                const uint32_t e = hq[wave][lane];
                const int y = (int)(e >> 9), x0 = 4 * (int)(e & 511);
                const ushort4 qa = *reinterpret_cast<const ushort4 *>(pl0 + (size_t)y * p.wp + x0);
                ushort4 qb = make_ushort4(0, 0, 0, 0);
                if (p.mirror) qb = *reinterpret_cast<const ushort4 *>(pl1 + (size_t)y * p.wp + x0);
                const int pa[4] = {qa.x, qa.y, qa.z, qa.w}, pb[4] = {qb.x, qb.y, qb.z, qb.w};
                const uint32_t *src = reinterpret_cast<const uint32_t *>(timg + (size_t)y * p.pitch + (size_t)x0 * 3);
                const uint32_t a = src[0], b = src[1], c = src[2];
                const int mx[4] = {max(max((int)(a & 255), (int)((a >> 8) & 255)), (int)((a >> 16) & 255)),
                                   max(max((int)(a >> 24), (int)(b & 255)), (int)((b >> 8) & 255)),
                                   max(max((int)((b >> 16) & 255), (int)(b >> 24)), (int)(c & 255)),
                                   max(max((int)((c >> 8) & 255), (int)((c >> 16) & 255)), (int)(c >> 24))};
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int x = x0 + j;
                    if (x >= p.W) break;
                    const bool sig = mx[j] > p.thr && !dev_in_rects(p.rects, p.nrects, x, y);
                    high0 += (sig && (pa[j] & kPlaneQH)) ? 1 : 0;
                    high1 += (sig && (pb[j] & kPlaneQH)) ? 1 : 0;
                }
            }
        };
        for (int y = ya + wave; y < yb; y += 4) {  // wave-uniform
            const int flags = rf[y];
            if (!flags) continue;
            const int q0 = (flags & 2) ? aq0 : gq0, q1 = (flags & 2) ? aq1 : gq1;
            for (int base = q0; base < q1; base += 64) {  // wave-uniform
                const int gi = base + lane;
                const int x0 = 4 * gi;
                ushort4 qa = make_ushort4(0, 0, 0, 0), qb = make_ushort4(0, 0, 0, 0);
                if (gi < q1) {
                    qa = *reinterpret_cast<const ushort4 *>(pl0 + (size_t)y * p.wp + x0);
                    if (p.mirror) qb = *reinterpret_cast<const ushort4 *>(pl1 + (size_t)y * p.wp + x0);
                }
                const int all = qa.x | qa.y | qa.z | qa.w | qb.x | qb.y | qb.z | qb.w;
                const bool has_gap = (all & (kPlaneS | kPlaneQM)) != 0;
                const unsigned long long bal = __ballot(has_gap);
                if (bal) {  // wave-uniform
                    if (has_gap) {
                        const int at = qn + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0));
                        gq[wave][at] = ((uint32_t)y << 9) | (uint32_t)gi;
                    }
                    qn += __popcll(bal);
                    if (qn >= 64) {
                        run_gaps(64);
                        const int rest = qn - 64;
                        uint32_t mv = 0;
                        if (lane < rest) mv = gq[wave][64 + lane];
                        if (lane < rest) gq[wave][lane] = mv;
                        qn = rest;
                    }
                }
                // high-expression groups: the target group is fetched here (12 bytes), the arithmetic only for the few that are
                // not black -- queued like the gap groups
                bool has_high = false;
                if (all & kPlaneQH) {
                    const uint32_t *src = reinterpret_cast<const uint32_t *>(timg + (size_t)y * p.pitch + (size_t)x0 * 3);
                    has_high = (src[0] | src[1] | src[2]) != 0;
                }
                const unsigned long long balh = __ballot(has_high);
                if (balh) {  // wave-uniform
                    if (has_high) {
                        const int at = hn + __builtin_amdgcn_mbcnt_hi((uint32_t)(balh >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)balh, 0));
                        hq[wave][at] = ((uint32_t)y << 9) | (uint32_t)gi;
                    }
                    hn += __popcll(balh);
                    if (hn >= 64) {
                        run_high(64);
                        const int rest = hn - 64;
                        uint32_t mv = 0;
                        if (lane < rest) mv = hq[wave][64 + lane];
                        if (lane < rest) hq[wave][lane] = mv;
                        hn = rest;
                    }
                }
            }
        }
        run_high(hn);  // hn < 64
        run_gaps(qn);  // qn < 64
        long long v[4] = {(long long)gap0, (long long)high0, (long long)gap1, (long long)high1};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            long long s = v[k];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
            if (lane == 0) red[wave][k] = s;
        }
        __syncthreads();
        if (threadIdx.x < 4) {
            long long s = 0;
            for (int w = 0; w < (int)(blockDim.x >> 6); w++) s += red[w][threadIdx.x];
            p.out[pi * 4 + threadIdx.x] = s;
        }
        __syncthreads();
    }
}

void zgap_written(cds_library *lib, int64_t first, int64_t count) {
    for (int64_t i = 0; i < count; i++) {
        lib->has_zgap[(size_t)(first + i)] = 1;
        lib->zslice_ok[(size_t)(first + i)] = 0;
    }
}

// The zslice entries of `slots` (distinct, with a zgap image) that are stale or were never made: one conversion launch over that
// list, on `st` (the scoring kernel follows on the same stream).  Called by a scoring call that has decided for the converted plane.
static int ensure_zslice(cds_library *lib, const std::vector<int64_t> &slots, hipStream_t st) {
    std::lock_guard<std::mutex> lk(lib->mu);
    std::vector<int32_t> todo;
    for (int64_t sl : slots)
        if (!lib->zslice_ok[(size_t)sl]) todo.push_back((int32_t)sl);
    if (todo.empty()) return CDS_OK;
    int rc = ensure_slice_lut(lib->ctx);
    if (rc) return rc;
    if (!lib->d_zslice) {
        const size_t bytes = (size_t)lib->capacity * lib->grad_stride;
        CDS_HIP(dev_malloc(&lib->d_zslice, bytes));
    }
    int32_t *d_slots = nullptr;
    CDS_HIP(hipMallocAsync((void **)&d_slots, todo.size() * sizeof(int32_t), st));
    hipError_t e = hipMemcpyAsync(d_slots, todo.data(), todo.size() * sizeof(int32_t), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);   // (`todo` is pageable host memory and leaves scope)
    if (e == hipSuccess) {
        ZsliceParams p{};
        p.zgap = lib->d_zgap;
        p.zslice = lib->d_zslice;
        p.slice_lut = lib->ctx->d_slice_lut;
        p.img_stride = lib->img_stride; p.grad_stride = lib->grad_stride;
        p.pitch = lib->pitch; p.gpitch = lib->gpitch; p.H = lib->H;
        p.gpr = std::min(lib->pitch / 12, lib->gpitch / 8);
        p.slots = d_slots;
        p.count = (int64_t)todo.size();
        const int64_t items = p.count * p.H * p.gpr;
        const int grid = (int)std::min<int64_t>((items + 255) / 256, (int64_t)lib->ctx->n_cus * 64);
        hipLaunchKernelGGL(zslice_kernel, dim3(grid), dim3(256), 0, st, p);
        e = hipGetLastError();
    }
    hipFreeAsync(d_slots, st);
    if (e != hipSuccess) return hip_fail(e, "zslice conversion");
    for (int32_t sl : todo) lib->zslice_ok[(size_t)sl] = 1;
    return CDS_OK;
}

}  // namespace cds

using namespace cds;

// ---------------------------------------------------------------------------------------------
// gradscore_list_kernel: the scorer over per-mask GROUP LISTS instead of a scan of the mask's plane windows.
// gradscore_kernel above finds its work by reading both orientation planes of every 4-pixel group of the mask's rows
// (~50 K groups, 800 KB from L2 per pair) although only ~25 K groups carry a high-expression bit and ~3.5 K a gap bit, and
// those are the same for the mask's 500 pairs.  The preparation therefore lists them once (prep_lists_kernel):
//   hlist entry = row << 17 | group << 8 | QH bits of the 4 pixels (plain) << 4 | QH bits (mirrored orientation)
//   glist entry = row << 9 | group
// and a pair is: (1) every hlist entry -> the target's 12-byte group; the non-black ones (a few percent) are queued and
// counted 64 at a time (max channel > threshold, not in a label rectangle, QH bit of the orientation); (2) every glist
// entry -> planes, gradient, zslice (plain and reversed) -> the gap arithmetic, all lanes busy, no queue.
// One workgroup per pair, the four waves take alternate 64-entry chunks.
// ---------------------------------------------------------------------------------------------
// S(z) of a zgap pixel under ColorTransformation.mask(thr), straight from its RGB bytes: what zs_of() reads out of a zslice entry.
// In two steps so that the table gathers of a group's pixels are independent loads issued together (dev_slice's branches would
// serialise them): slice_index = the table position dev_slice reads (same sextant rules, branch-free; valid for any colour, black
// reads entry 0), zs_pick = the value under the threshold rule.
__device__ __forceinline__ uint32_t slice_index(int r, int g, int b) {
    const bool R = r >= g && r >= b, G = !R && g >= r && g >= b;
    const int first = R ? r : (G ? g : b);
    const int c1 = R ? g : r, c2 = (R || G) ? b : g;   // R: (g, b)  G: (r, b)  B: (r, g) -- the first of the pair wins ties
    const bool pick = c1 >= c2;
    const int second = pick ? c1 : c2;
    const int sext = (R ? 0 : (G ? 2 : 4)) + (pick ? 0 : 1);
    return (uint32_t)(((sext * 256 + first) << 8) + second);
}
__device__ __forceinline__ int zs_pick(int lut_value, int r, int g, int b, int thr) {
    const int mx = max(r, max(g, b));
    return (mx > thr && mx > 0) ? lut_value + 1 : 0;
}
__device__ __forceinline__ int zs_rgb(const uint8_t *lut, int r, int g, int b, int thr) {
    return zs_pick(lut[slice_index(r, g, b)], r, g, b, thr);
}
// the four pixels of 12 packed RGB bytes (a, b, c): all four table gathers in flight together; out[k] = S(z) of pixel k
__device__ __forceinline__ void zs_group(const uint8_t *lut, uint32_t a, uint32_t b, uint32_t c, int thr, int (&out)[4]) {
    const int px[4][3] = {{(int)(a & 255), (int)((a >> 8) & 255), (int)((a >> 16) & 255)},
                          {(int)(a >> 24), (int)(b & 255), (int)((b >> 8) & 255)},
                          {(int)((b >> 16) & 255), (int)(b >> 24), (int)(c & 255)},
                          {(int)((c >> 8) & 255), (int)((c >> 16) & 255), (int)(c >> 24)}};
    int v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = lut[slice_index(px[k][0], px[k][1], px[k][2])];
#pragma unroll
    for (int k = 0; k < 4; k++) out[k] = zs_pick(v[k], px[k][0], px[k][1], px[k][2], thr);
}
typedef uint32_t u32_any_align __attribute__((aligned(1)));

// ZRGB = false: S(z) comes from the zslice plane (one conversion pass per uploaded zgap, zslice_kernel: worth it when a target
// serves many pairs).  ZRGB = true: the gap groups read the zgap image's own bytes and look the slice number up here (12 instead
// of 8 bytes per group and orientation plus up to 8 table gathers from L2, but NO pass over the ~685 K pixels of every zgap
// image -- a selected target of BASELINE configs[4] serves 2.2 pairs x ~3.5 K gap groups).  Same sums either way (tests).
template <bool ZRGB>
__global__ __launch_bounds__(256) void gradscore_list_kernel(ShapeParams p) {
    __shared__ long long red[4][4];
    __shared__ uint32_t hq[4][128];  // per wave: the queued high-expression entries whose target group is not black
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool even_w = (p.W & 1) == 0;
    for (int64_t pi = blockIdx.x; pi < p.n_pairs; pi += gridDim.x) {
        const int64_t slot = p.pair_slot[pi];
        if (slot < 0) continue;  // uniform per block
        const int m = p.pair_mask[pi];
        const uint8_t *timg = p.rgb + (size_t)slot * p.img_stride;
        const uint8_t *zimg = ZRGB ? p.zgap + (size_t)slot * p.img_stride : reinterpret_cast<const uint8_t *>(p.zslice) + (size_t)slot * p.grad_stride;
        const uint8_t *gimg = reinterpret_cast<const uint8_t *>(p.grad) + (size_t)slot * p.grad_stride;
        const uint32_t g0 = p.list_off[2 * m], g1 = p.list_off[2 * m + 2], h0 = p.list_off[2 * m + 1], h1 = p.list_off[2 * m + 3];
        unsigned gap0 = 0, gap1 = 0;  // per lane < 2^31 for H*W <= 2M pixels (gap <= 65535 per pixel)
        int high0 = 0, high1 = 0;
        int hn = 0;  // wave-uniform: queued high-expression entries
        auto run_high = [&](int n) {
            if (lane < n) {
                const uint32_t e = hq[wave][lane];
                const int y = (int)(e >> 17), x0 = 4 * (int)((e >> 8) & 511);
                const uint32_t *src = reinterpret_cast<const uint32_t *>(timg + (size_t)y * p.pitch + (size_t)x0 * 3);
                const uint32_t a = src[0], b = src[1], c = src[2];
                const int mx[4] = {max(max((int)(a & 255), (int)((a >> 8) & 255)), (int)((a >> 16) & 255)),
                                   max(max((int)(a >> 24), (int)(b & 255)), (int)((b >> 8) & 255)),
                                   max(max((int)((b >> 16) & 255), (int)(b >> 24)), (int)(c & 255)),
                                   max(max((int)((c >> 8) & 255), (int)((c >> 16) & 255)), (int)(c >> 24))};
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int x = x0 + j;
                    if (x >= p.W) break;
                    const bool sig = mx[j] > p.thr && !dev_in_rects(p.rects, p.nrects, x, y);
                    high0 += (sig && ((e >> (4 + j)) & 1u)) ? 1 : 0;
                    high1 += (sig && ((e >> j) & 1u)) ? 1 : 0;
                }
            }
        };
        // (1) high-expression groups
        for (uint32_t base = h0 + 64u * wave; base < h1; base += 256) {  // wave-uniform
            const uint32_t i = base + lane;
            bool nonblack = false;
            uint32_t e = 0;
            if (i < h1) {
                e = p.hlist[i];
                const int y = (int)(e >> 17), gi = (int)((e >> 8) & 511);
                const uint32_t *src = reinterpret_cast<const uint32_t *>(timg + (size_t)y * p.pitch + (size_t)gi * 12);
                nonblack = (src[0] | src[1] | src[2]) != 0;
            }
            const unsigned long long bal = __ballot(nonblack);
            if (bal) {  // wave-uniform
                if (nonblack) hq[wave][hn + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0))] = e;
                hn += __popcll(bal);
                if (hn >= 64) {
                    run_high(64);
                    const int rest = hn - 64;
                    uint32_t mv = 0;
                    if (lane < rest) mv = hq[wave][64 + lane];
                    if (lane < rest) hq[wave][lane] = mv;
                    hn = rest;
                }
            }
        }
        run_high(hn);  // hn < 64
        // (2) gap groups
        for (uint32_t base = g0 + 64u * wave; base < g1; base += 256) {  // wave-uniform
            const uint32_t i = base + lane;
            if (i < g1) {
                const uint32_t e = p.glist[i];
                const int y = (int)(e >> 9), x0 = 4 * (int)(e & 511);
                const ushort4 qa = p.gl_a[i], qb = p.gl_b[i];   // (the mirrored orientation's plane is all zero without mirroring)
                const int pa[4] = {qa.x, qa.y, qa.z, qa.w}, pb[4] = {qb.x, qb.y, qb.z, qb.w};
                const ushort4 g = *reinterpret_cast<const ushort4 *>(gimg + (size_t)y * p.gpitch + (size_t)x0 * 2);
                const int gg[4] = {g.x, g.y, g.z, g.w};
                int sz[4] = {0, 0, 0, 0}, szm[4] = {0, 0, 0, 0};   // S(z) at x0 .. x0+3, and at W-1-x for the mirrored orientation
                // gap_of() looks at S(z) only where the orientation's plane has a slice number S(q): a list entry usually carries gap
                // bits of ONE orientation (the other orientation's mask pixels lie elsewhere), so half of the z fetches never happen
                const bool need_a = ((pa[0] | pa[1] | pa[2] | pa[3]) & kPlaneS) != 0;
                const bool need_b = p.mirror && ((pb[0] | pb[1] | pb[2] | pb[3]) & kPlaneS) != 0;
                if (!ZRGB) {
                    const uint16_t *zrow = reinterpret_cast<const uint16_t *>(zimg + (size_t)y * p.gpitch);
                    if (need_a) {
                        const ushort4 z = *reinterpret_cast<const ushort4 *>(zrow + x0);
                        sz[0] = zs_of(z.x, p.thr), sz[1] = zs_of(z.y, p.thr), sz[2] = zs_of(z.z, p.thr), sz[3] = zs_of(z.w, p.thr);
                    }
                    if (need_b) {
                        int zm[4] = {0, 0, 0, 0};  // zslice at W-1-x for x = x0 .. x0+3
                        const int xm = p.W - 4 - x0;  // element index of the reversed run (may be < 0 at the far right)
                        if (even_w && xm >= 0) {
                            const uint2 r = *reinterpret_cast<const uint2 *>(zrow + xm);  // xm even: 4-byte aligned
                            zm[3] = r.x & 0xffff; zm[2] = r.x >> 16; zm[1] = r.y & 0xffff; zm[0] = r.y >> 16;
                        } else {
#pragma unroll
                            for (int j = 0; j < 4; j++)
                                if (p.W - 1 - x0 - j >= 0) zm[j] = zrow[p.W - 1 - x0 - j];
                        }
#pragma unroll
                        for (int j = 0; j < 4; j++) szm[j] = zs_of(zm[j], p.thr);
                    }
                } else {
                    // both orientations' bytes are requested before either is looked up (a lane that needs neither reads the row's
                    // first bytes: black or not, its sums ignore them), then all eight table gathers go out together
                    const uint8_t *zr = zimg + (size_t)y * p.pitch;
                    const int xm = p.W - 4 - x0;  // first pixel of the reversed run W-4-x0 .. W-1-x0 (< 0 at the far right: pixel by pixel)
                    const uint32_t *qa = reinterpret_cast<const uint32_t *>(zr + (need_a ? (size_t)x0 * 3 : 0));   // 4-byte aligned (x0 % 4 == 0)
                    const u32_any_align *qb = reinterpret_cast<const u32_any_align *>(zr + ((need_b && xm >= 0) ? (size_t)xm * 3 : 0));  // any alignment
                    const uint32_t a0 = qa[0], a1 = qa[1], a2 = qa[2];
                    const uint32_t b0 = qb[0], b1 = qb[1], b2 = qb[2];
                    int ta[4], tb[4];
                    zs_group(p.slice_lut, a0, a1, a2, p.thr, ta);
                    zs_group(p.slice_lut, b0, b1, b2, p.thr, tb);
                    if (need_a) { sz[0] = ta[0]; sz[1] = ta[1]; sz[2] = ta[2]; sz[3] = ta[3]; }
                    if (need_b) {
                        if (xm >= 0) {
                            szm[3] = tb[0]; szm[2] = tb[1]; szm[1] = tb[2]; szm[0] = tb[3];
                        } else {
#pragma unroll
                            for (int j = 0; j < 4; j++) {
                                const int xx = p.W - 1 - x0 - j;
                                if (xx >= 0) szm[j] = zs_rgb(p.slice_lut, zr[3 * xx], zr[3 * xx + 1], zr[3 * xx + 2], p.thr);
                            }
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if (x0 + j >= p.W) break;
                    gap0 += gap_of(pa[j], sz[j], gg[j]);
                    if (p.mirror) gap1 += gap_of(pb[j], szm[j], gg[j]);
                }
            }
        }
        long long v[4] = {(long long)gap0, (long long)high0, (long long)gap1, (long long)high1};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            long long sv = v[k];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) sv += __shfl_xor(sv, o, 64);
            if (lane == 0) red[wave][k] = sv;
        }
        __syncthreads();
        if (threadIdx.x < 4) {
            long long sv = 0;
            for (int w = 0; w < (int)(blockDim.x >> 6); w++) sv += red[w][threadIdx.x];
            p.out[pi * 4 + threadIdx.x] = sv;
        }
        __syncthreads();
    }
}

// The group lists of every mask from its two orientation planes.  FILL = false: count; FILL = true: write (a wave's entries
// of one row are contiguous and ascending; rows arrive in any order -- the scorer only sums).
struct ListParams {
    const uint16_t *planes;  // [n][2][H][wp]
    int H, wp, n;
    uint32_t *counts;        // [n][2]: gap groups, high-expression groups   (count pass; cursor in the fill pass)
    const uint32_t *list_off;  // [n + 1][2]
    uint32_t *glist, *hlist;
    ushort4 *gl_a, *gl_b;
};
template <bool FILL>
__global__ __launch_bounds__(256) void prep_lists_kernel(ListParams p) {
    const int lane = threadIdx.x & 63;
    const int64_t wid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wid >= (int64_t)p.n * p.H) return;  // wave-uniform
    const int m = (int)(wid / p.H), y = (int)(wid - (int64_t)m * p.H);
    const uint16_t *r0 = p.planes + ((size_t)m * 2 * p.H + y) * p.wp, *r1 = r0 + (size_t)p.H * p.wp;
    const int ngr = p.wp / 4;
    for (int g0 = 0; g0 < ngr; g0 += 64) {
        const int gi = g0 + lane;
        ushort4 qa = make_ushort4(0, 0, 0, 0), qb = make_ushort4(0, 0, 0, 0);
        if (gi < ngr) {
            qa = *reinterpret_cast<const ushort4 *>(r0 + 4 * gi);
            qb = *reinterpret_cast<const ushort4 *>(r1 + 4 * gi);
        }
        const int all = qa.x | qa.y | qa.z | qa.w | qb.x | qb.y | qb.z | qb.w;
        const bool has_gap = (all & (kPlaneS | kPlaneQM)) != 0, has_high = (all & kPlaneQH) != 0;
        const unsigned long long bg = __ballot(has_gap), bh = __ballot(has_high);
        if (bg) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(p.counts + 2 * m, (uint32_t)__popcll(bg));
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            if (FILL && has_gap) {
                const uint32_t at = p.list_off[2 * m] + base + __builtin_amdgcn_mbcnt_hi((uint32_t)(bg >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bg, 0));
                p.glist[at] = ((uint32_t)y << 9) | (uint32_t)gi;
                p.gl_a[at] = qa;
                p.gl_b[at] = qb;
            }
        }
        if (bh) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(p.counts + 2 * m + 1, (uint32_t)__popcll(bh));
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            if (FILL && has_high) {
                const uint32_t n0 = ((qa.x >> 10) & 1u) | ((qa.y >> 10) & 1u) << 1 | ((qa.z >> 10) & 1u) << 2 | ((qa.w >> 10) & 1u) << 3;
                const uint32_t n1 = ((qb.x >> 10) & 1u) | ((qb.y >> 10) & 1u) << 1 | ((qb.z >> 10) & 1u) << 2 | ((qb.w >> 10) & 1u) << 3;
                p.hlist[p.list_off[2 * m + 1] + base + __builtin_amdgcn_mbcnt_hi((uint32_t)(bh >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bh, 0))] =
                    ((uint32_t)y << 17) | ((uint32_t)gi << 8) | (n0 << 4) | n1;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Shape-mask preparation on the device (ColorDepthSearchAlgorithmProviderFactory.java:91-125).
// Q = clearRegion(mask); QM = gray(Q) > 2; S(q) = slice number; QH = (maxFilter20(Q) is black) && gray(maxFilter60(Q)) > 0.
// The two max filters are never materialised: "D20 black" <=> no pixel with a non-zero channel inside the r=20 kernel, and
// gray_nogamma(D60) > 0 <=> mr + mg + mb >= 2 for the channel maxima over the r=60 kernel ((int)(s/3. + 0.5) > 0 <=> s >= 2,
// checked for s = 1, 2 in the reference's double arithmetic: 0.8333 -> 0, 1.1667 -> 1) <=> some channel reaches 2, or two
// channels reach 1.  So seven binary dilations by ImageJ's kernel (rows dy = -k..k with half-widths hw[dy]) suffice, and a
// dilation is "some row of the kernel has a set pixel within hw[dy] of x" = a lookup of that row's horizontal-distance map.
// ---------------------------------------------------------------------------------------------
struct PrepParams {
    const uint8_t *masks;  // [n][H][3W]
    const uint8_t *roi;    // [H][W] or NULL
    int W, H, wp, n, mirror, nrects;
    int border;            // queryBorderSize: the frame every fold of the reference skips (LImage.java:89-113)
    cds_rect rects[CDS_MAX_RECTS];
    int k60, k20;
    const int *hw60, *hw20;
    uint8_t *flags;        // [n][H][W]  bit c: channel c >= 1, bit 3+c: channel c >= 2, bit 6: any channel >= 1  (of Q)
    uint8_t *hd;           // [n][7][H][W] horizontal distance to the nearest pixel with that flag in the same row, 255 = none within 62
    uint16_t *planes;      // [n][2][H][wp]
    int *acc;              // [n][16]: 0 ya 1 yb 2 gx0 3 gx1 4 hx0 5 hx1 (min / max accumulators), 8.. bbox of Q: bx0 by0 bx1 by1
    uint32_t *rowflags;    // [n][H]
    unsigned long long *sums;  // [n][2] sum QM, sum QH
    const uint8_t *slice_lut;
};
constexpr int kHdMax = 62;  // distances beyond the largest half-width (61) are all the same

__global__ __launch_bounds__(256) void prep_flags_kernel(PrepParams p) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, m = blockIdx.z;
    if (x >= p.W) return;
    const uint8_t *s = p.masks + ((size_t)m * p.H + y) * 3 * p.W + 3 * x;
    int r = s[0], g = s[1], b = s[2];
    if (dev_in_rects(p.rects, p.nrects, x, y)) r = g = b = 0;
    const int f = (r >= 1) | (g >= 1) << 1 | (b >= 1) << 2 | (r >= 2) << 3 | (g >= 2) << 4 | (b >= 2) << 5 | ((r | g | b) != 0) << 6;
    p.flags[((size_t)m * p.H + y) * p.W + x] = (uint8_t)f;
    if (f) {
        int *a = p.acc + (size_t)m * 16;
        atomicMin(a + 8, x); atomicMin(a + 9, y); atomicMax(a + 10, x + 1); atomicMax(a + 11, y + 1);
    }
}

__global__ __launch_bounds__(256) void prep_hdist_kernel(PrepParams p) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, m = blockIdx.z;
    if (x >= p.W) return;
    const int *a = p.acc + (size_t)m * 16;
    uint8_t *out = p.hd + (((size_t)m * 7) * p.H + y) * p.W + x;
    const size_t pstride = (size_t)p.H * p.W;
    int dist[7];
#pragma unroll
    for (int q = 0; q < 7; q++) dist[q] = 255;
    if (y >= a[9] && y < a[11] && x >= a[8] - kHdMax && x < a[10] + kHdMax) {  // rows / columns without signal keep 255
        const uint8_t *row = p.flags + ((size_t)m * p.H + y) * p.W;
        int found = 0;
        for (int d = 0; d < kHdMax && found != 0x7f; d++) {
            const int xl = x - d, xr = x + d;
            const int f = (xl >= 0 ? row[xl] : 0) | (xr < p.W ? row[xr] : 0);
            const int nf = f & ~found;
            if (nf) {
#pragma unroll
                for (int q = 0; q < 7; q++)
                    if ((nf >> q) & 1) dist[q] = d;
                found |= nf;
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 7; q++) out[(size_t)q * pstride] = (uint8_t)dist[q];
}

// rgbToGrayNoGammaCorrection(pv, 255) -- ColorTransformation.java:40-54 (doubles, no contraction: -ffp-contract=off)
__device__ __forceinline__ int dev_gray_nogamma(int r, int g, int b) {
    if ((r | g | b) == 0) return 0;
    const double w = 1 / 3.;
    double s = r * w;
    s = s + g * w;
    s = s + b * w;
    s = s + 0.5;
    return (int)((255.f / 255) * s);
}

__global__ __launch_bounds__(256) void prep_plane_kernel(PrepParams p) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, m = blockIdx.z;
    if (x >= p.W) return;
    int *a = p.acc + (size_t)m * 16;
    const int bx0 = a[8], by0 = a[9], bx1 = a[10], by1 = a[11];
    const uint8_t *s = p.masks + ((size_t)m * p.H + y) * 3 * p.W + 3 * x;
    int r = s[0], g = s[1], b = s[2];
    if (dev_in_rects(p.rects, p.nrects, x, y)) r = g = b = 0;
    const int qm0 = dev_gray_nogamma(r, g, b) > 2;                                  // Factory:108
    int qh = 0;
    if (bx1 > bx0 && x >= bx0 - p.k60 && x < bx1 + p.k60 && y >= by0 - p.k60 && y < by1 + p.k60) {
        const uint8_t *hd = p.hd + ((size_t)m * 7) * p.H * p.W + x;
        const size_t pstride = (size_t)p.H * p.W;
        bool d20black = true;                                                        // Factory:101-106
        for (int dy = -p.k20; dy <= p.k20 && d20black; dy++) {
            const int yy = y + dy;
            if (yy < by0 || yy >= by1) continue;
            if (hd[6 * pstride + (size_t)yy * p.W] <= p.hw20[dy + p.k20]) d20black = false;
        }
        if (d20black) {
            int got = 0;  // bit q: flag q occurs inside the r = 60 kernel
            for (int dy = -p.k60; dy <= p.k60 && !((got & 0x38) != 0 || __popc(got & 7) >= 2); dy++) {
                const int yy = y + dy;
                if (yy < by0 || yy >= by1) continue;
                const int h = p.hw60[dy + p.k60];
                const uint8_t *q = hd + (size_t)yy * p.W;
#pragma unroll
                for (int c = 0; c < 6; c++) got |= (q[c * pstride] <= h) << c;
            }
            qh = (got & 0x38) != 0 || __popc(got & 7) >= 2;
        }
    }
    // queryBorderSize (Factory:103): the max filters above still SEE the frame's pixels (the lazy query image keeps them), but
    // the reduce()d masks are 0 in the frame (toImageArray is a foldi over the inner rectangle) and the gap fold runs over
    // the inner rectangle only (combine4 inherits the query image's borders): a frame cell contributes nothing
    const bool frame = x < p.border || x >= p.W - p.border || y < p.border || y >= p.H - p.border;
    const int qm = frame ? 0 : qm0;
    if (frame) qh = 0;
    const int sq = (!frame && (r | g | b)) ? dev_slice(p.slice_lut, r, g, b, -1) : 0;
    const uint16_t v = (uint16_t)(sq | (qm ? kPlaneQM : 0) | (qh ? kPlaneQH : 0));
    // sums over the whole frame, before the ROI (Shape2DMatch...Test.java:53-54 checks them on the un-mirrored, un-masked images)
    const unsigned long long bm = __ballot(qm), bh = __ballot(qh);
    if ((threadIdx.x & 63) == 0) {
        if (bm) atomicAdd(p.sums + 2 * m, (unsigned long long)__popcll(bm));
        if (bh) atomicAdd(p.sums + 2 * m + 1, (unsigned long long)__popcll(bh));
    }
    if (!v) return;
    // identity orientation: value at x; mirrored orientation: the same value seen at W-1-x; the ROI (cleared in the excluded
    // regions, Factory:96-100) is applied after mirroring
    uint16_t *p0 = p.planes + ((size_t)m * 2) * p.H * p.wp, *p1 = p0 + (size_t)p.H * p.wp;
    uint32_t *rfl = p.rowflags + (size_t)m * p.H;
    auto put = [&](uint16_t *pl, int xx) {
        if (p.roi && (p.roi[(size_t)y * p.W + xx] == 0 || dev_in_rects(p.rects, p.nrects, xx, y))) return;
        pl[(size_t)y * p.wp + xx] = v;
        atomicMin(a + 0, y); atomicMax(a + 1, y + 1);
        uint32_t fl = 0;
        if (v & (kPlaneS | kPlaneQM)) { atomicMin(a + 2, xx); atomicMax(a + 3, xx + 1); fl |= 1; }
        if (v & kPlaneQH) { atomicMin(a + 4, xx); atomicMax(a + 5, xx + 1); fl |= 2; }
        atomicOr(rfl + y, fl);
    };
    put(p0, x);
    if (p.mirror) put(p1, p.W - 1 - x);
}

struct cds_shapeset {
    cds_context *ctx = nullptr;
    int W = 0, H = 0, wp = 0, n_masks = 0, thr = 0, mirror = 0, nrects = 0;
    cds_rect rects[CDS_MAX_RECTS];
    uint16_t *d_planes = nullptr;
    int *d_boxes = nullptr;
    uint8_t *d_rowflags = nullptr;
    uint32_t *d_glist = nullptr, *d_hlist = nullptr, *d_list_off = nullptr;  // group lists of gradscore_list_kernel
    ushort4 *d_gl_a = nullptr, *d_gl_b = nullptr;
    std::vector<int64_t> qm_sum, qh_sum;
    mutable std::mutex mu;
    mutable double last_ms = 0;
    mutable int last_launches = 0;
    mutable double last_convert_ms = 0;   // zgap -> zslice conversion of the last call (0 when it read the zgap bytes directly)
    mutable int last_mode = 0;            // 0: zgap bytes read by the scorer (ZRGB), 1: converted zslice plane
    mutable int64_t last_converted_slots = 0;
    mutable cds::PairSlotPool pair_pool;  // cds_shape_score_pair: resident slots (own streams), created on first use
};

extern "C" {

int cds_make_zgap(cds_context *ctx, const uint8_t *target_rgb, int32_t W, int32_t H, int32_t threshold, double radius,
                  const cds_rect *excluded, int32_t n_excluded, uint8_t *zgap_rgb) {
    if (!ctx || !target_rgb || !zgap_rgb) return fail(CDS_ERR_INVALID_ARGUMENT, "NULL argument");
    if (!image_size_ok(W, H)) return fail(CDS_ERR_UNSUPPORTED, kImageSizeMessage);
    if (n_excluded < 0 || n_excluded > CDS_MAX_RECTS || (n_excluded > 0 && !excluded)) return fail(CDS_ERR_INVALID_ARGUMENT, "bad excluded-region list");
    if (!(radius >= 0 && radius <= 200)) return fail(CDS_ERR_INVALID_ARGUMENT, "radius outside 0..200");
    // the bbox shortcut in max_filter_host looks at the unfiltered source: harmless (a superset)
    return max_filter_host(ctx, target_rgb, W, H, radius, threshold, excluded, n_excluded, zgap_rgb);
}

int cds_library_ensure_variants(cds_library *lib);

int cds_library_make_zgap_variants(cds_library *lib, int64_t first, int64_t count, int32_t threshold, double radius,
                                   const cds_rect *excluded, int32_t n_excluded) {
    if (!lib) return fail(CDS_ERR_INVALID_ARGUMENT, "library is NULL");
    if (first < 0 || count < 0 || first + count > lib->capacity) return fail(CDS_ERR_INVALID_ARGUMENT, "slot range outside the library");
    if (n_excluded < 0 || n_excluded > CDS_MAX_RECTS || (n_excluded > 0 && !excluded)) return fail(CDS_ERR_INVALID_ARGUMENT, "bad excluded-region list");
    if (!(radius >= 0 && radius <= 200)) return fail(CDS_ERR_INVALID_ARGUMENT, "radius outside 0..200");
    if (count == 0) return CDS_OK;
    int rc = cds_library_ensure_variants(lib);
    if (rc) return rc;
    cds_context *ctx = lib->ctx;
    CDS_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    std::vector<int> hw;
    const int k = line_radii(radius, hw);
    uint8_t *d_tmp = nullptr;
    int *d_hw = nullptr;
    CDS_HIP(dev_malloc(&d_tmp, lib->img_stride));
    hipError_t e = dev_malloc(&d_hw, hw.size() * sizeof(int));
    if (e == hipSuccess) e = hipMemcpyAsync(d_hw, hw.data(), hw.size() * sizeof(int), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemsetAsync(d_tmp, 0, lib->img_stride, st);
    MaxFilterParams p{};
    p.pitch = lib->pitch; p.W = lib->W; p.H = lib->H; p.k = k; p.thr = threshold; p.nrects = n_excluded; p.hw = d_hw;
    for (int i = 0; i < n_excluded; i++) p.rects[i] = excluded[i];
    p.bx0 = 0; p.by0 = 0; p.bx1 = lib->W; p.by1 = lib->H;
    for (int64_t i = 0; i < count && e == hipSuccess; i++) {
        p.src = lib->d_rgb + (size_t)(first + i) * lib->img_stride;
        p.dst = d_tmp;
        hipLaunchKernelGGL(prefilter_kernel, dim3((lib->W + 255) / 256, lib->H), dim3(256), 0, st, p);
        p.src = d_tmp;
        p.dst = lib->d_zgap + (size_t)(first + i) * lib->img_stride;
        hipLaunchKernelGGL(max_filter_kernel, dim3((lib->W + 255) / 256, lib->H), dim3(256), 0, st, p);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    hipFree(d_tmp);
    if (d_hw) hipFree(d_hw);
    if (e != hipSuccess) return hip_fail(e, "make_zgap_variants");
    zgap_written(lib, first, count);
    rc = CDS_OK;
    if (rc) return rc;
    CDS_HIP(hipStreamSynchronize(st));
    return CDS_OK;
}

int cds_shapeset_create(cds_context *ctx, int32_t n_masks, const uint8_t *masks_rgb, int32_t W, int32_t H,
                        int32_t query_threshold, int32_t mirror, const uint8_t *roi, const cds_rect *excluded,
                        int32_t n_excluded, cds_shapeset **out) {
    return cds_shapeset_create_bordered(ctx, n_masks, masks_rgb, W, H, query_threshold, 0, mirror, roi, excluded, n_excluded, out);
}

int cds_shapeset_create_bordered(cds_context *ctx, int32_t n_masks, const uint8_t *masks_rgb, int32_t W, int32_t H,
                                 int32_t query_threshold, int32_t query_border_size, int32_t mirror, const uint8_t *roi,
                                 const cds_rect *excluded, int32_t n_excluded, cds_shapeset **out) {
    if (!ctx || !out) return fail(CDS_ERR_INVALID_ARGUMENT, "NULL argument");
    *out = nullptr;
    if (query_border_size < 0) return fail(CDS_ERR_INVALID_ARGUMENT, "queryBorderSize is negative");
    if (n_masks <= 0 || !masks_rgb) return fail(CDS_ERR_INVALID_ARGUMENT, "no masks");
    if (!image_size_ok(W, H)) return fail(CDS_ERR_UNSUPPORTED, kImageSizeMessage);
    if (n_excluded < 0 || n_excluded > CDS_MAX_RECTS || (n_excluded > 0 && !excluded)) return fail(CDS_ERR_INVALID_ARGUMENT, "bad excluded-region list");
    int rc = ensure_slice_lut(ctx);
    if (rc) return rc;
    const size_t N = (size_t)W * H;
    const int wp = round_up(W, 4);
    cds_shapeset *ss = new cds_shapeset();
    ss->ctx = ctx; ss->W = W; ss->H = H; ss->wp = wp; ss->n_masks = n_masks; ss->thr = query_threshold; ss->mirror = mirror != 0;
    ss->nrects = n_excluded;
    for (int i = 0; i < n_excluded; i++) ss->rects[i] = excluded[i];
    ss->qm_sum.assign((size_t)n_masks, 0);
    ss->qh_sum.assign((size_t)n_masks, 0);
    CDS_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    std::vector<int> hw60, hw20;
    const int k60 = line_radii(60, hw60), k20 = line_radii(20, hw20);
    if (k60 >= kHdMax) { delete ss; return fail(CDS_ERR_UNSUPPORTED, "max-filter radius beyond the distance maps"); }
    const int batch = std::min(n_masks, 32);   // masks prepared per round of launches (5.5 MB of scratch each)
    // Rounds alternate between TWO streams with their own scratch: the upload of round i + 1 (64 MB of pageable host memory: the
    // host is busy staging it) runs while the three kernels of round i execute; every round writes its boxes / row flags / sums into
    // arrays for ALL masks that are read back once, after the last round (round 3 synchronised after every round: 1,024 masks
    // 0.12 s, most of it the 2.1 GB upload with the kernels' time on top).
    const int nset = n_masks > batch ? 2 : 1;
    uint8_t *d_masks[2] = {nullptr, nullptr}, *d_roi = nullptr, *d_flags[2] = {nullptr, nullptr}, *d_hd[2] = {nullptr, nullptr};
    int *d_hw = nullptr, *d_acc = nullptr;
    uint32_t *d_rfl = nullptr;
    unsigned long long *d_sums = nullptr;
    hipStream_t st2 = nullptr;
    hipEvent_t ev_a = nullptr, ev_b = nullptr;
    auto cleanup = [&]() {
        for (void *q : {(void *)d_masks[0], (void *)d_masks[1], (void *)d_roi, (void *)d_flags[0], (void *)d_flags[1], (void *)d_hd[0], (void *)d_hd[1],
                        (void *)d_hw, (void *)d_acc, (void *)d_rfl, (void *)d_sums})
            if (q) hipFree(q);
        if (st2) { hipStreamSynchronize(st2); hipStreamDestroy(st2); }
        if (ev_a) hipEventDestroy(ev_a);
        if (ev_b) hipEventDestroy(ev_b);
    };
#define SP_HIP(call) do { hipError_t _e = (call); if (_e != hipSuccess) { hipStreamSynchronize(st); cleanup(); cds_shapeset_destroy(ss); return hip_fail(_e, #call); } } while (0)
    SP_HIP(dev_malloc(&ss->d_planes, (size_t)n_masks * 2 * H * wp * sizeof(uint16_t)));
    SP_HIP(dev_malloc(&ss->d_boxes, (size_t)n_masks * 8 * sizeof(int)));
    SP_HIP(dev_malloc(&ss->d_rowflags, (size_t)n_masks * H));
    SP_HIP(hipMemsetAsync(ss->d_planes, 0, (size_t)n_masks * 2 * H * wp * sizeof(uint16_t), st));
    for (int k = 0; k < nset; k++) {
        SP_HIP(dev_malloc(&d_masks[k], (size_t)batch * 3 * N));
        SP_HIP(dev_malloc(&d_flags[k], (size_t)batch * N));
        SP_HIP(dev_malloc(&d_hd[k], (size_t)batch * 7 * N));
    }
    SP_HIP(dev_malloc(&d_hw, (hw60.size() + hw20.size()) * sizeof(int)));
    SP_HIP(dev_malloc(&d_acc, (size_t)n_masks * 16 * sizeof(int)));
    SP_HIP(dev_malloc(&d_rfl, (size_t)n_masks * H * sizeof(uint32_t)));
    SP_HIP(dev_malloc(&d_sums, (size_t)n_masks * 2 * sizeof(unsigned long long)));
    SP_HIP(hipMemcpyAsync(d_hw, hw60.data(), hw60.size() * sizeof(int), hipMemcpyHostToDevice, st));
    SP_HIP(hipMemcpyAsync(d_hw + hw60.size(), hw20.data(), hw20.size() * sizeof(int), hipMemcpyHostToDevice, st));
    if (roi) {
        SP_HIP(dev_malloc(&d_roi, N));
        SP_HIP(hipMemcpyAsync(d_roi, roi, N, hipMemcpyHostToDevice, st));
    }
    std::vector<int> acc((size_t)n_masks * 16), boxes((size_t)n_masks * 8, 0);
    for (int i = 0; i < n_masks; i++) {
        int *q = acc.data() + (size_t)16 * i;
        for (int j = 0; j < 16; j++) q[j] = 0;
        q[0] = H; q[2] = W; q[4] = W; q[8] = W; q[9] = H;   // the minima start at the far end
    }
    SP_HIP(hipMemcpyAsync(d_acc, acc.data(), acc.size() * sizeof(int), hipMemcpyHostToDevice, st));
    SP_HIP(hipMemsetAsync(d_rfl, 0, (size_t)n_masks * H * sizeof(uint32_t), st));
    SP_HIP(hipMemsetAsync(d_sums, 0, (size_t)n_masks * 2 * sizeof(unsigned long long), st));
    if (nset > 1) {   // the second stream starts after the set-up above
        SP_HIP(hipStreamCreateWithFlags(&st2, hipStreamNonBlocking));
        SP_HIP(hipEventCreateWithFlags(&ev_a, hipEventDisableTiming));
        SP_HIP(hipEventCreateWithFlags(&ev_b, hipEventDisableTiming));
        SP_HIP(hipEventRecord(ev_a, st));
        SP_HIP(hipStreamWaitEvent(st2, ev_a, 0));
    }
    std::vector<uint32_t> rfl((size_t)n_masks * H);
    std::vector<unsigned long long> sums((size_t)n_masks * 2);
    std::vector<uint8_t> rowflags((size_t)n_masks * H, 0);
    for (int m0 = 0, round = 0; m0 < n_masks; m0 += batch, round++) {
        const int nb = std::min(batch, n_masks - m0);
        const int k = nset > 1 ? (round & 1) : 0;
        hipStream_t sr = k ? st2 : st;
        SP_HIP(hipMemcpyAsync(d_masks[k], masks_rgb + (size_t)m0 * 3 * N, (size_t)nb * 3 * N, hipMemcpyHostToDevice, sr));
        PrepParams p{};
        p.masks = d_masks[k]; p.roi = d_roi; p.W = W; p.H = H; p.wp = wp; p.n = nb; p.mirror = mirror != 0; p.nrects = n_excluded;
        p.border = query_border_size;
        for (int i = 0; i < n_excluded; i++) p.rects[i] = excluded[i];
        p.k60 = k60; p.k20 = k20; p.hw60 = d_hw; p.hw20 = d_hw + hw60.size();
        p.flags = d_flags[k]; p.hd = d_hd[k]; p.planes = ss->d_planes + (size_t)m0 * 2 * H * wp;
        p.acc = d_acc + (size_t)m0 * 16; p.rowflags = d_rfl + (size_t)m0 * H; p.sums = d_sums + (size_t)m0 * 2;
        p.slice_lut = ctx->d_slice_lut;
        const dim3 grid((W + 255) / 256, H, nb);
        hipLaunchKernelGGL(prep_flags_kernel, grid, dim3(256), 0, sr, p);
        hipLaunchKernelGGL(prep_hdist_kernel, grid, dim3(256), 0, sr, p);
        hipLaunchKernelGGL(prep_plane_kernel, grid, dim3(256), 0, sr, p);
        SP_HIP(hipGetLastError());
    }
    if (nset > 1) {   // join: the context's stream continues after the second stream's last round
        SP_HIP(hipEventRecord(ev_b, st2));
        SP_HIP(hipStreamWaitEvent(st, ev_b, 0));
    }
    SP_HIP(hipMemcpyAsync(acc.data(), d_acc, acc.size() * sizeof(int), hipMemcpyDeviceToHost, st));
    SP_HIP(hipMemcpyAsync(rfl.data(), d_rfl, rfl.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    SP_HIP(hipMemcpyAsync(sums.data(), d_sums, sums.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    SP_HIP(hipStreamSynchronize(st));
    for (int m = 0; m < n_masks; m++) {
        const int *q = acc.data() + (size_t)16 * m;
        int ya = q[0], yb = q[1], gx0 = q[2], gx1 = q[3], hx0 = q[4], hx1 = q[5];
        if (yb <= ya) ya = yb = 0;
        if (gx1 <= gx0) gx0 = gx1 = 0;
        if (hx1 <= hx0) hx0 = hx1 = 0;
        if (hx1 == 0 && gx1 > 0) { hx0 = gx0; hx1 = gx0; }   // empty QH window inside the gap window
        if (gx1 == 0 && hx1 > 0) { gx0 = hx0; gx1 = hx0; }
        int *bxm = boxes.data() + (size_t)8 * m;
        bxm[0] = ya; bxm[1] = yb; bxm[2] = gx0; bxm[3] = gx1; bxm[4] = hx0; bxm[5] = hx1;
        for (int y = 0; y < H; y++) rowflags[(size_t)m * H + y] = (uint8_t)rfl[(size_t)m * H + y];
        ss->qm_sum[(size_t)m] = (int64_t)sums[(size_t)2 * m];
        ss->qh_sum[(size_t)m] = (int64_t)sums[(size_t)2 * m + 1];
    }
    SP_HIP(hipMemcpyAsync(ss->d_rowflags, rowflags.data(), rowflags.size(), hipMemcpyHostToDevice, st));
    SP_HIP(hipMemcpyAsync(ss->d_boxes, boxes.data(), boxes.size() * sizeof(int), hipMemcpyHostToDevice, st));
    {   // the group lists of gradscore_list_kernel: count, prefix on the host, fill
        uint32_t *d_cnt = nullptr;
        SP_HIP(dev_malloc(&d_cnt, (size_t)n_masks * 2 * sizeof(uint32_t)));
        auto free_cnt = [&]() { hipFree(d_cnt); };
        ListParams lp{};
        lp.planes = ss->d_planes; lp.H = H; lp.wp = wp; lp.n = n_masks; lp.counts = d_cnt;
        const unsigned lblocks = (unsigned)(((int64_t)n_masks * H + 3) / 4);
        hipError_t e = hipMemsetAsync(d_cnt, 0, (size_t)n_masks * 2 * sizeof(uint32_t), st);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(prep_lists_kernel<false>, dim3(lblocks), dim3(256), 0, st, lp);
            e = hipGetLastError();
        }
        std::vector<uint32_t> cnt((size_t)n_masks * 2), off((size_t)(n_masks + 1) * 2, 0);
        if (e == hipSuccess) e = hipMemcpyAsync(cnt.data(), d_cnt, cnt.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e == hipSuccess) {
            for (int m = 0; m < n_masks; m++) {
                off[(size_t)2 * (m + 1)] = off[(size_t)2 * m] + cnt[(size_t)2 * m];
                off[(size_t)2 * (m + 1) + 1] = off[(size_t)2 * m + 1] + cnt[(size_t)2 * m + 1];
            }
            e = dev_malloc(&ss->d_glist, std::max<size_t>(1, off[(size_t)2 * n_masks]) * sizeof(uint32_t));
        }
        if (e == hipSuccess) e = dev_malloc(&ss->d_gl_a, std::max<size_t>(1, off[(size_t)2 * n_masks]) * sizeof(ushort4));
        if (e == hipSuccess) e = dev_malloc(&ss->d_gl_b, std::max<size_t>(1, off[(size_t)2 * n_masks]) * sizeof(ushort4));
        if (e == hipSuccess) e = dev_malloc(&ss->d_hlist, std::max<size_t>(1, off[(size_t)2 * n_masks + 1]) * sizeof(uint32_t));
        if (e == hipSuccess) e = dev_malloc(&ss->d_list_off, off.size() * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMemcpyAsync(ss->d_list_off, off.data(), off.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipMemsetAsync(d_cnt, 0, (size_t)n_masks * 2 * sizeof(uint32_t), st);
        if (e == hipSuccess) {
            lp.list_off = ss->d_list_off; lp.glist = ss->d_glist; lp.hlist = ss->d_hlist; lp.gl_a = ss->d_gl_a; lp.gl_b = ss->d_gl_b;
            hipLaunchKernelGGL(prep_lists_kernel<true>, dim3(lblocks), dim3(256), 0, st, lp);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        free_cnt();
        if (e != hipSuccess) { cleanup(); cds_shapeset_destroy(ss); return hip_fail(e, "shape group lists"); }
    }
    SP_HIP(hipStreamSynchronize(st));
#undef SP_HIP
    cleanup();
    *out = ss;
    return CDS_OK;
}

int32_t cds_shapeset_size(const cds_shapeset *ss) { return ss ? ss->n_masks : 0; }

void cds_shapeset_destroy(cds_shapeset *ss) {
    if (!ss) return;
    hipSetDevice(ss->ctx->device);
    if (ss->d_planes) hipFree(ss->d_planes);
    if (ss->d_boxes) hipFree(ss->d_boxes);
    if (ss->d_rowflags) hipFree(ss->d_rowflags);
    if (ss->d_glist) hipFree(ss->d_glist);
    if (ss->d_hlist) hipFree(ss->d_hlist);
    if (ss->d_list_off) hipFree(ss->d_list_off);
    if (ss->d_gl_a) hipFree(ss->d_gl_a);
    if (ss->d_gl_b) hipFree(ss->d_gl_b);
    ss->pair_pool.slots.clear();
    delete ss;
}

int cds_shapeset_mask_stats(const cds_shapeset *ss, int64_t *mask_size, int64_t *high_expression_size) {
    if (!ss) return fail(CDS_ERR_INVALID_ARGUMENT, "NULL argument");
    for (int m = 0; m < ss->n_masks; m++) {
        if (mask_size) mask_size[m] = ss->qm_sum[(size_t)m];
        if (high_expression_size) high_expression_size[m] = ss->qh_sum[(size_t)m];
    }
    return CDS_OK;
}

int64_t cds_shape_score_value(int64_t gap, int64_t high) {  // GradientAreaGapUtils.java:199-207
    if (gap >= 0 && high >= 0) return gap + high / 3;
    return -1;
}

double cds_normalized_score(int32_t pix, int64_t shape, int64_t max_pix, int64_t max_shape) {  // GradientAreaGapUtils.java:219-235
    if (pix == 0 || max_pix == 0 || shape < 0 || max_shape <= 0) return pix;
    double np = (double)pix / max_pix;
    double ns = (double)shape / max_shape;
    double bounded = std::fmin(std::fmax(ns * 2.5, 0.002), 1.);
    return np / bounded * 100;
}

// A scoring call in two halves, so that one host thread can keep several GPUs busy (cds_gradscore_sharded, cds_multi.hip):
// gradscore_begin uploads the pair list, launches the kernel and requests the read-back -- all asynchronous on the stream --,
// gradscore_end waits for that stream and turns the four sums per pair into the reference's (gap, high, mirrored).
}  // extern "C"
namespace cds {
struct GradscoreJob {
    const cds_shapeset *ss = nullptr;
    hipStream_t st = nullptr;
    int64_t n_pairs = 0;
    std::vector<int64_t> slots;      // < 0: the pair has no variants (-1, -1)
    int32_t *d_pm = nullptr;
    int64_t *d_ps = nullptr, *d_out = nullptr;
    int64_t *h_res = nullptr;        // pinned: [n_pairs][4]
    hipEvent_t ev0 = nullptr, ev1 = nullptr, evc = nullptr, evd = nullptr;   // kernel start / end; start / end of the zslice conversion (if any)
    int mode = 0;                    // 0: the scorer reads the zgap bytes, 1: the converted zslice plane
    int64_t converted = 0;
    void release() {
        if (d_pm) hipFree(d_pm);
        if (d_ps) hipFree(d_ps);
        if (d_out) hipFree(d_out);
        if (h_res) hipHostFree(h_res);
        if (ev0) hipEventDestroy(ev0);
        if (ev1) hipEventDestroy(ev1);
        if (evc) hipEventDestroy(evc);
        if (evd) hipEventDestroy(evd);
        d_pm = nullptr; d_ps = d_out = h_res = nullptr; ev0 = ev1 = evc = evd = nullptr;
    }
};

cds_context *shapeset_context(const cds_shapeset *ss) { return ss ? ss->ctx : nullptr; }
int shapeset_n_masks(const cds_shapeset *ss) { return ss ? ss->n_masks : 0; }

// `on_stream` (the pair-slot path): launch on that stream; the library then belongs to a slot context of the same device
int gradscore_begin(const cds_shapeset *ss, const cds_library *lib, int64_t n_pairs, const int32_t *pair_mask, const int64_t *pair_slot,
                    hipStream_t on_stream, GradscoreJob **out) {
    *out = nullptr;
    if (!ss || !lib) return fail(CDS_ERR_INVALID_ARGUMENT, "NULL argument");
    if (n_pairs < 0) return fail(CDS_ERR_INVALID_ARGUMENT, "n_pairs < 0");
    if (n_pairs == 0) return CDS_OK;
    if (!pair_mask || !pair_slot) return fail(CDS_ERR_INVALID_ARGUMENT, "NULL argument");
    if (!on_stream && lib->ctx != ss->ctx) return fail(CDS_ERR_INVALID_ARGUMENT, "shape set and library belong to different contexts");
    if (lib->W != ss->W || lib->H != ss->H) return fail(CDS_ERR_INVALID_ARGUMENT, "target image size must match the query image size");
    std::unique_ptr<GradscoreJob> job(new GradscoreJob());
    job->ss = ss;
    job->n_pairs = n_pairs;
    job->slots.resize((size_t)n_pairs);
    for (int64_t i = 0; i < n_pairs; i++) {
        if (pair_mask[i] < 0 || pair_mask[i] >= ss->n_masks) return fail(CDS_ERR_INVALID_ARGUMENT, "pair mask index out of range");
        if (pair_slot[i] < 0 || pair_slot[i] >= lib->capacity) return fail(CDS_ERR_INVALID_ARGUMENT, "pair slot out of range");
        const bool ok = lib->d_grad && lib->d_zgap && lib->has_grad[(size_t)pair_slot[i]] && lib->has_zgap[(size_t)pair_slot[i]];
        job->slots[(size_t)i] = ok ? pair_slot[i] : -1;  // Shape2DMatch...Algorithm.java:155-158
    }
    cds_context *ctx = ss->ctx;
    CDS_HIP(hipSetDevice(ctx->device));
    hipStream_t st = on_stream ? on_stream : ctx->stream;
    job->st = st;
    GradscoreJob *j = job.get();
    // Where S(z) comes from.  Converting a zgap image into its zslice plane is a pass over all of its ~685 K pixels (3.4 MB of
    // traffic: about what THREE pairs cost the scorer); reading the zgap bytes in the scorer costs a pair a few more bytes and table
    // gathers.  So: the plane for slots that already have it or that serve many pairs of this call, the bytes otherwise -- BASELINE
    // configs[4] (top-500 of 1,024 masks: 2.2 pairs per selected target) and the per-pair entry point read the bytes.
    {
        std::vector<int64_t> distinct;
        int64_t n_ok = 0;
        for (int64_t v : j->slots)
            if (v >= 0) distinct.push_back(v), n_ok++;
        std::sort(distinct.begin(), distinct.end());
        distinct.erase(std::unique(distinct.begin(), distinct.end()), distinct.end());
        int64_t stale = 0;
        {
            std::lock_guard<std::mutex> lk(const_cast<cds_library *>(lib)->mu);
            for (int64_t v : distinct) stale += lib->zslice_ok[(size_t)v] ? 0 : 1;
        }
        int pairs_per_conversion = 8;   // convert when the call brings at least this many pairs per slot that still needs it
        if (const char *e = getenv("CDS_SHAPE_PAIRS_PER_CONVERSION")) pairs_per_conversion = std::max(0, atoi(e));
        bool plane = !distinct.empty() && (stale == 0 || n_ok >= (int64_t)pairs_per_conversion * stale);
        if (const char *e = getenv("CDS_SHAPE_ZSLICE")) plane = atoi(e) != 0;      // (A/B runs: 1 always convert, 0 never)
        if (getenv("CDS_SHAPE_SCAN_KERNEL")) plane = true;                          // (the plane-window scan reads zslice only)
        if (plane && !distinct.empty()) {
            hipError_t e = hipEventCreate(&j->evc);
            if (e == hipSuccess) e = hipEventRecord(j->evc, st);
            if (e != hipSuccess) { j->release(); return hip_fail(e, "hipEventRecord"); }
            const int rc = ensure_zslice(const_cast<cds_library *>(lib), distinct, st);
            if (rc) { hipStreamSynchronize(st); j->release(); return rc; }
            e = hipEventCreate(&j->evd);
            if (e == hipSuccess) e = hipEventRecord(j->evd, st);
            if (e != hipSuccess) { hipStreamSynchronize(st); j->release(); return hip_fail(e, "hipEventRecord"); }
            j->converted = stale;
        }
        j->mode = plane ? 1 : 0;
        if (!plane) {
            const int rc = ensure_slice_lut(ctx);
            if (rc) { j->release(); return rc; }
        }
    }
#define GS_HIP(call) do { hipError_t _e = (call); if (_e != hipSuccess) { hipStreamSynchronize(st); j->release(); return hip_fail(_e, #call); } } while (0)
    GS_HIP(dev_malloc(&j->d_pm, (size_t)n_pairs * 4));
    GS_HIP(dev_malloc(&j->d_ps, (size_t)n_pairs * 8));
    GS_HIP(dev_malloc(&j->d_out, (size_t)n_pairs * 32));
    GS_HIP(hipHostMalloc((void **)&j->h_res, (size_t)n_pairs * 32));
    // (pair_mask is the caller's and may be gone before the copy runs when begin / end are apart: the slots are the job's own copy,
    // the mask indices go through the pinned result buffer's first quarter, which the read-back overwrites later)
    memcpy(j->h_res, pair_mask, (size_t)n_pairs * 4);
    GS_HIP(hipMemcpyAsync(j->d_pm, j->h_res, (size_t)n_pairs * 4, hipMemcpyHostToDevice, st));
    GS_HIP(hipMemcpyAsync(j->d_ps, j->slots.data(), (size_t)n_pairs * 8, hipMemcpyHostToDevice, st));
    GS_HIP(hipMemsetAsync(j->d_out, 0, (size_t)n_pairs * 32, st));
    ShapeParams p{};
    p.rgb = lib->d_rgb; p.zslice = lib->d_zslice; p.grad = lib->d_grad;
    p.zgap = lib->d_zgap; p.slice_lut = ctx->d_slice_lut;
    p.img_stride = lib->img_stride; p.grad_stride = lib->grad_stride;
    p.pitch = lib->pitch; p.gpitch = lib->gpitch; p.W = ss->W; p.H = ss->H; p.wp = ss->wp;
    p.planes = ss->d_planes; p.boxes = ss->d_boxes; p.rowflags = ss->d_rowflags;
    p.thr = ss->thr; p.mirror = ss->mirror; p.nrects = ss->nrects;
    for (int i = 0; i < ss->nrects; i++) p.rects[i] = ss->rects[i];
    p.pair_mask = j->d_pm; p.pair_slot = j->d_ps; p.n_pairs = n_pairs; p.out = j->d_out;
    int per_cu = 32;
    if (const char *e = getenv("CDS_SHAPE_WGS_PER_CU")) per_cu = atoi(e);
    int64_t g64 = (int64_t)ctx->n_cus * per_cu;
    int grid = (int)std::min<int64_t>(n_pairs, g64);
    GS_HIP(hipEventCreate(&j->ev0));
    GS_HIP(hipEventCreate(&j->ev1));
    GS_HIP(hipEventRecord(j->ev0, st));
    p.glist = ss->d_glist; p.hlist = ss->d_hlist; p.list_off = ss->d_list_off; p.gl_a = ss->d_gl_a; p.gl_b = ss->d_gl_b;
    if (getenv("CDS_SHAPE_SCAN_KERNEL")) hipLaunchKernelGGL(gradscore_kernel, dim3(grid), dim3(256), 0, st, p);   // (the plane-window scan, for A/B runs)
    else if (j->mode) hipLaunchKernelGGL(gradscore_list_kernel<false>, dim3(grid), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(gradscore_list_kernel<true>, dim3(grid), dim3(256), 0, st, p);
    GS_HIP(hipGetLastError());
    GS_HIP(hipEventRecord(j->ev1, st));
    GS_HIP(hipMemcpyAsync(j->h_res, j->d_out, (size_t)n_pairs * 32, hipMemcpyDeviceToHost, st));
#undef GS_HIP
    *out = job.release();
    return CDS_OK;
}

// waits for the job's stream; outputs may be NULL (abandon: wait and free only).  The job is gone afterwards.
int gradscore_end(GradscoreJob *j, int64_t *gradient_area_gap, int64_t *high_expression_area, uint8_t *mirrored) {
    if (!j) return CDS_OK;
    const cds_shapeset *ss = j->ss;
    hipSetDevice(ss->ctx->device);
    const hipError_t e = hipStreamSynchronize(j->st);
    int rc = CDS_OK;
    if (e != hipSuccess) rc = hip_fail(e, "hipStreamSynchronize (gradscore)");
    if (rc == CDS_OK && gradient_area_gap) {
        float t = 0, tc = 0;
        hipEventElapsedTime(&t, j->ev0, j->ev1);
        if (j->evc && j->evd) hipEventElapsedTime(&tc, j->evc, j->evd);
        {
            std::lock_guard<std::mutex> lk(ss->mu);
            ss->last_ms = t;
            ss->last_launches = 1;
            ss->last_convert_ms = j->converted ? tc : 0;
            ss->last_mode = j->mode;
            ss->last_converted_slots = j->converted;
        }
        for (int64_t i = 0; i < j->n_pairs; i++) {
            if (j->slots[(size_t)i] < 0) {
                gradient_area_gap[i] = -1; high_expression_area[i] = -1; mirrored[i] = 0;
                continue;
            }
            const int64_t *r = &j->h_res[(size_t)i * 4];
            gradient_area_gap[i] = r[0]; high_expression_area[i] = r[1]; mirrored[i] = 0;
            // keep the mirrored orientation iff its int-cast score is lower (Shape2DMatch...Algorithm.java:181-183)
            if (ss->mirror && (int)cds_shape_score_value(r[2], r[3]) < (int)cds_shape_score_value(r[0], r[1])) {
                gradient_area_gap[i] = r[2]; high_expression_area[i] = r[3]; mirrored[i] = 1;
            }
        }
    }
    j->release();
    delete j;
    return rc;
}
}  // namespace cds
extern "C" {

static int gradscore_impl(const cds_shapeset *ss, const cds_library *lib, int64_t n_pairs, const int32_t *pair_mask, const int64_t *pair_slot,
                          int64_t *gradient_area_gap, int64_t *high_expression_area, uint8_t *mirrored, hipStream_t on_stream) {
    if (n_pairs > 0 && (!gradient_area_gap || !high_expression_area || !mirrored)) return fail(CDS_ERR_INVALID_ARGUMENT, "NULL argument");
    GradscoreJob *job = nullptr;
    const int rc = gradscore_begin(ss, lib, n_pairs, pair_mask, pair_slot, on_stream, &job);
    if (rc || !job) return rc;
    return gradscore_end(job, gradient_area_gap, high_expression_area, mirrored);
}

int cds_gradscore(const cds_shapeset *ss, const cds_library *lib, int64_t n_pairs, const int32_t *pair_mask,
                  const int64_t *pair_slot, int64_t *gradient_area_gap, int64_t *high_expression_area, uint8_t *mirrored) {
    return gradscore_impl(ss, lib, n_pairs, pair_mask, pair_slot, gradient_area_gap, high_expression_area, mirrored, nullptr);
}

int cds_shape_score_pair(const cds_shapeset *ss, int32_t mask_index, const uint8_t *target_rgb, const uint16_t *grad,
                         const uint8_t *zgap_rgb, int32_t W, int32_t H, int64_t *gradient_area_gap,
                         int64_t *high_expression_area, uint8_t *mirrored) {
    if (!ss || !target_rgb || !gradient_area_gap || !high_expression_area || !mirrored) return fail(CDS_ERR_INVALID_ARGUMENT, "NULL argument");
    if (mask_index < 0 || mask_index >= ss->n_masks) return fail(CDS_ERR_INVALID_ARGUMENT, "mask_index out of range");
    if (!grad || !zgap_rgb) {  // Shape2DMatch...Algorithm.java:155-158
        *gradient_area_gap = -1; *high_expression_area = -1; *mirrored = 0;
        return CDS_OK;
    }
    if (W != ss->W || H != ss->H) return fail(CDS_ERR_INVALID_ARGUMENT, "target image size must match the query image size");
    // resident slots (with their variant planes, each on its own stream) are kept with the mask set for the per-pair call of the
    // reference's interface: calls from different host threads overlap
    PairSlot *slot = nullptr;
    int rc = ss->pair_pool.acquire(ss->ctx, W, H, 0, &slot);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(slot->mu, std::adopt_lock);
    rc = cds_library_upload(slot->lib, 0, 1, target_rgb);
    if (rc == CDS_OK) rc = cds_library_upload_variants(slot->lib, 0, 1, grad, zgap_rgb);
    int64_t s0 = 0;
    if (rc == CDS_OK) rc = gradscore_impl(ss, slot->lib, 1, &mask_index, &s0, gradient_area_gap, high_expression_area, mirrored, slot->sub->stream);
    return rc;
}

int cds_shapeset_last_call_info(const cds_shapeset *ss, int32_t *used_zslice_plane, int64_t *converted_slots, double *conversion_ms) {
    if (!ss) return fail(CDS_ERR_INVALID_ARGUMENT, "NULL argument");
    std::lock_guard<std::mutex> lk(ss->mu);
    if (used_zslice_plane) *used_zslice_plane = ss->last_mode;
    if (converted_slots) *converted_slots = ss->last_converted_slots;
    if (conversion_ms) *conversion_ms = ss->last_convert_ms;
    return CDS_OK;
}

int cds_shapeset_last_kernel_ms(const cds_shapeset *ss, double *avg_ms, int32_t *launches) {
    if (!ss) return fail(CDS_ERR_INVALID_ARGUMENT, "NULL argument");
    std::lock_guard<std::mutex> lk(ss->mu);
    if (avg_ms) *avg_ms = ss->last_ms;
    if (launches) *launches = ss->last_launches;
    return CDS_OK;
}

}  // extern "C"
