// cds_synth.hip -- on-device generator of the synthetic MIP libraries BASELINE.json asks for
// ("synthetic 1210x566 RGB libraries").  Every pixel is a pure function of
// (seed, image index, x, y): any GPU of a node regenerates its shard independently, and the CPU twin in
// oracle/cds_synth_oracle.c (whose header holds the normative specification) reproduces it bit for bit.
#include <algorithm>

#include "cds_internal.hpp"
#include "cds_lut.hpp"

namespace cds {

static __device__ __constant__ uint8_t kDepthLut[256 * 3] = {CDS_DEPTH_LUT_VALUES};

struct Stroke {
    int orient, hw, z0, dz, bright, slope, amp, freq, phase, a0, a1, c0;
};

__device__ __forceinline__ uint64_t mix64(uint64_t v) {
    v ^= v >> 30; v *= 0xBF58476D1CE4E5B9ULL;
    v ^= v >> 27; v *= 0x94D049BB133111EBULL;
    v ^= v >> 31;
    return v;
}
__device__ __forceinline__ uint64_t draw(uint64_t seed, uint64_t i, uint64_t k, uint64_t j) {
    return mix64(seed ^ ((i + 1) * 0xD1B54A32D192ED03ULL) ^ ((k + 1) * 0x9E3779B97F4A7C15ULL) ^ ((j + 1) * 0xC2B2AE3D27D4EB4FULL));
}
__device__ __forceinline__ int iabs_(int v) { return v < 0 ? -v : v; }
__device__ __forceinline__ int tri_(int u) { return iabs_((u & 255) - 128) - 64; }
__device__ __forceinline__ int f_of(const Stroke &s, int t) {
    return s.c0 + (((t - s.a0) * s.slope) >> 8) + ((tri_((t - s.a0) * s.freq + s.phase) * s.amp) >> 7);
}
__device__ __forceinline__ int z_of(const Stroke &s, int t) {
    int z = s.z0 + (((t - s.a0) * s.dz) >> 8);
    return z < 0 ? 0 : (z > 255 ? 255 : z);
}
__device__ void make_stroke(Stroke &s, uint64_t seed, uint64_t i, int k, int hw_base, int lox, int hix, int loy, int hiy,
                            int len_min, int len_span) {
    uint64_t r0 = draw(seed, i, (uint64_t)k, 0), r1 = draw(seed, i, (uint64_t)k, 1), r2 = draw(seed, i, (uint64_t)k, 2);
    s.orient = (int)(r0 & 1);
    s.hw = hw_base + (int)((r0 >> 1) & 3);
    s.z0 = (int)((r0 >> 8) & 255);
    s.dz = (int)((r0 >> 16) & 255) - 128;
    s.bright = 77 + (int)(((r0 >> 24) & 0xffff) % 179);
    s.slope = (int)((r0 >> 40) & 511) - 256;
    s.amp = (int)((r0 >> 49) & 127);
    s.freq = 1 + (int)((r0 >> 56) & 15);
    s.phase = (int)(r1 >> 56);
    int lo_a = s.orient ? loy : lox, hi_a = s.orient ? hiy : hix;
    int lo_c = s.orient ? lox : loy, hi_c = s.orient ? hix : hiy;
    s.a0 = lo_a + (int)((r1 & 0xffffffffULL) % (uint64_t)(hi_a - lo_a));
    int len = len_min + (int)(((r1 >> 32) & 0xffffff) % (uint64_t)len_span);
    s.a1 = s.a0 + len < hi_a ? s.a0 + len : hi_a;
    s.c0 = lo_c + (int)((r2 & 0xffffffffULL) % (uint64_t)(hi_c - lo_c));
}

__device__ __forceinline__ bool in_label(int W, int x, int y) { return (x < 330 && y < 100) || (x >= W - 270 && y < 90); }

// colour of pixel (x,y): strokes dilated by `dil`; 0 = black.  Returns r | g<<8 | b<<16.
__device__ uint32_t stroke_pixel(const Stroke *s, int K, int x, int y, int dil) {
    for (int k = K - 1; k >= 0; k--) {
        const Stroke &q = s[k];
        int t = q.orient ? y : x, o = q.orient ? x : y;
        if (t < q.a0 - dil || t >= q.a1 + dil) continue;
        int tc = t < q.a0 ? q.a0 : (t > q.a1 - 1 ? q.a1 - 1 : t);
        if (iabs_(o - f_of(q, tc)) > q.hw + dil) continue;
        int z = z_of(q, tc);
        uint32_t r = (kDepthLut[3 * z] * q.bright + 127) / 255, g = (kDepthLut[3 * z + 1] * q.bright + 127) / 255,
                 b = (kDepthLut[3 * z + 2] * q.bright + 127) / 255;
        return r | (g << 8) | (b << 16);
    }
    return 0;
}

struct SynthParams {
    uint8_t *rgb;    // first slot
    uint8_t *zgap;   // first slot or NULL
    uint16_t *grad;  // first slot or NULL
    size_t img_stride, grad_stride;
    int pitch, gpitch, W, H, wpad;
    uint64_t seed;
    int64_t first_index, count;
    const int64_t *index_list;   // NULL: image i of the launch is stream index first_index + i; else index_list[i]
    int kind;
    int blocks_per_image;
};

// one thread = 4 consecutive pixels of one row (12 output bytes)
__global__ __launch_bounds__(256) void synth_kernel(SynthParams p) {
    __shared__ Stroke strokes[16];
    __shared__ int sK;
    const int64_t img = blockIdx.x / p.blocks_per_image;
    const int blk = blockIdx.x % p.blocks_per_image;
    const uint64_t index = (uint64_t)(p.index_list ? p.index_list[img] : p.first_index + img);
    if (p.kind == 0) {
        int K = 3 + (int)(draw(p.seed, index, 255, 0) % 10);
        if (threadIdx.x < K) make_stroke(strokes[threadIdx.x], p.seed, index, threadIdx.x, 2, 0, p.W, 0, p.H, 300, 900);
        if (threadIdx.x == 0) sK = K;
    } else if (p.kind == 2) {
        if (threadIdx.x < 11) make_stroke(strokes[threadIdx.x], p.seed, index, threadIdx.x, 4, 316, 894, 165, 335, 200, 300);
        if (threadIdx.x == 0) sK = 11;
    }
    __syncthreads();
    const int gpr = p.wpad / 4;
    const int item = blk * 256 + threadIdx.x;
    if (item >= gpr * p.H) return;
    const int y = item / gpr, x0 = (item % gpr) * 4;
    uint32_t px[4], zp[4];
    uint16_t gr[4];
    for (int j = 0; j < 4; j++) {
        const int x = x0 + j;
        px[j] = 0; zp[j] = 0; gr[j] = 0;
        if (x >= p.W) continue;
        if (p.kind == 1) {
            uint64_t pi = (uint64_t)y * p.W + x;
            uint64_t h = mix64(p.seed ^ ((index + 1) * 0xD1B54A32D192ED03ULL) ^ ((pi + 1) * 0x9E3779B97F4A7C15ULL));
            px[j] = (uint32_t)(h & 0xffffff);
        } else if (p.kind == 0) {
            const bool lab = in_label(p.W, x, y);
            if (!lab) px[j] = stroke_pixel(strokes, sK, x, y, 0);
            if (p.zgap && !lab) zp[j] = stroke_pixel(strokes, sK, x, y, 10);
            if (p.grad) {
                int best = 300;
                for (int k = 0; k < sK; k++) {
                    const Stroke &q = strokes[k];
                    int t = q.orient ? y : x, o = q.orient ? x : y;
                    int tc = t < q.a0 ? q.a0 : (t > q.a1 - 1 ? q.a1 - 1 : t);
                    int d = iabs_(o - f_of(q, tc)) - q.hw;
                    d = d < 0 ? 0 : d;
                    int e = q.a0 - t > t - (q.a1 - 1) ? q.a0 - t : t - (q.a1 - 1);
                    e = e < 0 ? 0 : e;
                    best = d + e < best ? d + e : best;
                }
                gr[j] = (uint16_t)(2 * best > 600 ? 600 : 2 * best);
            }
        } else {
            if (x >= 316 && x < 894 && y >= 165 && y < 335) px[j] = stroke_pixel(strokes, sK, x, y, 0);
            const bool anchor = ((y == 165 || y == 334) && x >= 600 && x < 616) || ((x == 316 || x == 893) && y >= 250 && y < 258);
            if (anchor) px[j] = (uint32_t)kDepthLut[3 * 128] | ((uint32_t)kDepthLut[3 * 128 + 1] << 8) | ((uint32_t)kDepthLut[3 * 128 + 2] << 16);
        }
    }
    // pack 4 x RGB into 3 dwords: bytes r0 g0 b0 r1 | g1 b1 r2 g2 | b2 r3 g3 b3
    const size_t off = (size_t)img * p.img_stride + (size_t)y * p.pitch + (size_t)x0 * 3;
    uint32_t *o = reinterpret_cast<uint32_t *>(p.rgb + off);
    o[0] = px[0] | (px[1] << 24);
    o[1] = (px[1] >> 8) | (px[2] << 16);
    o[2] = (px[2] >> 16) | (px[3] << 8);
    if (p.zgap) {
        uint32_t *z = reinterpret_cast<uint32_t *>(p.zgap + off);
        z[0] = zp[0] | (zp[1] << 24);
        z[1] = (zp[1] >> 8) | (zp[2] << 16);
        z[2] = (zp[2] >> 16) | (zp[3] << 8);
    }
    if (p.grad) {
        uint16_t *g = reinterpret_cast<uint16_t *>(reinterpret_cast<uint8_t *>(p.grad) + (size_t)img * p.grad_stride + (size_t)y * p.gpitch) + x0;
        for (int j = 0; j < 4; j++)
            if (x0 + j < p.gpitch / 2) g[j] = gr[j];
    }
}

}  // namespace cds

using namespace cds;

extern "C" int cds_library_ensure_variants(cds_library *lib);

static int generate_impl(cds_library *lib, uint64_t seed, int64_t first_index, const int64_t *indices, int64_t first_slot, int64_t count,
                         int32_t kind, int32_t with_variants);

extern "C" int cds_library_generate_synthetic(cds_library *lib, uint64_t seed, int64_t first_index, int64_t first_slot,
                                              int64_t count, int32_t kind, int32_t with_variants) {
    if (first_index < 0) return fail(CDS_ERR_INVALID_ARGUMENT, "first_index is negative");
    return generate_impl(lib, seed, first_index, nullptr, first_slot, count, kind, with_variants);
}

// The same for an arbitrary LIST of stream indices (slot first_slot + i gets image indices[i]): how a benchmark makes the selected
// targets of a gradient-score step resident -- the synthetic stand-in for loading just those targets' files.
extern "C" int cds_library_generate_synthetic_list(cds_library *lib, uint64_t seed, const int64_t *indices, int64_t first_slot,
                                                   int64_t count, int32_t kind, int32_t with_variants) {
    if (count > 0 && !indices) return fail(CDS_ERR_INVALID_ARGUMENT, "indices is NULL");
    for (int64_t i = 0; i < count; i++)
        if (indices[i] < 0) return fail(CDS_ERR_INVALID_ARGUMENT, "a stream index is negative");
    return generate_impl(lib, seed, 0, indices, first_slot, count, kind, with_variants);
}

static int generate_impl(cds_library *lib, uint64_t seed, int64_t first_index, const int64_t *indices, int64_t first_slot, int64_t count,
                         int32_t kind, int32_t with_variants) {
    if (!lib) return fail(CDS_ERR_INVALID_ARGUMENT, "library is NULL");
    if (first_slot < 0 || count < 0 || first_slot + count > lib->capacity)
        return fail(CDS_ERR_INVALID_ARGUMENT, "slot range outside the library");
    if (kind < 0 || kind > 2) return fail(CDS_ERR_INVALID_ARGUMENT, "kind must be 0, 1 or 2");
    if (with_variants && kind != 0) return fail(CDS_ERR_INVALID_ARGUMENT, "variants exist for kind 0 only");
    if (kind == 2 && (lib->W < 894 || lib->H < 335)) return fail(CDS_ERR_INVALID_ARGUMENT, "arbor masks need at least 894 x 335");
    if (count == 0) return CDS_OK;
    CDS_HIP(hipSetDevice(lib->ctx->device));
    if (with_variants) {
        int rc = cds_library_ensure_variants(lib);
        if (rc) return rc;
    }
    int64_t *d_idx = nullptr;
    if (indices) {
        CDS_HIP(dev_malloc(&d_idx, (size_t)count * sizeof(int64_t)));
        const hipError_t e = hipMemcpy(d_idx, indices, (size_t)count * sizeof(int64_t), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            hipFree(d_idx);
            return hip_fail(e, "hipMemcpy(indices)");
        }
    }
    SynthParams p{};
    p.rgb = lib->d_rgb + (size_t)first_slot * lib->img_stride;
    p.zgap = with_variants ? lib->d_zgap + (size_t)first_slot * lib->img_stride : nullptr;
    p.grad = with_variants ? (uint16_t *)((uint8_t *)lib->d_grad + (size_t)first_slot * lib->grad_stride) : nullptr;
    p.img_stride = lib->img_stride; p.grad_stride = lib->grad_stride;
    p.pitch = lib->pitch; p.gpitch = lib->gpitch; p.W = lib->W; p.H = lib->H; p.wpad = lib->pitch / 3;
    p.seed = seed; p.first_index = first_index; p.kind = kind;
    const int items = (p.wpad / 4) * p.H;
    p.blocks_per_image = (items + 255) / 256;
    // HIP limits a launch to fewer than 2^32 work-items per grid dimension (blocks x 256 threads): generate in slices
    const int64_t max_imgs = std::max<int64_t>(1, (int64_t)(0xffffffffu / 256u) / p.blocks_per_image);
    for (int64_t done = 0; done < count; done += max_imgs) {
        SynthParams q = p;
        const int64_t n = std::min<int64_t>(max_imgs, count - done);
        q.rgb = p.rgb + (size_t)done * lib->img_stride;
        if (p.zgap) q.zgap = p.zgap + (size_t)done * lib->img_stride;
        if (p.grad) q.grad = (uint16_t *)((uint8_t *)p.grad + (size_t)done * lib->grad_stride);
        q.first_index = first_index + done;
        q.index_list = d_idx ? d_idx + done : nullptr;
        q.count = n;
        hipLaunchKernelGGL(synth_kernel, dim3((unsigned)(n * p.blocks_per_image)), dim3(256), 0, lib->ctx->stream, q);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) {
            hipStreamSynchronize(lib->ctx->stream);
            if (d_idx) hipFree(d_idx);
            return hip_fail(e, "synth_kernel");
        }
    }
    if (with_variants) {
        for (int64_t i = 0; i < count; i++) lib->has_grad[(size_t)(first_slot + i)] = 1;
        zgap_written(lib, first_slot, count);
    }
    const hipError_t es = hipStreamSynchronize(lib->ctx->stream);
    if (d_idx) hipFree(d_idx);
    if (es != hipSuccess) return hip_fail(es, "hipStreamSynchronize");
    return CDS_OK;
}
