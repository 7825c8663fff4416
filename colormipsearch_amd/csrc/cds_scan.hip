// cds_scan.hip -- kernel 1: the mask-vs-library pixel-match scan for gfx950 (MI355X).
//
// What it computes (reference: API/cds/PixelMatchColorDepthSearchAlgorithm.java:166-263 with
// calculatePixelGap API/cds/AbstractColorDepthSearchAlgorithm.java:157-390): for every (mask, target)
// pair and every transform (9 xy-shifts x {plain, mirrored}) the number of mask pixels whose
// transformed target pixel passes the data threshold and lies within zTolerance in hue-ratio space;
// then max over shifts, mirrored only if strictly greater.
//
// How (MI355X-first, not the reference's gather loops):
//   * one workgroup streams ONE target: the rows the mask can touch, and of those only the column
//     window the mask (and its mirror image) can touch, as 12-byte (4 pixel) coalesced loads;
//   * every loaded pixel is converted ONCE to a 32-bit "hue code" = class<<29 | bits(med/max)>>1 (0 for
//     below-threshold / class-less pixels) and written to an LDS ring of rows;  med/max is computed as
//     fl32(med) * v_rcp_f32(max); the host builds its acceptance intervals from the very same
//     v_rcp_f32 table (read back from this device at context creation), so the device predicate is an
//     integer range test that is bit-exact with the reference's double arithmetic by construction
//     (cds_host.hip: build_ratio_table / intervals_for);
//   * each mask pixel (two code intervals: own hue class + at most one adjacent class) probes the 18
//     transformed positions in LDS; per-lane counters (one mask) or wave ballots (mask batches) count;
//   * no MFMA: byte compare / integer count, HBM- and VALU-bound.
#include "cds_internal.hpp"

namespace cds {

struct u3 {
    uint32_t a, b, c;
};
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
typedef u32x3 u32x3_a4 __attribute__((aligned(4)));  // 12-byte groups are only 4-byte aligned

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

__device__ __forceinline__ uint4 convert4(u3 v, float thr) {
    uint4 o;
    o.x = hue_code((float)(v.a & 255), (float)((v.a >> 8) & 255), (float)((v.a >> 16) & 255), thr);
    o.y = hue_code((float)(v.a >> 24), (float)(v.b & 255), (float)((v.b >> 8) & 255), thr);
    o.z = hue_code((float)((v.b >> 16) & 255), (float)(v.b >> 24), (float)(v.c & 255), thr);
    o.w = hue_code((float)((v.c >> 8) & 255), (float)((v.c >> 16) & 255), (float)(v.c >> 24), thr);
    return o;
}

// Stage target rows [ra, rb) (relative to row m0-halo) of the column window into the LDS ring.
template <int RR, int NT>
__device__ __forceinline__ void stage_rows(const ScanParams &p, const uint8_t *img, uint32_t *ring, int ra, int rb) {
    const int items = (rb - ra) * p.gpr;
    const int rowbase = p.m0 - p.halo;
    constexpr int U = 4;
    for (int it0 = threadIdx.x; it0 < items; it0 += NT * U) {
        u3 v[U];
        int dst[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            int it = it0 + u * NT;
            v[u].a = v[u].b = v[u].c = 0;
            dst[u] = -1;
            if (it < items) {
                int rr = (int)__umulhi((uint32_t)it, p.gpr_magic);
                int gi = it - rr * p.gpr;
                int rel = ra + rr;
                int row = rel + rowbase;
                int x0 = p.cx0 + 4 * gi;
                dst[u] = (rel & (RR - 1)) * p.wc + 4 * gi;
                if (row >= 0 && row < p.H && x0 >= 0 && x0 + 4 <= p.wpad) {
                    const u32x3_a4 *src = reinterpret_cast<const u32x3_a4 *>(img + (size_t)row * p.pitch + (size_t)x0 * 3);
                    const u32x3 w = __builtin_nontemporal_load(src);
                    v[u].a = w.x; v[u].b = w.y; v[u].c = w.z;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (dst[u] >= 0) {
                uint4 codes = make_uint4(0, 0, 0, 0);
                if (!(p.debug & 2) && __any((v[u].a | v[u].b | v[u].c) != 0)) codes = convert4(v[u], p.data_thr);
                *reinterpret_cast<uint4 *>(ring + dst[u]) = codes;
            }
        }
    }
}

// Software-pipelined staging: issue_loads() puts the raw 12-byte groups of rows [ra, rb) into registers
// (the loads stay in flight while the previous band is probed), commit() converts them and writes the ring.
template <int RR, int NT, int UP>
__device__ __forceinline__ void issue_loads(const ScanParams &p, const uint8_t *img, int ra, int rb, u3 (&v)[UP], int (&dst)[UP]) {
    const int items = (rb - ra) * p.gpr;
    const int rowbase = p.m0 - p.halo;
#pragma unroll
    for (int u = 0; u < UP; u++) {
        const int it = threadIdx.x + u * NT;
        v[u].a = v[u].b = v[u].c = 0;
        dst[u] = -1;
        if (it < items) {
            const int rr = (int)__umulhi((uint32_t)it, p.gpr_magic);
            const int gi = it - rr * p.gpr;
            const int rel = ra + rr;
            const int row = rel + rowbase;
            const int x0 = p.cx0 + 4 * gi;
            dst[u] = (rel & (RR - 1)) * p.wc + 4 * gi;
            if (row >= 0 && row < p.H && x0 >= 0 && x0 + 4 <= p.wpad) {
                const u32x3_a4 *src = reinterpret_cast<const u32x3_a4 *>(img + (size_t)row * p.pitch + (size_t)x0 * 3);
                const u32x3 w = __builtin_nontemporal_load(src);
                v[u].a = w.x; v[u].b = w.y; v[u].c = w.z;
            }
        }
    }
}

template <int UP>
__device__ __forceinline__ void commit(const ScanParams &p, uint32_t *ring, const u3 (&v)[UP], const int (&dst)[UP]) {
#pragma unroll
    for (int u = 0; u < UP; u++) {
        if (dst[u] >= 0) {
            uint4 codes = make_uint4(0, 0, 0, 0);
            if (!(p.debug & 2) && __any((v[u].a | v[u].b | v[u].c) != 0)) codes = convert4(v[u], p.data_thr);
            *reinterpret_cast<uint4 *>(ring + dst[u]) = codes;
        }
        __builtin_amdgcn_sched_barrier(0);  // one group at a time: keeps the live ranges (VGPRs) short
    }
}

__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// XY2: xyShift == 2 (9 shifts) else 0 (1 shift).  MIRROR: also the mirrored orientation.
// MULTI: several masks share the staged rows (counts by wave ballot into LDS) else one mask with
// per-lane counters kept in registers for the whole target.
template <bool XY2, bool MIRROR, bool MULTI, int BH, int RR, int NT, int UP>
__global__ __launch_bounds__(NT) void scan_kernel(ScanParams p) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    constexpr int NS = XY2 ? 9 : 1;
    constexpr int NSX = XY2 ? 3 : 1;
    constexpr int NO = MIRROR ? 2 : 1;
    constexpr int NC = NS * NO;
    uint32_t *ring = lds;
    int *cnt = reinterpret_cast<int *>(ring + RR * p.wc);  // MULTI: [n_masks][NC]; single: [NC]
    const int lane = threadIdx.x & 63;
    const int halo = XY2 ? 2 : 0;
    const int m1rel = p.nrows;  // mask rows are [0, nrows) relative to m0

    for (int64_t t = blockIdx.x; t < p.count; t += gridDim.x) {
        const uint8_t *img = p.lib + (size_t)t * p.img_stride;
        const int ncnt = MULTI ? p.n_masks * NC : NC;
        for (int i = threadIdx.x; i < ncnt; i += NT) cnt[i] = 0;
        int acc[NC];
#pragma unroll
        for (int k = 0; k < NC; k++) acc[k] = 0;

        // Band k = mask rows [k*BH, (k+1)*BH) needs ring rows up to (k+1)*BH + 2*halo (relative to m0-halo).
        // The first band is staged directly; every later band is loaded into registers one band ahead
        // (pipelined when it fits UP*NT groups) and committed to the ring after the previous band's probes.
        constexpr int UPA = UP > 0 ? UP : 1;
        const bool pipelined = UP > 0 && BH * p.gpr <= UP * NT;
        u3 pv[UPA];
        int pdst[UPA];
        int staged = ((BH < m1rel) ? BH : m1rel) + 2 * halo;
        stage_rows<RR, NT>(p, img, ring, 0, staged);
        int pre_end = staged;  // rows [staged, pre_end) are in flight in pv
        if (pipelined && BH < m1rel) {
            pre_end = ((2 * BH < m1rel) ? 2 * BH : m1rel) + 2 * halo;
            issue_loads<RR, NT, UPA>(p, img, staged, pre_end, pv, pdst);
        }
        __syncthreads();
        for (int y0 = 0; y0 < m1rel; y0 += BH) {
            const int y1 = (y0 + BH < m1rel) ? y0 + BH : m1rel;
            const int nm = (p.debug & 1) ? 0 : (MULTI ? p.n_masks : 1);
            for (int g = 0; g < nm; g++) {
                const uint32_t *rp = p.rowptr + (size_t)g * (p.nrows + 1);
                const uint32_t s = rp[y0], e = rp[y1];
                int sacc[NC];
                if (MULTI) {
#pragma unroll
                    for (int k = 0; k < NC; k++) sacc[k] = 0;
                }
                for (uint32_t i = s + threadIdx.x; i < e; i += NT) {
                    const uint32_t pos = p.px_pos[i];
                    const uint4 iv = p.px_iv[i];
                    const int cpl = pos & 2047, cmi = (pos >> 11) & 2047, yrel = pos >> 22;
#pragma unroll
                    for (int iy = 0; iy < NSX; iy++) {
                        const int rel = yrel + (XY2 ? 2 * iy : 0);  // = yrel + halo + dy
                        const uint32_t *rowp = ring + (rel & (RR - 1)) * p.wc;
#pragma unroll
                        for (int ix = 0; ix < NSX; ix++) {
                            const int dx = XY2 ? 2 * (ix - 1) : 0;
                            const int k = ix * NSX + iy;
                            {
                                const uint32_t c = rowp[cpl + dx];
                                const bool hit = ((c - iv.x) <= iv.y) | ((c - iv.z) <= iv.w);
                                if (MULTI) sacc[k] += __popcll(__ballot(hit));
                                else acc[k] += hit ? 1 : 0;
                            }
                            if (MIRROR) {
                                const uint32_t c = rowp[cmi - dx];
                                const bool hit = ((c - iv.x) <= iv.y) | ((c - iv.z) <= iv.w);
                                if (MULTI) sacc[NS + k] += __popcll(__ballot(hit));
                                else acc[NS + k] += hit ? 1 : 0;
                            }
                        }
                    }
                }
                if (MULTI) {
                    if (lane == 0) {
#pragma unroll
                        for (int k = 0; k < NC; k++)
                            if (sacc[k]) atomicAdd(&cnt[g * NC + k], sacc[k]);
                    }
                }
            }
            if (y1 < m1rel) {  // make the next band's rows available
                if (pipelined) {
                    commit<UPA>(p, ring, pv, pdst);
                    staged = pre_end;
                    const int y2 = (y1 + BH < m1rel) ? y1 + BH : m1rel;
                    if (y2 < m1rel) {
                        pre_end = ((y2 + BH < m1rel) ? y2 + BH : m1rel) + 2 * halo;
                        issue_loads<RR, NT, UPA>(p, img, staged, pre_end, pv, pdst);
                    }
                } else {
                    const int need = ((y1 + BH < m1rel) ? y1 + BH : m1rel) + 2 * halo;
                    stage_rows<RR, NT>(p, img, ring, staged, need);
                    staged = need;
                }
                __syncthreads();
            }
        }
        if (!MULTI) {
#pragma unroll
            for (int k = 0; k < NC; k++) {
                int v = wave_sum(acc[k]);
                if (lane == 0 && v) atomicAdd(&cnt[k], v);
            }
        }
        __syncthreads();
        const int nm = MULTI ? p.n_masks : 1;
        for (int g = threadIdx.x; g < nm; g += NT) {
            const int *c = cnt + g * NC;
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
            if (p.counts) {
                int32_t *co = p.counts + ((size_t)g * p.out_stride + p.out_offset + t) * CDS_N_TRANSFORMS;
                for (int k = 0; k < CDS_N_TRANSFORMS; k++) co[k] = 0;
                for (int k = 0; k < NS; k++) {
                    co[k] = c[k];
                    if (MIRROR) co[9 + k] = c[NS + k];
                }
            }
        }
        __syncthreads();  // cnt and ring are reused by the next target
    }
}

template <bool XY2, bool MIRROR, bool MULTI, int BH, int RR, int NT, int UP>
static int launch_cfg(const ScanParams &p, int grid, size_t lds_bytes, hipStream_t stream) {
    auto kern = scan_kernel<XY2, MIRROR, MULTI, BH, RR, NT, UP>;
    CDS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds_bytes));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds_bytes, stream, p);
    CDS_HIP(hipGetLastError());
    return CDS_OK;
}

template <bool XY2, bool MIRROR, bool MULTI>
static int launch_shape(cds_context *ctx, const ScanParams &p, hipStream_t stream) {
    constexpr int NC = (XY2 ? 9 : 1) * (MIRROR ? 2 : 1);
    const size_t cnt_bytes = (size_t)(MULTI ? p.n_masks : 1) * NC * sizeof(int);
    const size_t ring32 = (size_t)32 * p.wc * 4, ring16 = (size_t)16 * p.wc * 4;
    const size_t lds_max = ctx->lds_per_block;
    int cfg = 0;  // 0: RR=16/BH=6, 1: RR=32/BH=12
    (void)ring32;
    if (const char *e = getenv("CDS_SCAN_CFG")) cfg = atoi(e);
    const bool big = (cfg == 1 || cfg == 3);
    if (big && ring32 + cnt_bytes > lds_max) cfg = 0;
    if (!(cfg == 1 || cfg == 3) && ring16 + cnt_bytes > lds_max) return fail(CDS_ERR_UNSUPPORTED, "column window too wide for LDS");
    int per_cu = 8;
    if (const char *e = getenv("CDS_SCAN_WGS_PER_CU")) per_cu = atoi(e);
    int64_t g64 = (int64_t)ctx->n_cus * per_cu;
    int grid = (int)(p.count < g64 ? p.count : g64);
    if (grid < 1) grid = 1;
    switch (cfg) {
        case 1: return launch_cfg<XY2, MIRROR, MULTI, 12, 32, 512, 0>(p, grid, ring32 + cnt_bytes, stream);
        case 2: return launch_cfg<XY2, MIRROR, MULTI, 6, 16, 512, 0>(p, grid, ring16 + cnt_bytes, stream);
        case 3: return launch_cfg<XY2, MIRROR, MULTI, 12, 32, 1024, 0>(p, grid, ring32 + cnt_bytes, stream);
        case 4: return launch_cfg<XY2, MIRROR, MULTI, 6, 16, 512, 2>(p, grid, ring16 + cnt_bytes, stream);
        case 5: return launch_cfg<XY2, MIRROR, MULTI, 6, 16, 256, 4>(p, grid, ring16 + cnt_bytes, stream);
        default: return launch_cfg<XY2, MIRROR, MULTI, 6, 16, 256, 0>(p, grid, ring16 + cnt_bytes, stream);
    }
}

int launch_scan(cds_context *ctx, const ScanParams &p, bool xy2, bool mirror, hipStream_t stream) {
    const bool multi = p.n_masks > 1;
    if (xy2) {
        if (mirror) return multi ? launch_shape<true, true, true>(ctx, p, stream) : launch_shape<true, true, false>(ctx, p, stream);
        return multi ? launch_shape<true, false, true>(ctx, p, stream) : launch_shape<true, false, false>(ctx, p, stream);
    }
    if (mirror) return multi ? launch_shape<false, true, true>(ctx, p, stream) : launch_shape<false, true, false>(ctx, p, stream);
    return multi ? launch_shape<false, false, true>(ctx, p, stream) : launch_shape<false, false, false>(ctx, p, stream);
}

// v_rcp_f32 of 0..255 as this device computes it -- the host's interval builder uses the same values.
__global__ void rcp_table_kernel(float *out) {
    int i = threadIdx.x;
    out[i] = __builtin_amdgcn_rcpf((float)i);
}

int init_rcp_table(cds_context *ctx) {
    float *d = nullptr;
    CDS_HIP(hipMalloc(&d, 256 * sizeof(float)));
    hipLaunchKernelGGL(rcp_table_kernel, dim3(1), dim3(256), 0, ctx->stream, d);
    CDS_HIP(hipGetLastError());
    CDS_HIP(hipMemcpyAsync(ctx->rcp, d, 256 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    CDS_HIP(hipStreamSynchronize(ctx->stream));
    CDS_HIP(hipFree(d));
    return CDS_OK;
}

}  // namespace cds
