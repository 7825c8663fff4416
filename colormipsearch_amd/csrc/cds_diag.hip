// cds_diag.hip -- measurement helpers of libcds_hip.so (include/cds_hip_debug.h); nothing here is on the search path.
//
// cds_debug_valu_issue_ns: the time one gfx950 SIMD needs per wave64 integer VALU instruction when several waves are
// resident -- the "peak" the instruction-issue roofline of the sparse scan is priced against (bench.py: VALU
// wave-instructions per target from the PMC counters x this time / (SIMDs x kernel time)).  Same kernel as
// tools/valubench/valubench.hip mode 0 (8 independent chains of v_add / v_xor, 8 waves per SIMD).
#include "cds_internal.hpp"

namespace cds {

__global__ __launch_bounds__(1024) void valu_issue_kernel(uint32_t *out, int iters) {
    uint32_t a = threadIdx.x, b = a * 3 + 1, c = a ^ 5, d = a + 7, e = a * 5, f = a + 11, g = a ^ 9, h = a + 3;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            a += b; b ^= c; c += d; d ^= e; e += f; f ^= g; g += h; h ^= a;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + e + f + g + h;
}

// cds_debug_sparse_stream_read: a plain streaming read of a sparse library's record arena (16 bytes per lane, four independent loads in
// flight per lane, grid-stride) -- what THIS box's HBM delivers for the simplest access pattern over the same bytes; the reference point
// bench.py puts beside the scan's own traffic rate and its loads-only twin.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void stream_read_kernel(const u32x4 *p, size_t n16, unsigned long long *out) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        const u32x4 a = __builtin_nontemporal_load(p + i), b = __builtin_nontemporal_load(p + i + stride),
                    c = __builtin_nontemporal_load(p + i + 2 * stride), d = __builtin_nontemporal_load(p + i + 3 * stride);
        acc ^= a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.y ^ c.z ^ c.w ^ d.x ^ d.y ^ d.z ^ d.w;
    }
    for (; i < n16; i += stride) {
        const u32x4 a = __builtin_nontemporal_load(p + i);
        acc ^= a.x ^ a.y ^ a.z ^ a.w;
    }
    if (acc == 0x9e3779b9u) atomicAdd(out, 1ull);   // (keeps the loads alive; practically never taken)
}

}  // namespace cds

using namespace cds;

extern "C" int cds_debug_sparse_stream_read(const cds_sparse_library *sl, double *gb_per_s, double *bytes_read) {
    if (!sl || !gb_per_s) return fail(CDS_ERR_INVALID_ARGUMENT, "NULL argument");
    cds_context *ctx = sl->ctx;
    CDS_HIP(hipSetDevice(ctx->device));
    const size_t n16 = (size_t)sl->n_px / 4;   // records are 4 bytes; the arena is hipMalloc-aligned
    if (n16 < ((size_t)1 << 20)) return fail(CDS_ERR_INVALID_ARGUMENT, "the library is too small for a bandwidth measurement (< 16 MB of records)");
    unsigned long long *d = nullptr;
    CDS_HIP(dev_malloc(&d, sizeof(unsigned long long)));
    hipError_t e = hipMemsetAsync(d, 0, sizeof(unsigned long long), ctx->stream);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    float best = 0;
    for (int rep = 0; rep < 4 && e == hipSuccess; rep++) {   // first repetition warms up; keep the fastest
        e = hipEventRecord(e0, ctx->stream);
        hipLaunchKernelGGL(stream_read_kernel, dim3((unsigned)ctx->n_cus * 16), dim3(256), 0, ctx->stream, reinterpret_cast<const u32x4 *>(sl->d_px), n16, d);
        if (e == hipSuccess) e = hipGetLastError();
        if (e == hipSuccess) e = hipEventRecord(e1, ctx->stream);
        if (e == hipSuccess) e = hipEventSynchronize(e1);
        float ms = 0;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && (best == 0 || ms < best)) best = ms;
    }
    if (e0) hipEventDestroy(e0);
    if (e1) hipEventDestroy(e1);
    hipFree(d);
    if (e != hipSuccess) return hip_fail(e, "stream read measurement");
    *gb_per_s = (double)n16 * 16.0 / ((double)best * 1e-3) / 1e9;
    if (bytes_read) *bytes_read = (double)n16 * 16.0;
    return CDS_OK;
}

extern "C" int cds_debug_valu_issue_ns(cds_context *ctx, double *ns_per_wave_instruction_per_simd) {
    if (!ctx || !ns_per_wave_instruction_per_simd) return fail(CDS_ERR_INVALID_ARGUMENT, "NULL argument");
    CDS_HIP(hipSetDevice(ctx->device));
    const int wgs = ctx->n_cus * 2, threads = 1024, iters = 4000;   // 2 x 16 waves per CU = 8 waves per SIMD
    uint32_t *d = nullptr;
    CDS_HIP(dev_malloc(&d, (size_t)wgs * threads * sizeof(uint32_t)));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    float best = 0;
    for (int rep = 0; rep < 3 && e == hipSuccess; rep++) {   // first repetition warms up; keep the fastest
        if (e == hipSuccess) e = hipEventRecord(e0, ctx->stream);
        hipLaunchKernelGGL(valu_issue_kernel, dim3(wgs), dim3(threads), 0, ctx->stream, d, iters);
        if (e == hipSuccess) e = hipGetLastError();
        if (e == hipSuccess) e = hipEventRecord(e1, ctx->stream);
        if (e == hipSuccess) e = hipEventSynchronize(e1);
        float ms = 0;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && (best == 0 || ms < best)) best = ms;
    }
    if (e0) hipEventDestroy(e0);
    if (e1) hipEventDestroy(e1);
    hipFree(d);
    if (e != hipSuccess) return hip_fail(e, "valu issue calibration");
    const double waves_per_simd = 2.0 * (threads / 64) / 4.0, valu_per_wave = 8.0 * 16 * iters;
    *ns_per_wave_instruction_per_simd = (double)best * 1e6 / (valu_per_wave * waves_per_simd);
    return CDS_OK;
}
