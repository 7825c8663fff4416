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

}  // namespace cds

using namespace cds;

extern "C" int cds_debug_valu_issue_ns(cds_context *ctx, double *ns_per_wave_instruction_per_simd) {
    if (!ctx || !ns_per_wave_instruction_per_simd) return fail(CDS_ERR_INVALID_ARGUMENT, "NULL argument");
    CDS_HIP(hipSetDevice(ctx->device));
    const int wgs = ctx->n_cus * 2, threads = 1024, iters = 4000;   // 2 x 16 waves per CU = 8 waves per SIMD
    uint32_t *d = nullptr;
    CDS_HIP(hipMalloc(&d, (size_t)wgs * threads * sizeof(uint32_t)));
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
