// valubench.hip -- how many wave64 integer VALU instructions per cycle does one gfx950 SIMD issue with W waves resident?
// (calibrates the instruction-issue model used in DESIGN.md for the scan kernels)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int MODE>
__global__ __launch_bounds__(1024) void k(uint32_t *out, int iters) {
    uint32_t a = threadIdx.x, b = a * 3 + 1, c = a ^ 5, d = a + 7, e = a * 5, f = a + 11, g = a ^ 9, h = a + 3;
    unsigned long long sacc = 0;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            if (MODE == 0) {  // 8 independent chains of v_add/v_xor
                a += b; b ^= c; c += d; d ^= e; e += f; f ^= g; g += h; h ^= a;
            } else if (MODE == 1) {  // compare + ballot + s_bcnt1 + s_add mix (like the scan's rounds)
                a += b; b ^= c;
                sacc += __popcll(__ballot(a > b));
                c += d; d ^= e;
                sacc += __popcll(__ballot(c > d));
            } else {  // dependent chain
                a += b; a ^= c; a += d; a ^= e; a += f; a ^= g; a += h; a ^= b;
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + e + f + g + h + (uint32_t)sacc;
}
int main() {
    uint32_t *d;
    hipMalloc(&d, 1024 * 1024 * 4 * 4);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const int iters = 20000;
    for (int mode = 0; mode < 3; mode++)
        for (int wg_per_cu = 1; wg_per_cu <= 2; wg_per_cu++)
            for (int threads : {256, 512, 1024}) {
                hipEvent_t e0, e1;
                hipEventCreate(&e0); hipEventCreate(&e1);
                auto kern = mode == 0 ? k<0> : mode == 1 ? k<1> : k<2>;
                hipLaunchKernelGGL(kern, dim3(cus * wg_per_cu), dim3(threads), 0, 0, d, 100);
                hipEventRecord(e0);
                hipLaunchKernelGGL(kern, dim3(cus * wg_per_cu), dim3(threads), 0, 0, d, iters);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                const double waves_per_simd = wg_per_cu * threads / 64 / 4.0;
                const double valu_per_wave = (mode == 1 ? 6.0 : 8.0) * 16 * iters;  // mode 1: 4 alu + 2 cmp per unrolled step
                const double ns_per_instr_simd = ms * 1e6 / (valu_per_wave * waves_per_simd);
                printf("mode %d waves/SIMD %.1f: %.3f ms, %.3f ns per wave-VALU-instruction per SIMD (= %.2f cycles at 2.4 GHz)\n", mode,
                       waves_per_simd, ms, ns_per_instr_simd, ns_per_instr_simd * 2.4);
            }
    return 0;
}
