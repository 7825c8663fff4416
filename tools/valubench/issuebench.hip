// issuebench.hip -- issue rates of exact instruction mixes on gfx950 (inline asm, independent instructions), 8 waves per SIMD:
// does scalar work share issue bandwidth with vector work?  Calibrates the instruction-issue model of DESIGN.md.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
template <int MODE>
__global__ __launch_bounds__(1024) void k(uint32_t *out, int iters) {
    uint32_t v0 = threadIdx.x, v1 = 1, v2 = 2, v3 = 3;
    uint32_t s0 = blockIdx.x, s1 = 1, s2 = 2, s3 = 3;
    __shared__ uint32_t lds[2048];
    lds[threadIdx.x] = threadIdx.x; lds[threadIdx.x + 1024] = 0;
    __syncthreads();
    uint32_t l0 = 0, addr = (threadIdx.x & 1023) * 4;
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {  // 64 VALU
            REP16(asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(l0));)
        } else if (MODE == 1) {  // 64 SALU
            REP16(asm volatile("s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1" : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) :: "scc");)
        } else if (MODE == 2) {  // 64 VALU + 64 SALU interleaved
            REP16(asm volatile("v_add_u32 %0, %0, %8\n s_add_u32 %4, %4, 1\n v_add_u32 %1, %1, %8\n s_add_u32 %5, %5, 1\n v_add_u32 %2, %2, %8\n s_add_u32 %6, %6, 1\n v_add_u32 %3, %3, %8\n s_add_u32 %7, %7, 1"
                               : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "v"(l0) : "scc");)
        } else if (MODE == 3) {  // 64 VALU + 32 SALU
            REP16(asm volatile("v_add_u32 %0, %0, %6\n s_add_u32 %4, %4, 1\n v_add_u32 %1, %1, %6\n v_add_u32 %2, %2, %6\n s_add_u32 %5, %5, 1\n v_add_u32 %3, %3, %6"
                               : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+s"(s0), "+s"(s1) : "v"(l0) : "scc");)
        } else if (MODE == 4) {  // 64 VALU + 16 ds_read_b32 (independent)
            REP16(asm volatile("v_add_u32 %0, %0, %5\n v_add_u32 %1, %1, %5\n ds_read_b32 %4, %6\n v_add_u32 %2, %2, %5\n v_add_u32 %3, %3, %5" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "=v"(l0) : "v"(v1), "v"(addr));)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (MODE == 5) {  // 32 x (v_cmp -> s_bcnt1 -> s_add): the counting chain of the scan's rounds
            REP16(asm volatile("v_cmp_gt_u32 vcc, %0, %1\n s_bcnt1_i32_b64 %3, vcc\n s_add_u32 %2, %2, %3\n v_cmp_gt_u32 vcc, %1, %0\n s_bcnt1_i32_b64 %3, vcc\n s_add_u32 %2, %2, %3" : "+v"(v0), "+v"(v1), "+s"(s0), "+s"(s1) :: "vcc", "scc");)
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = v0 + v1 + v2 + v3 + s0 + s1 + s2 + s3 + l0;
}
int main() {
    uint32_t *d;
    (void)hipMalloc(&d, 1024 * 1024 * 4 * 4);
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount, iters = 4000;
    const char *names[] = {"64 VALU", "64 SALU", "64 VALU + 64 SALU", "64 VALU + 32 SALU", "64 VALU + 16 ds_read_b32", "32 x (v_cmp, s_bcnt1, s_add)"};
    const double instr[] = {64, 64, 128, 96, 80, 96};
    for (int mode = 0; mode < 6; mode++)
        for (int wg_per_cu = 1; wg_per_cu <= 2; wg_per_cu++) {
            hipEvent_t e0, e1;
            (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            auto kern = mode == 0 ? k<0> : mode == 1 ? k<1> : mode == 2 ? k<2> : mode == 3 ? k<3> : mode == 4 ? k<4> : k<5>;
            hipLaunchKernelGGL(kern, dim3(cus * wg_per_cu), dim3(1024), 0, 0, d, 10);
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(cus * wg_per_cu), dim3(1024), 0, 0, d, iters);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            const double waves_per_simd = wg_per_cu * 4.0;
            const double ns_iter = ms * 1e6 / iters / waves_per_simd;   // ns of SIMD time per wave-iteration
            printf("%-30s %d waves/SIMD: %7.3f ms  %6.1f ns per wave-iteration per SIMD = %5.2f ns per instruction\n", names[mode], (int)waves_per_simd, ms, ns_iter, ns_iter / instr[mode]);
        }
    return 0;
}
