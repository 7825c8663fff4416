// ldsalign.hip -- does gfx950 serve UNALIGNED ds_read_b64 (byte-granular addresses) correctly?  Each lane reads 8 bytes at an
// arbitrary byte offset of a known LDS pattern through inline asm (the compiler would split a misaligned access); the host
// compares with the pattern.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__global__ void k(uint32_t *out, int shift) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = (uint8_t)(i * 7 + 3);
    __syncthreads();
    const uint32_t off = (threadIdx.x * 37 + shift) % 8000;   // arbitrary byte offsets
    const uint32_t addr = (uint32_t)(uintptr_t)lds + off;
    u32x2 v;
    asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    out[threadIdx.x * 3] = v.x;
    out[threadIdx.x * 3 + 1] = v.y;
    out[threadIdx.x * 3 + 2] = off;
}
int main() {
    uint32_t *d, h[256 * 3];
    (void)hipMalloc(&d, sizeof h);
    int bad = 0, total = 0;
    for (int shift = 0; shift < 8; shift++) {
        hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, d, shift);
        (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        for (int t = 0; t < 256; t++) {
            const uint32_t off = h[t * 3 + 2];
            uint64_t want = 0;
            for (int b = 0; b < 8; b++) want |= (uint64_t)(uint8_t)((off + b) * 7 + 3) << (8 * b);
            const uint64_t got = (uint64_t)h[t * 3] | (uint64_t)h[t * 3 + 1] << 32;
            total++;
            if (got != want) {
                if (bad < 5) printf("off %u (mod 8 = %u): got %016llx want %016llx\n", off, off & 7, (unsigned long long)got, (unsigned long long)want);
                bad++;
            }
        }
    }
    printf("unaligned ds_read_b64: %d of %d reads wrong\n", bad, total);
    return 0;
}
