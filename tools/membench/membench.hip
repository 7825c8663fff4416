// membench.hip -- what limits the staging phase of the scan kernel?  Load-only variants of its access pattern
// (one workgroup per image, rows [165-2, 334+2], a column window of each 3648-byte row) on an MI355X.
// Build: hipcc -O3 --offload-arch=gfx950 membench.hip -o membench ; run: ./membench [n_images]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int PITCH = 3648, H = 566, R0 = 163, NR = 174;
constexpr size_t STRIDE = (size_t)PITCH * H;
constexpr int WOFF = 304 * 3, WBYTES = 592 * 3;  // column window: 1776 bytes at byte 912

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// MODE 0: dwordx3 nt, 64 lanes x 12 B per instruction, 3 segments per row (148 groups), one row per wave, 12-row bands + barriers
// MODE 1: same without barriers     MODE 2: same, no nt     MODE 3: dwordx4 nt over the same window (111 lanes)
// MODE 4: dwordx4 nt over FULL rows (228 lanes -> 4 instr/row)   MODE 5: dwordx4 over the whole image, linear
// MODE 6: like 0 but 24 rows in flight per band (6 per wave)
template <int MODE>
__global__ __launch_bounds__(256) void k(const uint8_t *lib, int n, uint32_t *out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t acc = 0;
    for (int t = blockIdx.x; t < n; t += gridDim.x) {
        const uint8_t *img = lib + (size_t)t * STRIDE;
        if (MODE == 5) {
            const u32x4 *p = (const u32x4 *)img;
            const int n16 = (int)(STRIDE / 16);
            for (int i = threadIdx.x; i < n16; i += 256 * 4) {
                u32x4 v[4];
#pragma unroll
                for (int j = 0; j < 4; j++) { int ii = i + j * 256; v[j] = __builtin_nontemporal_load(p + (ii < n16 ? ii : i)); }
#pragma unroll
                for (int j = 0; j < 4; j++) acc |= v[j].x | v[j].y | v[j].z | v[j].w;
            }
            continue;
        }
        constexpr int BAND = (MODE == 6) ? 24 : 12;
        constexpr int RPW = BAND / 4;
        for (int b = 0; b < NR; b += BAND) {
            if (MODE <= 2 || MODE == 6) {
                u32x3 v[RPW][3];
#pragma unroll
                for (int j = 0; j < RPW; j++) {
                    int row = R0 + b + wave + 4 * j; row = row < R0 + NR ? row : R0 + NR - 1;
                    const uint8_t *rp = img + (size_t)row * PITCH + WOFF;
#pragma unroll
                    for (int s = 0; s < 3; s++) {
                        int g = 64 * s + lane; g = g < 148 ? g : 0;
                        const u32x3 *q = (const u32x3 *)(rp + 12 * g);
                        v[j][s] = (MODE == 2) ? *q : __builtin_nontemporal_load(q);
                    }
                }
#pragma unroll
                for (int j = 0; j < RPW; j++)
#pragma unroll
                    for (int s = 0; s < 3; s++) acc |= v[j][s].x | v[j][s].y | v[j][s].z;
            } else if (MODE == 3) {
                u32x4 v[3][2];
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    int row = R0 + b + wave + 4 * j; row = row < R0 + NR ? row : R0 + NR - 1;
                    const uint8_t *rp = img + (size_t)row * PITCH + WOFF;
#pragma unroll
                    for (int s = 0; s < 2; s++) {
                        int g = 64 * s + lane; g = g < 111 ? g : 0;
                        v[j][s] = __builtin_nontemporal_load((const u32x4 *)(rp + 16 * g));
                    }
                }
#pragma unroll
                for (int j = 0; j < 3; j++)
#pragma unroll
                    for (int s = 0; s < 2; s++) acc |= v[j][s].x | v[j][s].y | v[j][s].z | v[j][s].w;
            } else if (MODE == 4) {
                u32x4 v[3][4];
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    int row = R0 + b + wave + 4 * j; row = row < R0 + NR ? row : R0 + NR - 1;
                    const uint8_t *rp = img + (size_t)row * PITCH;
#pragma unroll
                    for (int s = 0; s < 4; s++) {
                        int g = 64 * s + lane; g = g < 228 ? g : 0;
                        v[j][s] = __builtin_nontemporal_load((const u32x4 *)(rp + 16 * g));
                    }
                }
#pragma unroll
                for (int j = 0; j < 3; j++)
#pragma unroll
                    for (int s = 0; s < 4; s++) acc |= v[j][s].x | v[j][s].y | v[j][s].z | v[j][s].w;
            }
            if (MODE == 0 || MODE == 6) { __syncthreads(); __syncthreads(); __syncthreads(); }
        }
    }
    if (acc == 0x12345678u) out[blockIdx.x] = acc;  // keep the loads alive
}

template <int MODE>
static void run(const char *name, const uint8_t *lib, int n, uint32_t *out, double bytes_per_img, int wgs_per_cu) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    int grid = 256 * wgs_per_cu; if (grid > n) grid = n;
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, lib, n, out);
    CHECK(hipDeviceSynchronize());
    float best = 1e9;
    for (int r = 0; r < 3; r++) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, lib, n, out);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    printf("%-44s wgs/cu %3d  %8.3f ms  %7.2f TB/s requested  %6.2f M images/s\n", name, wgs_per_cu, best, bytes_per_img * n / best / 1e9,
           n / best / 1e3);
}

int main(int argc, char **argv) {
    int n = argc > 1 ? atoi(argv[1]) : 16384;
    uint8_t *lib; uint32_t *out;
    CHECK(hipMalloc(&lib, (size_t)n * STRIDE)); CHECK(hipMalloc(&out, 1 << 20));
    CHECK(hipMemset(lib, 0, (size_t)n * STRIDE));
    const double win = 148 * 12.0 * NR, win16 = 111 * 16.0 * NR, full = 228 * 16.0 * NR;
    for (int w : {4, 8, 16, 64}) {
        run<0>("x3 nt window, bands+barriers", lib, n, out, win, w);
        run<1>("x3 nt window, no barriers", lib, n, out, win, w);
        run<2>("x3 (no nt) window, no barriers", lib, n, out, win, w);
        run<3>("x4 nt window", lib, n, out, win16, w);
        run<4>("x4 nt full rows", lib, n, out, full, w);
        run<6>("x3 nt window, 24-row bands+barriers", lib, n, out, win, w);
    }
    run<5>("x4 nt whole image linear", lib, n, out, (double)STRIDE, 8);
    run<5>("x4 nt whole image linear", lib, n, out, (double)STRIDE, 16);
    return 0;
}
