// fetchcal.hip -- what does rocprofv3's FETCH_SIZE count for SCATTERED loads on gfx950?   
//
// MI355X_MICROARCH.md establishes that FETCH_SIZE reports HALF the bytes of a wide coalesced streaming read (128-B requests tallied
// at 64 B) and prescribes "x 2".  Two kernels of this repository are dominated by scattered loads instead: the GradientAreaGap list
// scorer (12-byte groups at scattered pixels: gradscore_list_kernel) and the evaluation rounds of the stream scan (one 4-byte record
// per lane).  This tool issues exactly those shapes over a KNOWN set of bytes far larger than the 256 MB Infinity Cache, every sector
// touched exactly once, so that the right factor between the counter and the bytes the memory system must move can be read off:
//   mode 0  stream   : 16 B per lane, consecutive (the calibrated case: factor 2 expected)
//   mode 1  gather4  : one dword per lane, every lane in a 64-B sector of its own (a round's gather when entries are sparse)
//   mode 2  gather12 : 12 B per lane at a 12-byte-aligned offset inside a 64-B sector of its own (the scorer's target / zgap groups)
//   mode 3  gather4x4: four lanes share a 64-B sector (4-byte loads 16 B apart: a round's gather at ~25 % survivors)
//   mode 4  gather8  : 8 B per lane (the stream kernel's plane rows from L2 / the scorer's zslice + gradient pairs), own sector
//   mode 5 / 6       : modes 1 / 2 WITHOUT the non-temporal hint -- the cache policy of this repository's kernels' gathers: L2 allocates
//                      the whole 128-B line, so the request is 128 B wide although the lane wants 4 or 12 bytes
// For every mode the tool prints the bytes REQUESTED (lanes x width), the bytes in the 32-B / 64-B / 128-B granules touched, and the
// kernel time; run under `rocprofv3 --pmc FETCH_SIZE TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum` (tools/fetchcal/run.sh) the summary
// divides the granule bytes by FETCH_SIZE x 1024.
// Build: hipcc -O3 --offload-arch=gfx950 fetchcal.hip -o fetchcal ; run: ./fetchcal [GiB of buffer, default 6]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// Every lane visits "its" granule of STRIDE bytes once; the order of granules is scrambled per wave-instruction by an odd multiplier
// modulo a power of two (a bijection), so that neighbouring lanes touch far-apart lines and nothing is served twice from a cache.
template <int MODE>
__global__ __launch_bounds__(256) void cal(const uint8_t *buf, uint64_t n_granules, uint32_t *out) {
    constexpr uint64_t STRIDE = MODE == 0 ? 16 : MODE == 3 ? 16 : 64;
    uint32_t acc = 0;
    const uint64_t mask = n_granules - 1;   // power of two
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n_granules; i += (uint64_t)gridDim.x * 256) {
        uint64_t g = i;
        if (MODE == 1 || MODE == 2 || MODE == 4 || MODE == 5 || MODE == 6) g = (i * 0x9E3779B97F4A7C15ull) & mask;   // own sector, scattered
        if (MODE == 3) g = ((((i >> 2) * 0x9E3779B97F4A7C15ull) & (mask >> 2)) << 2) | (i & 3);      // 4 neighbours share a 64-B sector
        const uint8_t *p = buf + g * STRIDE;
        if (MODE == 0) { const u32x4 v = __builtin_nontemporal_load((const u32x4 *)p); acc ^= v.x ^ v.y ^ v.z ^ v.w; }
        if (MODE == 1 || MODE == 3) acc ^= __builtin_nontemporal_load((const uint32_t *)(p + 4 * (g % 3)));
        if (MODE == 2) { const u32x3 v = __builtin_nontemporal_load((const u32x3 *)(p + 12 * (g & 3))); acc ^= v.x ^ v.y ^ v.z; }
        if (MODE == 4) { const u32x2 v = __builtin_nontemporal_load((const u32x2 *)(p + 8 * (g & 7))); acc ^= v.x ^ v.y; }
        if (MODE == 5) acc ^= *(const uint32_t *)(p + 4 * (g % 3));
        if (MODE == 6) { const u32x3 v = *(const u32x3 *)(p + 12 * (g & 3)); acc ^= v.x ^ v.y ^ v.z; }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int MODE>
static void run(const char *name, const uint8_t *buf, uint64_t bytes, int width, uint64_t stride, uint32_t *out) {
    uint64_t n = 1;
    while (n * 2 * stride <= bytes) n *= 2;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(cal<MODE>, dim3(256 * 16), dim3(256), 0, 0, buf, n, out);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double req = (double)n * width;
    // granules touched: mode 0 touches everything contiguously; own-sector modes touch one 64-B sector (and the 32-B half / 128-B line
    // around it) per lane; mode 3 one sector per 4 lanes
    const double lanes_per_sector = MODE == 0 ? 4.0 : MODE == 3 ? 4.0 : 1.0;
    const double sectors = (double)n / lanes_per_sector;
    const double b32 = MODE == 0 ? req : MODE == 3 ? sectors * 64.0 : sectors * (width > 20 ? 64.0 : 32.0);   // (12 B at offsets 0/12/24/36: within one or two 32-B halves)
    printf("{\"mode\": \"%s\", \"kernel\": \"cal<%d>\", \"lanes\": %llu, \"bytes_requested\": %.0f, \"bytes_32B_granules_min\": %.0f, \"bytes_64B_sectors\": %.0f, "
           "\"bytes_128B_lines\": %.0f, \"ms\": %.3f, \"GBps_by_64B_sectors\": %.1f}\n",
           name, MODE, (unsigned long long)n, req, b32, sectors * 64.0, MODE == 0 ? req : (MODE == 3 ? sectors * 64.0 : sectors * 128.0), ms,
           sectors * 64.0 / (ms * 1e-3) / 1e9);
}

int main(int argc, char **argv) {
    const double gib = argc > 1 ? atof(argv[1]) : 6.0;
    const uint64_t bytes = (uint64_t)(gib * (1ull << 30));
    uint8_t *buf = nullptr;
    uint32_t *out = nullptr;
    CHECK(hipMalloc(&buf, bytes));
    CHECK(hipMalloc(&out, 64));
    CHECK(hipMemset(buf, 1, bytes));
    CHECK(hipDeviceSynchronize());
    run<0>("stream16", buf, bytes, 16, 16, out);
    run<1>("gather4", buf, bytes, 4, 64, out);
    run<2>("gather12", buf, bytes, 12, 64, out);
    run<3>("gather4x4", buf, bytes, 4, 16, out);
    run<4>("gather8", buf, bytes, 8, 64, out);
    run<5>("gather4_default_policy", buf, bytes, 4, 64, out);
    run<6>("gather12_default_policy", buf, bytes, 12, 64, out);
    CHECK(hipDeviceSynchronize());
    return 0;
}
