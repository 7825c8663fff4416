// Host-only AddressSanitizer run of the image decoder: mutated copies of real files through cds_image_probe/decode.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <string>
#include <random>
#include "cds_hip.h"
static std::vector<uint8_t> slurp(const char *p) {
    std::vector<uint8_t> v; FILE *f = fopen(p, "rb"); if (!f) return v;
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET); v.resize(n); if (fread(v.data(), 1, n, f) != (size_t)n) v.clear(); fclose(f); return v;
}
int main(int argc, char **argv) {
    int iters = atoi(argv[1]);
    std::mt19937_64 rng(12345);
    long ok = 0, bad = 0;
    for (int a = 2; a < argc; a++) {
        std::vector<uint8_t> src = slurp(argv[a]);
        if (src.empty()) { fprintf(stderr, "cannot read %s\n", argv[a]); return 2; }
        for (int it = 0; it < iters; it++) {
            // exact-size heap copy so that any over-read trips the sanitizer
            size_t n = src.size();
            int mode = it % 4;
            if (mode == 1) n = rng() % (src.size() + 1);          // truncation
            uint8_t *buf = (uint8_t *)malloc(n ? n : 1);
            memcpy(buf, src.data(), n);
            if (it > 0 && n > 0) {
                int flips = mode == 2 ? 1 + rng() % 4 : (mode == 3 ? 1 + rng() % 64 : 0);
                for (int k = 0; k < flips; k++) {
                    size_t at = (mode == 2) ? rng() % std::min<size_t>(n, 512) : rng() % n;   // headers / anywhere
                    buf[at] = (uint8_t)rng();
                }
            }
            cds_image_info info;
            if (cds_image_probe(buf, n, &info) == 0) {
                size_t bytes = (size_t)info.width * info.height * info.channels * (info.bits / 8);
                if (bytes > 0 && bytes < ((size_t)1 << 31)) {
                    uint8_t *out = (uint8_t *)malloc(bytes);
                    if (cds_image_decode(buf, n, out, bytes) == 0) ok++; else bad++;
                    free(out);
                } else bad++;
            } else bad++;
            free(buf);
        }
    }
    printf("decoded %ld, refused %ld\n", ok, bad);
    return 0;
}
