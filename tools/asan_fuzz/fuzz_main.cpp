// Host-only AddressSanitizer run of the image decoder: mutated copies of real files through cds_image_probe/decode.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <string>
#include <random>
#include <algorithm>
#include "cds_hip.h"
static std::vector<uint8_t> slurp(const char *p) {
    std::vector<uint8_t> v; FILE *f = fopen(p, "rb"); if (!f) return v;
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET); v.resize(n); if (fread(v.data(), 1, n, f) != (size_t)n) v.clear(); fclose(f); return v;
}
static void spill(const std::string &path, const uint8_t *p, size_t n) {
    FILE *f = fopen(path.c_str(), "wb");
    if (f) { fwrite(p, 1, n, f); fclose(f); }
}
// a mutated copy of `src` (iteration `it`): untouched / truncated / a few header bytes / bytes anywhere (tail = the archive's directories)
static std::vector<uint8_t> mutate(const std::vector<uint8_t> &src, int it, std::mt19937_64 &rng, bool tail_is_header) {
    size_t n = src.size();
    const int mode = it % 4;
    if (mode == 1) n = rng() % (src.size() + 1);
    std::vector<uint8_t> buf(src.begin(), src.begin() + (std::ptrdiff_t)n);
    if (it > 0 && n > 0) {
        const int flips = mode == 2 ? 1 + (int)(rng() % 4) : (mode == 3 ? 1 + (int)(rng() % 64) : 0);
        for (int k = 0; k < flips; k++) {
            const size_t span = std::min<size_t>(n, 512);
            const size_t at = mode == 2 ? (tail_is_header ? n - 1 - rng() % span : rng() % span) : rng() % n;
            buf[at] = (uint8_t)rng();
        }
    }
    return buf;
}
// zip archives: mutated copies through the archive reader + decoders + the variant shard's packer (all host code), then the shard that
// was written -- and mutated copies of it -- through cds_variant_shard_open
static int fuzz_zip(const char *seed, int iters, std::mt19937_64 &rng, long &ok, long &bad) {
    std::vector<uint8_t> src = slurp(seed);
    if (src.empty()) { fprintf(stderr, "cannot read %s\n", seed); return 2; }
    for (int it = 0; it < iters; it++) {
        const std::string zp = "fz_" + std::to_string(it % 16) + ".zip";   // (16 names: the reader keeps the 8 most recent archives)
        const std::vector<uint8_t> buf = mutate(src, it, rng, true);
        spill(zp, buf.data(), buf.size());
        cds_variant_writer *w = nullptr;
        if (cds_variant_shard_create("fz.cdsv", 64, 48, &w) != 0) return 3;
        const std::string t = zp + "!mips/a.tif", t2 = zp + "!/elsewhere/b.png", g = zp + "!grad/a.png";
        const char *tp[2] = {t.c_str(), t2.c_str()}, *gp[2] = {g.c_str(), nullptr};
        const int rc = cds_variant_shard_append_files(w, 2, tp, gp, nullptr, 1);
        rc == 0 ? ok++ : bad++;
        if (cds_variant_shard_finish(w) != 0) bad++;
        if (rc == 0) {
            std::vector<uint8_t> sh = slurp("fz.cdsv");
            for (int k = 0; k < 8; k++) {
                const std::vector<uint8_t> m = mutate(sh, k, rng, k & 1);
                spill("fz_m.cdsv", m.data(), m.size());
                cds_variant_shard *s = nullptr;
                if (cds_variant_shard_open("fz_m.cdsv", &s) == 0) {
                    int32_t W = 0, H = 0;
                    int64_t n = 0, bytes = 0;
                    cds_variant_shard_info(s, &W, &H, &n, &bytes);
                    uint32_t f[2] = {0, 0}, b[2] = {0, 0};
                    if (n >= 2) cds_variant_shard_target_info(s, 0, 2, f, b);
                    cds_variant_shard_close(s);
                    ok++;
                } else bad++;
            }
        }
    }
    return 0;
}

int main(int argc, char **argv) {
    int iters = atoi(argv[1]);
    std::mt19937_64 rng(12345);
    long ok = 0, bad = 0;
    for (int a = 2; a < argc; a++) {
        const size_t len = strlen(argv[a]);
        if (len > 4 && strcmp(argv[a] + len - 4, ".zip") == 0) {
            const int rc = fuzz_zip(argv[a], std::max(1, iters / 10), rng, ok, bad);
            if (rc) return rc;
            continue;
        }
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
