/*
 * cds_synth_oracle.c -- CPU twin of the synthetic-library generator.  TEST INFRASTRUCTURE ONLY
 * (same rules as cds_oracle.c: only tests/, smoke() and bench.py's cpu_baseline leg may use it).
 *
 * The reference has no synthetic generator (its tests use 11 real images); BASELINE.json asks for
 * "synthetic 1210x566 RGB libraries", SURVEY.md section 8(d) sketches them.  The generator is part of
 * the product (colormipsearch_amd/csrc/cds_synth.hip, a HIP kernel: every pixel is a pure function
 * of (seed, image index, x, y)); this file restates that specification on the CPU so that tests can
 * check the device generator and feed identical inputs to the oracle.
 *
 * Specification (all integer arithmetic, 64-bit wrap-around, arithmetic right shift):
 *   mix64(v)  : v ^= v>>30; v *= 0xBF58476D1CE4E5B9; v ^= v>>27; v *= 0x94D049BB133111EB; v ^= v>>31
 *   draw(i,k,j) = mix64(seed ^ ((i+1) * 0xD1B54A32D192ED03) ^ ((k+1) * 0x9E3779B97F4A7C15) ^ ((j+1) * 0xC2B2AE3D27D4EB4F))
 *   image i has K strokes; stroke k is described by r0 = draw(i,k,0), r1 = draw(i,k,1), r2 = draw(i,k,2):
 *     orient = r0 & 1 (0: across = y as a function of along = x; 1: the transpose)
 *     hw     = hw_base + ((r0 >> 1) & 3)
 *     z0 = (r0 >> 8) & 255; dz = ((r0 >> 16) & 255) - 128; bright = 77 + ((r0 >> 24) & 0xffff) % 179
 *     slope = ((r0 >> 40) & 511) - 256; amp = (r0 >> 49) & 127; freq = 1 + ((r0 >> 56) & 15); phase = r1 >> 56
 *     a0 = lo_a + (r1 & 0xffffffff) % (hi_a - lo_a); len = len_min + ((r1 >> 32) & 0xffffff) % len_span
 *     a1 = min(a0 + len, hi_a); c0 = lo_c + (r2 & 0xffffffff) % (hi_c - lo_c)
 *     f(t)   = c0 + (((t-a0) * slope) >> 8) + ((tri((t-a0)*freq + phase) * amp) >> 7),  tri(u) = |(u & 255) - 128| - 64
 *     z(t)   = clamp(z0 + (((t-a0) * dz) >> 8), 0, 255); colour c = (LUT[z][c] * bright + 127) / 255
 *     pixel (x,y) belongs to stroke k iff a0 <= t < a1 and |o - f(t)| <= hw   (t,o = x,y or y,x by orient)
 *   pixel value = colour of the belonging stroke with the largest k, else black.
 *   kind 0 "sparse-lut" target: K = 3 + draw(i,255,0) % 10, hw_base 2, whole frame, len 300..1199; the
 *        reference's label regions (x<330&&y<100, x>=W-270&&y<90) are left black.
 *   kind 1 "dense-uniform" target: r,g,b = low three bytes of mix64(seed ^ (i+1)*0xD1B5.. ^ (pixel+1)*0x9E37..).
 *   kind 2 "arbor" mask: K = 11, hw_base 4, strokes and pixels confined to the box cols 316..893, rows 165..334
 *        (bbox of the reference fixture 12191_JRC2018U_FL.tif), len 200..499, plus four anchor dashes of LUT[128]
 *        that pin the bounding box: row 165 and row 334 at x 600..615, col 316 and col 893 at y 250..257.
 *   variants of a kind-0 target for the shape scorer:
 *     zgap : same strokes dilated by 10 px (|o - f(tc)| <= hw+10, a0-10 <= t < a1+10, tc = clamp(t,a0,a1-1)),
 *            colour taken at tc, label regions black;
 *     grad : u16 = min(600, 2 * min_k( max(|o - f(tc)| - hw, 0) + max(a0 - t, t - (a1-1), 0) ))  (600 if K == 0).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

const uint8_t *cdso_depth_lut(void);

static uint64_t mix64(uint64_t v) {
    v ^= v >> 30; v *= 0xBF58476D1CE4E5B9ULL;
    v ^= v >> 27; v *= 0x94D049BB133111EBULL;
    v ^= v >> 31;
    return v;
}
static uint64_t draw(uint64_t seed, uint64_t i, uint64_t k, uint64_t j) {
    return mix64(seed ^ ((i + 1) * 0xD1B54A32D192ED03ULL) ^ ((k + 1) * 0x9E3779B97F4A7C15ULL) ^
                 ((j + 1) * 0xC2B2AE3D27D4EB4FULL));
}

typedef struct {
    int orient, hw, z0, dz, bright, slope, amp, freq, phase, a0, a1, c0;
} stroke;

static int sra(int v, int s) { return v >> s; } /* arithmetic on gcc/clang */
static int iabs(int v) { return v < 0 ? -v : v; }
static int tri(int u) { return iabs((u & 255) - 128) - 64; }
static int f_of(const stroke *s, int t) {
    return s->c0 + sra((t - s->a0) * s->slope, 8) + sra(tri((t - s->a0) * s->freq + s->phase) * s->amp, 7);
}
static int z_of(const stroke *s, int t) {
    int z = s->z0 + sra((t - s->a0) * s->dz, 8);
    return z < 0 ? 0 : (z > 255 ? 255 : z);
}

static void make_stroke(stroke *s, uint64_t seed, uint64_t i, int k, int hw_base, int lox, int hix, int loy, int hiy,
                        int len_min, int len_span) {
    uint64_t r0 = draw(seed, i, (uint64_t)k, 0), r1 = draw(seed, i, (uint64_t)k, 1), r2 = draw(seed, i, (uint64_t)k, 2);
    s->orient = (int)(r0 & 1);
    s->hw = hw_base + (int)((r0 >> 1) & 3);
    s->z0 = (int)((r0 >> 8) & 255);
    s->dz = (int)((r0 >> 16) & 255) - 128;
    s->bright = 77 + (int)(((r0 >> 24) & 0xffff) % 179);
    s->slope = (int)((r0 >> 40) & 511) - 256;
    s->amp = (int)((r0 >> 49) & 127);
    s->freq = 1 + (int)((r0 >> 56) & 15);
    s->phase = (int)(r1 >> 56);
    int lo_a = s->orient ? loy : lox, hi_a = s->orient ? hiy : hix;
    int lo_c = s->orient ? lox : loy, hi_c = s->orient ? hix : hiy;
    s->a0 = lo_a + (int)((r1 & 0xffffffffULL) % (uint64_t)(hi_a - lo_a));
    int len = len_min + (int)(((r1 >> 32) & 0xffffff) % (uint64_t)len_span);
    s->a1 = s->a0 + len < hi_a ? s->a0 + len : hi_a;
    s->c0 = lo_c + (int)((r2 & 0xffffffffULL) % (uint64_t)(hi_c - lo_c));
}

static int in_label(int W, int x, int y) { return (x < 330 && y < 100) || (x >= W - 270 && y < 90); }

static void put(uint8_t *img, int W, int x, int y, const uint8_t *lut, int z, int bright) {
    uint8_t *p = img + 3 * ((size_t)y * W + x);
    for (int c = 0; c < 3; c++) p[c] = (uint8_t)((lut[3 * z + c] * bright + 127) / 255);
}

/* paint strokes in increasing k: later strokes overwrite == "largest k wins" */
static void paint(uint8_t *img, int W, int H, const stroke *s, int K, int dil, int lox, int hix, int loy, int hiy,
                  int skip_labels) {
    const uint8_t *lut = cdso_depth_lut();
    for (int k = 0; k < K; k++) {
        const stroke *q = &s[k];
        for (int t = q->a0 - dil; t < q->a1 + dil; t++) {
            int tc = t < q->a0 ? q->a0 : (t > q->a1 - 1 ? q->a1 - 1 : t);
            if (q->a1 <= q->a0) continue;
            int f = f_of(q, tc), z = z_of(q, tc);
            for (int o = f - q->hw - dil; o <= f + q->hw + dil; o++) {
                int x = q->orient ? o : t, y = q->orient ? t : o;
                if (x < lox || x >= hix || y < loy || y >= hiy) continue;
                if (x < 0 || x >= W || y < 0 || y >= H) continue;
                if (skip_labels && in_label(W, x, y)) continue;
                put(img, W, x, y, lut, z, q->bright);
            }
        }
    }
}

static int target_strokes(stroke *s, uint64_t seed, uint64_t i, int W, int H) {
    int K = 3 + (int)(draw(seed, i, 255, 0) % 10);
    for (int k = 0; k < K; k++) make_stroke(&s[k], seed, i, k, 2, 0, W, 0, H, 300, 900);
    return K;
}

/* kind: 0 sparse-lut target, 1 dense-uniform target, 2 arbor mask.  out: packed RGB 3*W*H. */
void cdso_synth_image(uint64_t seed, uint64_t index, int kind, int W, int H, uint8_t *out) {
    stroke s[16];
    if (kind == 1) {
        for (size_t p = 0; p < (size_t)W * H; p++) {
            uint64_t h = mix64(seed ^ ((index + 1) * 0xD1B54A32D192ED03ULL) ^ ((uint64_t)(p + 1) * 0x9E3779B97F4A7C15ULL));
            out[3 * p] = (uint8_t)h;
            out[3 * p + 1] = (uint8_t)(h >> 8);
            out[3 * p + 2] = (uint8_t)(h >> 16);
        }
        return;
    }
    memset(out, 0, (size_t)3 * W * H);
    if (kind == 0) {
        int K = target_strokes(s, seed, index, W, H);
        paint(out, W, H, s, K, 0, 0, W, 0, H, 1);
    } else {
        const int lox = 316, hix = 894, loy = 165, hiy = 335;
        for (int k = 0; k < 11; k++) make_stroke(&s[k], seed, index, k, 4, lox, hix, loy, hiy, 200, 300);
        paint(out, W, H, s, 11, 0, lox, hix, loy, hiy, 0);
        const uint8_t *lut = cdso_depth_lut();
        for (int x = 600; x < 616; x++) { put(out, W, x, 165, lut, 128, 255); put(out, W, x, 334, lut, 128, 255); }
        for (int y = 250; y < 258; y++) { put(out, W, 316, y, lut, 128, 255); put(out, W, 893, y, lut, 128, 255); }
    }
}

/* zgap (RGB) and gradient (u16) variants of kind-0 target `index`. Either output may be NULL. */
void cdso_synth_variants(uint64_t seed, uint64_t index, int W, int H, uint8_t *zgap, uint16_t *grad) {
    stroke s[16];
    int K = target_strokes(s, seed, index, W, H);
    if (zgap) {
        memset(zgap, 0, (size_t)3 * W * H);
        paint(zgap, W, H, s, K, 10, 0, W, 0, H, 1);
    }
    if (grad) {
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) {
                int best = 300;
                for (int k = 0; k < K; k++) {
                    const stroke *q = &s[k];
                    if (q->a1 <= q->a0) continue;
                    int t = q->orient ? y : x, o = q->orient ? x : y;
                    int tc = t < q->a0 ? q->a0 : (t > q->a1 - 1 ? q->a1 - 1 : t);
                    int d = iabs(o - f_of(q, tc)) - q->hw;
                    if (d < 0) d = 0;
                    int e = q->a0 - t > t - (q->a1 - 1) ? q->a0 - t : t - (q->a1 - 1);
                    if (e < 0) e = 0;
                    if (d + e < best) best = d + e;
                }
                grad[(size_t)y * W + x] = (uint16_t)(2 * best > 600 ? 600 : 2 * best);
            }
    }
}
