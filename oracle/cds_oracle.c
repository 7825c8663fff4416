/*
 * cds_oracle.c -- CPU ORACLE. TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of the reference's (JaneliaSciComp/colormipsearch v3.1.1, 100% Java)
 * color-depth pixel-match scan and GradientAreaGap ("shape") scorer.  It exists to CHECK the
 * HIP path; nothing under colormipsearch_amd/ may include, link or call it.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks this file against every
 * known-answer value the reference's own JUnit tests hold for this path
 * (PixelMatchColorDepthSearchAlgorithmTest.java:48-52,72-103;
 *  Shape2DMatchColorDepthSearchAlgorithmTest.java:53-54,86-132,230-291;
 *  GradientAreaGapUtilsTest.java:31-39) on the reference's own fixture images
 * (tests/golden/cds_fixtures.npz, made by tests/golden/make_fixtures.py).
 *
 * The reference cannot be compiled here (no JDK in the image), so there is no oracle/_ref.
 *
 * Abbreviations for citations (paths under colormipsearch-api/src/main/java/org/janelia/colormipsearch/):
 *   ACDS  = cds/AbstractColorDepthSearchAlgorithm.java
 *   PM    = cds/PixelMatchColorDepthSearchAlgorithm.java
 *   S2D   = cds/Shape2DMatchColorDepthSearchAlgorithm.java
 *   GAG   = cds/GradientAreaGapUtils.java
 *   FAC   = cds/ColorDepthSearchAlgorithmProviderFactory.java
 *   CT    = imageprocessing/ColorTransformation.java
 *   IT    = imageprocessing/ImageTransformation.java
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off, no fast-math: the predicate is an
 * IEEE-754 double comparison and must not be contracted or re-associated).
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int x0, y0, x1, y1; /* half-open rectangle [x0,x1) x [y0,y1) */
} cdso_rect;

/* ------------------------------------------------------------------------------------------
 * Excluded ("label") regions.  The reference passes a predicate (ImageRegionDefinition); every
 * caller builds it as a union of rectangles (tools cmd/AbstractColorDepthMatchArgs.java:101-119,
 * test ImageTestUtils.java:12-24, PixelMatch...Test.java:45).
 * ------------------------------------------------------------------------------------------ */
static int in_rects(const cdso_rect *r, int n, int x, int y) {
    for (int i = 0; i < n; i++)
        if (x >= r[i].x0 && x < r[i].x1 && y >= r[i].y0 && y < r[i].y1) return 1;
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * calculatePixelGap -- ACDS:157-390.
 * Hue class by STRICT max (ties => no class => 10000); ratio = second/max in double, 0 when
 * the numerator is 0 (ACDS:195-256).  Same class & both ratios > 0 => |b-a| (ACDS:259-268 etc).
 * Adjacent classes: ACDS:269-276 (BR/BG), 284-298, 306-320, 328-342, 351-365, 373-381.
 * The "== 255" branches are dead (ratios < 1).  No RB<->BR wrap.
 * ------------------------------------------------------------------------------------------ */
enum { C_NONE = 0, C_BR = 1, C_BG = 2, C_GB = 3, C_GR = 4, C_RG = 5, C_RB = 6 };

static int hue_class(int r, int g, int b, double *ratio) {
    *ratio = 0.0;
    if (b > r && b > g) {
        if (r > g) {
            if (r != 0) *ratio = (double)r / (double)b;
            return C_BR;
        }
        if (g != 0) *ratio = (double)g / (double)b;
        return C_BG;
    }
    if (g > b && g > r) {
        if (b > r) {
            if (b != 0) *ratio = (double)b / (double)g;
            return C_GB;
        }
        if (r != 0) *ratio = (double)r / (double)g;
        return C_GR;
    }
    if (r > b && r > g) {
        if (g > b) {
            if (g != 0) *ratio = (double)g / (double)r;
            return C_RG;
        }
        if (b != 0) *ratio = (double)b / (double)r;
        return C_RB;
    }
    return C_NONE;
}

double cdso_pixel_gap(int r1, int g1, int b1, int r2, int g2, int b2) {
    const double BrBg = 0.354862745, BgGb = 0.996078431, GbGr = 0.505882353;
    const double GrRg = 0.996078431, RgRb = 0.505882353;
    double a, b, gap = 10000.0;
    int c1 = hue_class(r1, g1, b1, &a);
    int c2 = hue_class(r2, g2, b2, &b);
    if (c1 == C_NONE || c2 == C_NONE) return gap;
    if (c1 == c2) {
        if (a > 0 && b > 0) gap = (a != b) ? fabs(b - a) : 0.0;
        return gap;
    }
    /* order the pair so that lo is the lower class id; the sums below are commutative in IEEE */
    int lo = c1 < c2 ? c1 : c2, hi = c1 < c2 ? c2 : c1;
    double rlo = c1 < c2 ? a : b, rhi = c1 < c2 ? b : a;
    if (lo == C_BR && hi == C_BG) {
        if (rlo < 0.44 && rhi < 0.54) gap = (rlo - BrBg) + (rhi - BrBg);
    } else if (lo == C_BG && hi == C_GB) {
        if (rlo > 0.8 && rhi > 0.8) gap = (BgGb - rlo) + (BgGb - rhi);
    } else if (lo == C_GB && hi == C_GR) {
        if (rlo < 0.7 && rhi < 0.7) gap = (rlo - GbGr) + (rhi - GbGr);
    } else if (lo == C_GR && hi == C_RG) {
        if (rlo > 0.8 && rhi > 0.8) gap = (GrRg - rlo) + (GrRg - rhi);
    } else if (lo == C_RG && hi == C_RB) {
        if (rlo < 0.7 && rhi < 0.7) gap = (rlo - RgRb) + (rhi - RgRb);
    }
    return gap;
}

/* ------------------------------------------------------------------------------------------
 * getMaskPosArray -- ACDS:96-126.  Row-major linear indices of mask pixels outside the excluded
 * regions with any channel > threshold.  Returns the count; pos may be NULL to count only.
 * bbox[4] (optional) = minx, miny, maxx_exclusive, maxy_exclusive as the reference keeps them.
 * ------------------------------------------------------------------------------------------ */
int cdso_mask_positions(const uint8_t *mask, int W, int H, int thr, const cdso_rect *rects, int nrects,
                        int *pos, int *bbox) {
    int n = 0, minx = W, miny = H, maxx = 0, maxy = 0;
    for (int pi = 0; pi < W * H; pi++) {
        int x = pi % W, y = pi / W;
        if (in_rects(rects, nrects, x, y)) continue;
        int r = mask[3 * pi], g = mask[3 * pi + 1], b = mask[3 * pi + 2];
        if (r > thr || g > thr || b > thr) {
            if (pos) pos[n] = pi;
            n++;
            if (x < minx) minx = x;
            if (x + 1 > maxx) maxx = x + 1;
            if (y < miny) miny = y;
            if (y + 1 > maxy) maxy = y + 1;
        }
    }
    if (bbox) {
        bbox[0] = minx; bbox[1] = miny; bbox[2] = maxx; bbox[3] = maxy;
    }
    return n;
}

/* ------------------------------------------------------------------------------------------
 * Prepared query ("algorithm object"): position lists for every transform, built once per mask
 * exactly as the reference constructor does -- PM:29-57 (generateShiftedMasks PM:113-130,
 * shiftMaskPosArray PM:132-144, mirrorMask PM:146-158).
 * xyShift must be 0 or 2: PM:114-125 overruns its array for >= 4 (ArrayIndexOutOfBounds).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int W, H, P, nshifts, mirror;
    int data_thr;
    double ztol;
    const uint8_t *mask; /* borrowed */
    int *src;            /* P */
    int *tgt[9];         /* nshifts x P */
    int *mtgt[9];        /* nshifts x P when mirror */
} cdso_query;

void cdso_query_free(cdso_query *q) {
    if (!q) return;
    free(q->src);
    for (int i = 0; i < 9; i++) {
        free(q->tgt[i]);
        free(q->mtgt[i]);
    }
    free(q);
}

cdso_query *cdso_query_create(const uint8_t *mask, int W, int H, int mask_thr, int data_thr, double ztol,
                              int xy_shift, int mirror, const cdso_rect *rects, int nrects) {
    if (xy_shift != 0 && xy_shift != 2) return NULL;
    cdso_query *q = (cdso_query *)calloc(1, sizeof(*q));
    q->W = W; q->H = H; q->mask = mask; q->data_thr = data_thr; q->ztol = ztol; q->mirror = mirror;
    q->src = (int *)malloc(sizeof(int) * (size_t)W * H);
    q->P = cdso_mask_positions(mask, W, H, mask_thr, rects, nrects, q->src, NULL);
    int P = q->P;
    q->nshifts = 1 + (xy_shift / 2) * 8;
    if (q->nshifts == 1) {
        q->tgt[0] = (int *)malloc(sizeof(int) * (size_t)(P + 1));
        memcpy(q->tgt[0], q->src, sizeof(int) * (size_t)P);
    } else {
        int id = 0;
        for (int i = 2; i <= xy_shift; i += 2)
            for (int xx = -i; xx <= i; xx += i)
                for (int yy = -i; yy <= i; yy += i) {
                    int *o = (int *)malloc(sizeof(int) * (size_t)(P + 1));
                    for (int k = 0; k < P; k++) {
                        int x = q->src[k] % W + xx, y = q->src[k] / W + yy;
                        o[k] = (x >= 0 && x < W && y >= 0 && y < H) ? y * W + x : -1;
                    }
                    q->tgt[id++] = o;
                }
    }
    if (mirror) {
        for (int s = 0; s < q->nshifts; s++) {
            int *o = (int *)malloc(sizeof(int) * (size_t)(P + 1));
            for (int k = 0; k < P; k++) {
                int p = q->tgt[s][k];
                o[k] = (p == -1) ? -1 : p + (W - 1) - 2 * (p % W);
            }
            q->mtgt[s] = o;
        }
    }
    return q;
}

int cdso_query_size(const cdso_query *q) { return q->P; }

/* calculateScore -- PM:235-263 (the innermost hot loop). */
static int score_one(const cdso_query *q, const uint8_t *target, const int *tpos) {
    int score = 0;
    const uint8_t *m = q->mask;
    for (int i = 0; i < q->P; i++) {
        int sp = q->src[i], tp = tpos[i];
        if (tp == -1 || sp == -1) continue;
        int r2 = target[3 * tp], g2 = target[3 * tp + 1], b2 = target[3 * tp + 2];
        if (r2 > q->data_thr || g2 > q->data_thr || b2 > q->data_thr) {
            double gap = cdso_pixel_gap(m[3 * sp], m[3 * sp + 1], m[3 * sp + 2], r2, g2, b2);
            if (gap <= q->ztol) score++;
        }
    }
    return score;
}

/* calculateMatchingScore -- PM:166-219 with calculateMaxScoreForAllTargetTransformations PM:221-233.
 * counts (optional, 18 ints): per-transform match counts, [0..8] plain shifts in the reference's
 * order, [9..17] mirrored; unused slots are 0. */
void cdso_query_score(const cdso_query *q, const uint8_t *target, int *score, int *mirrored, int *counts) {
    int c[18];
    memset(c, 0, sizeof(c));
    int best = 0, bm = 0;
    *mirrored = 0;
    if (q->P > 0) {
        for (int s = 0; s < q->nshifts; s++) {
            c[s] = score_one(q, target, q->tgt[s]);
            if (c[s] > best) best = c[s];
        }
        if (q->mirror) {
            for (int s = 0; s < q->nshifts; s++) {
                c[9 + s] = score_one(q, target, q->mtgt[s]);
                if (c[9 + s] > bm) bm = c[9 + s];
            }
            if (bm > best) {
                best = bm;
                *mirrored = 1;
            }
        }
    }
    *score = best;
    if (counts) memcpy(counts, c, sizeof(c));
}

/* A library scan on the CPU with the reference's own parallel structure: targets are cut into
 * partitions and handed to a pool of worker threads, every worker sharing the one immutable query
 * (tools cmd/cdsprocess/LocalColorMIPSearchProcessor.java:88-107).  Used for the cpu_baseline. */
typedef struct {
    const cdso_query *q;
    const uint8_t *lib;
    size_t stride;
    long n;
    int *scores;
    uint8_t *mirrored;
    long part;
    long next;
    pthread_mutex_t mu;
} scan_job;

static void *scan_worker(void *arg) {
    scan_job *j = (scan_job *)arg;
    for (;;) {
        pthread_mutex_lock(&j->mu);
        long a = j->next;
        j->next += j->part;
        pthread_mutex_unlock(&j->mu);
        if (a >= j->n) break;
        long b = a + j->part < j->n ? a + j->part : j->n;
        for (long t = a; t < b; t++) {
            int s, m;
            cdso_query_score(j->q, j->lib + (size_t)t * j->stride, &s, &m, NULL);
            j->scores[t] = s;
            j->mirrored[t] = (uint8_t)m;
        }
    }
    return NULL;
}

void cdso_query_scan(const cdso_query *q, const uint8_t *lib, long n_targets, size_t stride_bytes,
                     int *scores, uint8_t *mirrored, int n_threads, int partition) {
    scan_job j;
    j.q = q; j.lib = lib; j.stride = stride_bytes; j.n = n_targets; j.scores = scores; j.mirrored = mirrored;
    j.part = partition > 0 ? partition : 100;
    j.next = 0;
    pthread_mutex_init(&j.mu, NULL);
    if (n_threads <= 1) {
        scan_worker(&j);
    } else {
        pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)n_threads);
        for (int i = 0; i < n_threads; i++) pthread_create(&th[i], NULL, scan_worker, &j);
        for (int i = 0; i < n_threads; i++) pthread_join(th[i], NULL);
        free(th);
    }
    pthread_mutex_destroy(&j.mu);
}

/* PixelMatchScore.getNormalizedScore (cds/PixelMatchScore.java:21-24) and ColorMIPSearch.isMatch
 * (cds/ColorMIPSearch.java:42-45): ratio = score/(double)P, reported as float; match iff
 * score > 0 && (float)ratio > pctPositivePixels/100 (float promoted to double for the compare). */
float cdso_normalized_pixel_score(int score, int P) {
    if (P == 0) return 0.0f;
    return (float)((double)score / (double)P);
}
int cdso_is_match(int score, int P, double pct_positive_pixels) {
    double thr = pct_positive_pixels / 100;
    return score > 0 && (double)cdso_normalized_pixel_score(score, P) > thr;
}

/* ==========================================================================================
 *                              GradientAreaGap ("shape") scorer
 * ========================================================================================== */

/* The 256-entry colour-depth LUT -- data of GAG:132-155. */
static const uint8_t DEPTH_LUT[256][3] = {
    {127, 0, 255}, {125, 3, 255}, {124, 6, 255}, {122, 9, 255}, {121, 12, 255}, {120, 15, 255}, {119, 18, 255}, {118, 21, 255},
    {116, 24, 255}, {115, 27, 255}, {114, 30, 255}, {113, 33, 255}, {112, 36, 255}, {110, 39, 255}, {109, 42, 255}, {108, 45, 255},
    {106, 48, 255}, {105, 51, 255}, {104, 54, 255}, {103, 57, 255}, {101, 60, 255}, {100, 63, 255}, {99, 66, 255}, {98, 69, 255},
    {96, 72, 255}, {95, 75, 255}, {94, 78, 255}, {93, 81, 255}, {92, 84, 255}, {90, 87, 255}, {89, 90, 255}, {87, 93, 255},
    {86, 96, 255}, {84, 99, 255}, {83, 102, 255}, {81, 105, 255}, {80, 108, 255}, {78, 111, 255}, {77, 114, 255}, {75, 117, 255},
    {74, 120, 255}, {72, 123, 255}, {71, 126, 255}, {69, 129, 255}, {68, 132, 255}, {66, 135, 255}, {65, 138, 255}, {63, 141, 255},
    {62, 144, 255}, {60, 147, 255}, {59, 150, 255}, {57, 153, 255}, {56, 156, 255}, {54, 159, 255}, {53, 162, 255}, {51, 165, 255},
    {50, 168, 255}, {48, 171, 255}, {47, 174, 255}, {45, 177, 255}, {44, 180, 255}, {42, 183, 255}, {41, 186, 255}, {39, 189, 255},
    {38, 192, 255}, {36, 195, 255}, {35, 198, 255}, {33, 201, 255}, {32, 204, 255}, {30, 207, 255}, {29, 210, 255}, {27, 213, 255},
    {26, 216, 255}, {24, 219, 255}, {23, 222, 255}, {21, 225, 255}, {20, 228, 255}, {18, 231, 255}, {16, 234, 255}, {14, 237, 255},
    {12, 240, 255}, {9, 243, 255}, {6, 246, 255}, {3, 249, 255}, {1, 252, 255}, {0, 254, 255}, {3, 255, 252}, {6, 255, 249},
    {9, 255, 246}, {12, 255, 243}, {15, 255, 240}, {18, 255, 237}, {21, 255, 234}, {24, 255, 231}, {27, 255, 228}, {30, 255, 225},
    {33, 255, 222}, {36, 255, 219}, {39, 255, 216}, {42, 255, 213}, {45, 255, 210}, {48, 255, 207}, {51, 255, 204}, {54, 255, 201},
    {57, 255, 198}, {60, 255, 195}, {63, 255, 192}, {66, 255, 189}, {69, 255, 186}, {72, 255, 183}, {75, 255, 180}, {78, 255, 177},
    {81, 255, 174}, {84, 255, 171}, {87, 255, 168}, {90, 255, 165}, {93, 255, 162}, {96, 255, 159}, {99, 255, 156}, {102, 255, 153},
    {105, 255, 150}, {108, 255, 147}, {111, 255, 144}, {114, 255, 141}, {117, 255, 138}, {120, 255, 135}, {123, 255, 132}, {126, 255, 129},
    {129, 255, 126}, {132, 255, 123}, {135, 255, 120}, {138, 255, 117}, {141, 255, 114}, {144, 255, 111}, {147, 255, 108}, {150, 255, 105},
    {153, 255, 102}, {156, 255, 99}, {159, 255, 96}, {162, 255, 93}, {165, 255, 90}, {168, 255, 87}, {171, 255, 84}, {174, 255, 81},
    {177, 255, 78}, {180, 255, 75}, {183, 255, 72}, {186, 255, 69}, {189, 255, 66}, {192, 255, 63}, {195, 255, 60}, {198, 255, 57},
    {201, 255, 54}, {204, 255, 51}, {207, 255, 48}, {210, 255, 45}, {213, 255, 42}, {216, 255, 39}, {219, 255, 36}, {222, 255, 33},
    {225, 255, 30}, {228, 255, 27}, {231, 255, 24}, {234, 255, 21}, {237, 255, 18}, {240, 255, 15}, {243, 255, 12}, {246, 255, 9},
    {249, 255, 6}, {252, 255, 3}, {254, 255, 0}, {255, 252, 3}, {255, 249, 6}, {255, 246, 9}, {255, 243, 12}, {255, 240, 15},
    {255, 237, 18}, {255, 234, 21}, {255, 231, 24}, {255, 228, 27}, {255, 225, 30}, {255, 222, 33}, {255, 219, 36}, {255, 216, 39},
    {255, 213, 42}, {255, 210, 45}, {255, 207, 48}, {255, 204, 51}, {255, 201, 54}, {255, 198, 57}, {255, 195, 60}, {255, 192, 63},
    {255, 189, 66}, {255, 186, 69}, {255, 183, 72}, {255, 180, 75}, {255, 177, 78}, {255, 174, 81}, {255, 171, 84}, {255, 168, 87},
    {255, 165, 90}, {255, 162, 93}, {255, 159, 96}, {255, 156, 99}, {255, 153, 102}, {255, 150, 105}, {255, 147, 108}, {255, 144, 111},
    {255, 141, 114}, {255, 138, 117}, {255, 135, 120}, {255, 132, 123}, {255, 129, 126}, {255, 126, 129}, {255, 123, 132}, {255, 120, 135},
    {255, 117, 138}, {255, 114, 141}, {255, 111, 144}, {255, 108, 147}, {255, 105, 150}, {255, 102, 153}, {255, 99, 156}, {255, 96, 159},
    {255, 93, 162}, {255, 90, 165}, {255, 87, 168}, {255, 84, 171}, {255, 81, 173}, {255, 78, 174}, {255, 75, 175}, {255, 72, 176},
    {255, 69, 177}, {255, 66, 178}, {255, 63, 179}, {255, 60, 180}, {255, 57, 181}, {255, 54, 182}, {255, 51, 183}, {255, 48, 184},
    {255, 45, 185}, {255, 42, 186}, {255, 39, 187}, {255, 36, 188}, {255, 33, 189}, {255, 30, 190}, {255, 27, 191}, {255, 24, 192},
    {255, 21, 193}, {255, 18, 194}, {255, 15, 195}, {255, 12, 196}, {255, 9, 197}, {255, 6, 198}, {255, 3, 199}, {255, 0, 200}};

const uint8_t *cdso_depth_lut(void) { return &DEPTH_LUT[0][0]; }

/* findSliceNumberInLUT -- GAG:131-197: exact ratio hit, else FIRST minimum of |ratio - lutRatio|. */
static int slice_in_lut(int lo, int hi, double ratio) {
    int slice = 0;
    double mingap = 1000;
    for (int i = lo; i <= hi; i++) {
        double R = DEPTH_LUT[i][0], G = DEPTH_LUT[i][1], B = DEPTH_LUT[i][2], lr = 0;
        if (B > R && B > G) {
            if (R > G) lr = R / B;
            else if (G > R) lr = G / B;
        } else if (G > R && G > B) {
            if (R > B) lr = R / G;
            else if (B > R) lr = B / G;
        } else if (R > G && R > B) {
            if (G > B) lr = G / R;
            else if (B > G) lr = B / R;
        }
        if (lr == ratio) return i + 1;
        double gp = fabs(ratio - lr);
        if (gp < mingap) {
            mingap = gp;
            slice = i + 1;
        }
    }
    return slice;
}

/* Slice number of one colour: the per-pixel half of calculateSliceGap (GAG:18-105) +
 * findSliceNumber (GAG:107-129).  Channel order of the ">=" tests is R, G, B.  Black -> NaN
 * ratio -> 0 (every comparison with NaN is false). */
int cdso_slice_number(int r, int g, int b) {
    int first, second, c1, c2; /* c: 0=R 1=G 2=B */
    if (r >= g && r >= b) {
        first = r; c1 = 0;
        if (g >= b) { second = g; c2 = 1; } else { second = b; c2 = 2; }
    } else if (g >= r && g >= b) {
        first = g; c1 = 1;
        if (r >= b) { second = r; c2 = 0; } else { second = b; c2 = 2; }
    } else {
        first = b; c1 = 2;
        if (r >= g) { second = r; c2 = 0; } else { second = g; c2 = 1; }
    }
    double ratio = (double)second / (double)first; /* 0/0 = NaN for black, as in Java */
    if (c1 == 0) return c2 == 1 ? slice_in_lut(171, 212, ratio) : slice_in_lut(213, 255, ratio);
    if (c1 == 1) return c2 == 0 ? slice_in_lut(128, 170, ratio) : slice_in_lut(86, 127, ratio);
    return c2 == 0 ? slice_in_lut(0, 29, ratio) : slice_in_lut(30, 85, ratio);
}

/* calculateSliceGap -- GAG:97-104. */
int cdso_slice_gap(int r1, int g1, int b1, int r2, int g2, int b2) {
    int m = cdso_slice_number(r1, g1, b1), d = cdso_slice_number(r2, g2, b2);
    if (d == 0 || m == 0) return d;
    return abs(m - d);
}

/* makeLineRadii -- IT:549-572.  Returns kRadius; hw[dy + kRadius] = half width of kernel row dy. */
static int line_radii(double radius_arg, int *hw, int cap) {
    double radius = radius_arg;
    if (radius_arg >= 1.5 && radius_arg < 1.75) radius = 1.75;
    else if (radius_arg >= 2.5 && radius_arg < 2.85) radius = 2.85;
    int r2 = (int)(radius * radius) + 1;
    int k = (int)(sqrt(r2 + 1e-10));
    if (2 * k + 1 > cap) return -1;
    hw[k] = k;
    for (int y = 1; y <= k; y++) {
        int dx = (int)(sqrt(r2 - y * y + 1e-10));
        hw[k - y] = dx;
        hw[k + y] = dx;
    }
    return k;
}

/* Circular max filter of a packed RGB image, per channel, pixels outside the image ignored --
 * the result of unsafeMaxFilterWithHistogram (IT:353-535) when the image is traversed row by row
 * from (0,0) with no border (how LImage.fold/toImageArray drive it, LImage.java:89-97).
 * Computed by scattering every non-black source pixel over its (symmetric) kernel. */
void cdso_max_filter_rgb(const uint8_t *in, int W, int H, double radius, uint8_t *out) {
    int hw[1024];
    int k = line_radii(radius, hw, 1024);
    memset(out, 0, (size_t)3 * W * H);
    if (k < 0) return;
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const uint8_t *p = in + 3 * ((size_t)y * W + x);
            if (!(p[0] | p[1] | p[2])) continue;
            for (int dy = -k; dy <= k; dy++) {
                int yy = y + dy;
                if (yy < 0 || yy >= H) continue;
                int h = hw[dy + k];
                int xa = x - h < 0 ? 0 : x - h, xb = x + h >= W ? W - 1 : x + h;
                uint8_t *o = out + 3 * ((size_t)yy * W + xa);
                for (int xx = xa; xx <= xb; xx++, o += 3) {
                    if (p[0] > o[0]) o[0] = p[0];
                    if (p[1] > o[1]) o[1] = p[1];
                    if (p[2] > o[2]) o[2] = p[2];
                }
            }
        }
}

/* rgbToGrayNoGammaCorrection(pv, 255) -- CT:40-54. */
static int gray_nogamma(int r, int g, int b) {
    if ((r | g | b) == 0) return 0;
    double rw = 1 / 3., gw = 1 / 3., bw = 1 / 3.;
    float maxGray = 255;
    return (int)((maxGray / 255) * (r * rw + g * gw + b * bw + 0.5));
}

/* ColorTransformation.mask(threshold) on RGB -- CT:29-38,114-131: black if all channels <= thr. */
void cdso_mask_rgb(const uint8_t *in, long npix, int thr, uint8_t *out) {
    for (long i = 0; i < npix; i++) {
        int r = in[3 * i], g = in[3 * i + 1], b = in[3 * i + 2];
        int keep = !(r <= thr && g <= thr && b <= thr);
        out[3 * i] = keep ? (uint8_t)r : 0;
        out[3 * i + 1] = keep ? (uint8_t)g : 0;
        out[3 * i + 2] = keep ? (uint8_t)b : 0;
    }
}

/* clearRegion -- IT:182-193. */
void cdso_clear_regions(const uint8_t *in, int W, int H, const cdso_rect *rects, int nrects, uint8_t *out) {
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            size_t i = 3 * ((size_t)y * W + x);
            int ex = in_rects(rects, nrects, x, y);
            out[i] = ex ? 0 : in[i];
            out[i + 1] = ex ? 0 : in[i + 1];
            out[i + 2] = ex ? 0 : in[i + 2];
        }
}

/* The zgap image the reference's tests synthesise when no ZGapImage file exists
 * (Shape2D...Test.java:171-174): clearRegion -> mask(thr) -> unsafeMaxFilter(radius). */
void cdso_make_zgap(const uint8_t *target, int W, int H, int thr, double radius, const cdso_rect *rects,
                    int nrects, uint8_t *out) {
    uint8_t *tmp = (uint8_t *)malloc((size_t)3 * W * H);
    cdso_clear_regions(target, W, H, rects, nrects, tmp);
    cdso_mask_rgb(tmp, (long)W * H, thr, tmp);
    cdso_max_filter_rgb(tmp, W, H, radius, out);
    free(tmp);
}

/* Mask preparation for the shape scorer -- FAC:91-125 (border size 0):
 *   Q  = mask with excluded regions black;
 *   QM = gray(Q) > 2;
 *   QH = maxfilter20(Q) black && gray(maxfilter60(Q)) > 0.
 * Outputs: Q (3*W*H), QM (W*H, 0/1), QH (W*H, 0/1). */
void cdso_shape_prepare(const uint8_t *mask, int W, int H, const cdso_rect *rects, int nrects, uint8_t *Q, uint8_t *QM, uint8_t *QH);
/* queryBorderSize (FAC:103: LImageUtils.create(queryImageArray, b, b, b, b)): the lazy query image keeps its pixels, but
 * every fold (LImage.java:89-113) skips a frame of `border` pixels -- the reduce()d masks QM and QH (FAC:104-108, toImageArray
 * is a foldi) are therefore 0 in the frame although the max filters still SEE the frame's pixels, and the gap sum
 * (combine4 inherits the query image's borders, LImageUtils.java:44-49) runs over the inner rectangle only. */
void cdso_shape_prepare_b(const uint8_t *mask, int W, int H, const cdso_rect *rects, int nrects, int border, uint8_t *Q,
                          uint8_t *QM, uint8_t *QH) {
    cdso_shape_prepare(mask, W, H, rects, nrects, Q, QM, QH);
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++)
            if (x < border || x >= W - border || y < border || y >= H - border) QM[(size_t)y * W + x] = QH[(size_t)y * W + x] = 0;
}

void cdso_shape_prepare(const uint8_t *mask, int W, int H, const cdso_rect *rects, int nrects, uint8_t *Q,
                        uint8_t *QM, uint8_t *QH) {
    size_t N = (size_t)W * H;
    cdso_clear_regions(mask, W, H, rects, nrects, Q);
    uint8_t *d60 = (uint8_t *)malloc(3 * N), *d20 = (uint8_t *)malloc(3 * N);
    cdso_max_filter_rgb(Q, W, H, 60, d60);
    cdso_max_filter_rgb(Q, W, H, 20, d20);
    for (size_t i = 0; i < N; i++) {
        QM[i] = gray_nogamma(Q[3 * i], Q[3 * i + 1], Q[3 * i + 2]) > 2;
        int p2black = (d20[3 * i] | d20[3 * i + 1] | d20[3 * i + 2]) == 0;
        QH[i] = p2black ? (gray_nogamma(d60[3 * i], d60[3 * i + 1], d60[3 * i + 2]) > 0) : 0;
    }
    free(d60);
    free(d20);
}

/* calculateNegativeScores -- S2D:196-245 with PIXEL_GAP_OP S2D:28-42, for one orientation.
 * roi may be NULL; when given it is a W*H byte plane (non-zero = inside), already cleared in the
 * excluded regions (FAC:96-100); it is applied AFTER mirroring and is itself never mirrored
 * (S2D:206-217).  zgap must already be masked by the threshold (S2D:161). */
static void negative_scores(const uint8_t *Q, const uint8_t *QM, const uint8_t *QH, const uint8_t *roi,
                            const uint8_t *tclr, const uint16_t *grad, const uint8_t *zm, int W, int H, int thr,
                            int mirrored, int border, int64_t *gap_out, int64_t *high_out) {
    int64_t gaps = 0, high = 0;
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const int inner = x >= border && x < W - border && y >= border && y < H - border;  /* the gap fold's range */
            int xo = mirrored ? W - 1 - x : x;
            size_t i = (size_t)y * W + x, io = (size_t)y * W + xo;
            int inroi = roi ? roi[i] != 0 : 1;
            int qr = inroi ? Q[3 * io] : 0, qg = inroi ? Q[3 * io + 1] : 0, qb = inroi ? Q[3 * io + 2] : 0;
            int qm = inroi ? QM[io] : 0, qh = inroi ? QH[io] : 0;
            int zr = zm[3 * io], zg = zm[3 * io + 1], zb = zm[3 * io + 2];
            int g = grad[i];
            int gap;
            if ((qr | qg | qb) != 0 && (zr | zg | zb) != 0) {
                int sg = cdso_slice_gap(qr, qg, qb, zr, zg, zb);
                gap = (40 <= sg - 40) ? sg - 40 : qm * g;
            } else {
                gap = qm * g;
            }
            if (inner) gaps += gap > 3 ? gap : 0;
            if (qh == 1) {
                int r1 = tclr[3 * i], g1 = tclr[3 * i + 1], b1 = tclr[3 * i + 2];
                if (r1 > thr || g1 > thr || b1 > thr) high++;
            }
        }
    *gap_out = gaps;
    *high_out = high;
}

/* calculate2DShapeScore -- GAG:199-207; ShapeMatchScore.getScore (cds/ShapeMatchScore.java:29-33). */
int64_t cdso_shape_score_value(int64_t gap, int64_t high) {
    if (gap >= 0 && high >= 0) return gap + high / 3;
    return -1;
}

/* calculateMatchingScore -- S2D:150-186.  target: RGB; grad: u16 (a GRAY8 gradient is widened by
 * the caller); zgap: RGB, NOT yet masked; either NULL => (-1,-1,false) (S2D:155-158).
 * out[0]=gradientAreaGap out[1]=highExpressionArea out[2]=mirrored(0/1). */
void cdso_shape_score_b(const uint8_t *Q, const uint8_t *QM, const uint8_t *QH, const uint8_t *roi,
                        const uint8_t *target, const uint16_t *grad, const uint8_t *zgap, int W, int H, int thr,
                        int mirror, const cdso_rect *rects, int nrects, int border, int64_t *out);
void cdso_shape_score(const uint8_t *Q, const uint8_t *QM, const uint8_t *QH, const uint8_t *roi,
                      const uint8_t *target, const uint16_t *grad, const uint8_t *zgap, int W, int H, int thr,
                      int mirror, const cdso_rect *rects, int nrects, int64_t *out) {
    cdso_shape_score_b(Q, QM, QH, roi, target, grad, zgap, W, H, thr, mirror, rects, nrects, 0, out);
}

void cdso_shape_score_b(const uint8_t *Q, const uint8_t *QM, const uint8_t *QH, const uint8_t *roi,
                        const uint8_t *target, const uint16_t *grad, const uint8_t *zgap, int W, int H, int thr,
                        int mirror, const cdso_rect *rects, int nrects, int border, int64_t *out) {
    if (!grad || !zgap) {
        out[0] = -1; out[1] = -1; out[2] = 0;
        return;
    }
    size_t N = (size_t)W * H;
    uint8_t *tclr = (uint8_t *)malloc(3 * N), *zm = (uint8_t *)malloc(3 * N);
    cdso_clear_regions(target, W, H, rects, nrects, tclr);
    cdso_mask_rgb(zgap, (long)N, thr, zm);
    int64_t g0, h0;
    negative_scores(Q, QM, QH, roi, tclr, grad, zm, W, H, thr, 0, border, &g0, &h0);
    out[0] = g0; out[1] = h0; out[2] = 0;
    if (mirror) {
        int64_t g1, h1;
        negative_scores(Q, QM, QH, roi, tclr, grad, zm, W, H, thr, 1, border, &g1, &h1);
        if ((int)cdso_shape_score_value(g1, h1) < (int)cdso_shape_score_value(g0, h0)) {
            out[0] = g1; out[1] = h1; out[2] = 1;
        }
    }
    free(tclr);
    free(zm);
}

/* calculateNormalizedScore -- GAG:219-235. */
double cdso_normalized_score(int pix, int64_t shape, int64_t max_pix, int64_t max_shape) {
    if (pix == 0 || max_pix == 0 || shape < 0 || max_shape <= 0) return pix;
    double np = (double)pix / max_pix;
    double ns = (double)shape / max_shape;
    double bounded = fmin(fmax(ns * 2.5, 0.002), 1.);
    return np / bounded * 100;
}

/* Test helper: out[i*m + j] = 1 iff target colour j passes the data threshold and
 * calculatePixelGap(mask colour i, target colour j) <= ztol  (the body of PM:251-258). */
void cdso_match_matrix(const uint8_t *mask_rgb, int n, const uint8_t *target_rgb, int m, int data_thr, double ztol,
                       uint8_t *out) {
    for (int i = 0; i < n; i++)
        for (int j = 0; j < m; j++) {
            int r2 = target_rgb[3 * j], g2 = target_rgb[3 * j + 1], b2 = target_rgb[3 * j + 2];
            int ok = 0;
            if (r2 > data_thr || g2 > data_thr || b2 > data_thr)
                ok = cdso_pixel_gap(mask_rgb[3 * i], mask_rgb[3 * i + 1], mask_rgb[3 * i + 2], r2, g2, b2) <= ztol;
            out[(size_t)i * m + j] = (uint8_t)ok;
        }
}
