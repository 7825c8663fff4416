/*
 * cds_hip_jni.c -- JNI implementations of org.janelia.colormipsearch.cds.gpu.CdsHip (one function per native method),
 * over the C ABI of libcds_hip.so (include/cds_hip.h).  Build: see ../../../Makefile (needs a JDK for jni.h; the image this
 * repository is developed in has none, so no JDK has compiled this file -- tests/test_java_sources.py parses it with
 * `gcc -fsyntax-only -Wall -Werror` against tests/jni_stub/jni.h, a minimal declaration set that is test scaffolding only).
 *
 * Error mapping: CDS_ERR_INVALID_ARGUMENT -> IllegalArgumentException (where the reference throws it),
 * everything else -> IllegalStateException; the message is cds_last_error().
 * Every C-ABI call here may block on the GPU (stream syncs, hipMalloc, host threads), and the per-pair calls come from many
 * Java threads at once (LocalColorMIPSearchProcessor.java:88-105): arrays are therefore accessed with Get/Release<Type>ArrayElements
 * (or direct buffers), never with GetPrimitiveArrayCritical, which would hold the GC lock across those waits.
 * Every array length is checked against the dimensions BEFORE the native call: a short Java array must become an
 * IllegalArgumentException, not an out-of-bounds access on the Java heap.
 */
#include <jni.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "cds_hip.h"

#define CLS(name) Java_org_janelia_colormipsearch_cds_gpu_CdsHip_##name
#define P(type, h) ((type *)(intptr_t)(h))

static void throw_for(JNIEnv *env, int rc) {
    const char *cls = rc == CDS_ERR_INVALID_ARGUMENT ? "java/lang/IllegalArgumentException" : "java/lang/IllegalStateException";
    (*env)->ThrowNew(env, (*env)->FindClass(env, cls), cds_last_error());
}
static void throw_iae(JNIEnv *env, const char *msg) {
    (*env)->ThrowNew(env, (*env)->FindClass(env, "java/lang/IllegalArgumentException"), msg);
}
/* array present and at least `need` elements long?  (need < 0: an overflowed product)  Throws and returns 0 otherwise. */
static int has_len(JNIEnv *env, jarray a, int64_t need, const char *what) {
    if (!a || need < 0 || (int64_t)(*env)->GetArrayLength(env, a) < need) {
        throw_iae(env, what);
        return 0;
    }
    return 1;
}
/* direct buffer present and at least `need` bytes long? */
static const void *direct_bytes(JNIEnv *env, jobject buf, int64_t need, const char *what) {
    void *p = buf ? (*env)->GetDirectBufferAddress(env, buf) : NULL;
    if (!p || need < 0 || (int64_t)(*env)->GetDirectBufferCapacity(env, buf) < need) {
        throw_iae(env, what);
        return NULL;
    }
    return p;
}
static int64_t image_bytes(jint w, jint h) { return w > 0 && h > 0 ? (int64_t)3 * w * h : -1; }
/* n handles (jlong) -> a malloc'ed pointer array; NULL (exception pending) when the array is missing or not `n` long */
static void **handle_array(JNIEnv *env, jlongArray handles, jsize n, const char *what) {
    if (!handles || (*env)->GetArrayLength(env, handles) != n) {
        throw_iae(env, what);
        return NULL;
    }
    void **out = (void **)calloc((size_t)n + 1, sizeof(void *));
    jlong *h = (*env)->GetLongArrayElements(env, handles, NULL);
    for (jsize i = 0; i < n; i++) out[i] = (void *)(intptr_t)h[i];
    (*env)->ReleaseLongArrayElements(env, handles, h, JNI_ABORT);
    return out;
}

/* x0,y0,x1,y1 per rectangle -> cds_rect[]; returns the count or -1 (exception pending) */
static int get_rects(JNIEnv *env, jintArray rects, cds_rect *out) {
    jsize n = rects ? (*env)->GetArrayLength(env, rects) / 4 : 0;
    if (n > CDS_MAX_RECTS) {
        (*env)->ThrowNew(env, (*env)->FindClass(env, "java/lang/IllegalArgumentException"), "more than 8 excluded rectangles");
        return -1;
    }
    if (n) (*env)->GetIntArrayRegion(env, rects, 0, 4 * n, (jint *)out);
    return (int)n;
}

/* ---- context ---- */
JNIEXPORT jlong JNICALL CLS(contextCreate)(JNIEnv *env, jclass c, jint device) {
    cds_context *ctx = NULL;
    int rc = cds_context_create(device, &ctx);
    if (rc) { throw_for(env, rc); return 0; }
    return (jlong)(intptr_t)ctx;
}
JNIEXPORT void JNICALL CLS(contextDestroy)(JNIEnv *env, jclass c, jlong ctx) { cds_context_destroy(P(cds_context, ctx)); }
JNIEXPORT void JNICALL CLS(contextSynchronize)(JNIEnv *env, jclass c, jlong ctx) {
    int rc = cds_context_synchronize(P(cds_context, ctx));
    if (rc) throw_for(env, rc);
}
JNIEXPORT jlong JNICALL CLS(contextTrim)(JNIEnv *env, jclass c, jlong ctx) {
    uint64_t freed = 0;
    int rc = cds_context_trim(P(cds_context, ctx), &freed);
    if (rc) { throw_for(env, rc); return 0; }
    return (jlong)freed;
}
JNIEXPORT void JNICALL CLS(contextSetCacheLimit)(JNIEnv *env, jclass c, jlong ctx, jlong bytes) {
    int rc = cds_context_set_cache_limit(P(cds_context, ctx), bytes < 0 ? 0 : (uint64_t)bytes);
    if (rc) throw_for(env, rc);
}

/* ---- pixel-match mask sets ---- */
JNIEXPORT jlong JNICALL CLS(masksetCreate)(JNIEnv *env, jclass c, jlong ctx, jint nMasks, jbyteArray masksRgb, jint w, jint h, jint maskThr,
                                           jint dataThr, jdouble zTol, jint xyShift, jboolean mirror, jintArray rects) {
    cds_pixmatch_params p = {maskThr, dataThr, zTol, xyShift, mirror ? 1 : 0};
    cds_rect r[CDS_MAX_RECTS];
    int n = get_rects(env, rects, r);
    if (n < 0) return 0;
    if (nMasks < 1 || !has_len(env, masksRgb, image_bytes(w, h) < 0 ? -1 : nMasks * image_bytes(w, h), "masksRgb is shorter than nMasks * 3 * w * h")) return 0;
    jbyte *px = (*env)->GetByteArrayElements(env, masksRgb, NULL);
    cds_maskset *ms = NULL;
    int rc = cds_maskset_create(P(cds_context, ctx), nMasks, (const uint8_t *)px, w, h, &p, r, n, &ms);
    (*env)->ReleaseByteArrayElements(env, masksRgb, px, JNI_ABORT);
    if (rc) { throw_for(env, rc); return 0; }
    return (jlong)(intptr_t)ms;
}
JNIEXPORT void JNICALL CLS(masksetDestroy)(JNIEnv *env, jclass c, jlong ms) { cds_maskset_destroy(P(cds_maskset, ms)); }
JNIEXPORT void JNICALL CLS(masksetQuerySizes)(JNIEnv *env, jclass c, jlong ms, jintArray sizes) {
    if (!has_len(env, sizes, cds_maskset_size(P(cds_maskset, ms)), "sizes is shorter than the mask set")) return;
    jint *s = (*env)->GetIntArrayElements(env, sizes, NULL);
    int rc = cds_maskset_query_sizes(P(cds_maskset, ms), (int32_t *)s);
    (*env)->ReleaseIntArrayElements(env, sizes, s, 0);
    if (rc) throw_for(env, rc);
}
JNIEXPORT jlong JNICALL CLS(scorePair)(JNIEnv *env, jclass c, jlong ms, jint maskIndex, jbyteArray targetRgb, jint w, jint h) {
    int32_t score = 0;
    uint8_t mirrored = 0;
    if (!has_len(env, targetRgb, image_bytes(w, h), "targetRgb is shorter than 3 * w * h")) return 0;
    jbyte *px = (*env)->GetByteArrayElements(env, targetRgb, NULL);
    int rc = cds_score_pair(P(cds_maskset, ms), maskIndex, (const uint8_t *)px, w, h, &score, &mirrored);
    (*env)->ReleaseByteArrayElements(env, targetRgb, px, JNI_ABORT);
    if (rc) { throw_for(env, rc); return 0; }
    return ((jlong)mirrored << 32) | (uint32_t)score;
}

/* ---- dense resident library ---- */
JNIEXPORT jlong JNICALL CLS(libraryCreate)(JNIEnv *env, jclass c, jlong ctx, jint w, jint h, jlong capacity) {
    cds_library *lib = NULL;
    int rc = cds_library_create(P(cds_context, ctx), w, h, capacity, &lib);
    if (rc) { throw_for(env, rc); return 0; }
    return (jlong)(intptr_t)lib;
}
JNIEXPORT void JNICALL CLS(libraryDestroy)(JNIEnv *env, jclass c, jlong lib) { cds_library_destroy(P(cds_library, lib)); }
/* bytes of `count` images of a library's size: 3 * W * H each (RGB) or 2 * W * H (gradient); -1 for a bad count / handle */
static int64_t library_bytes(jlong lib, jlong count, int per_pixel) {
    int32_t w = 0, h = 0;
    if (count < 0 || cds_library_dims(P(cds_library, lib), &w, &h, NULL) != CDS_OK) return -1;
    return (int64_t)per_pixel * w * h * count;
}
JNIEXPORT void JNICALL CLS(libraryUpload)(JNIEnv *env, jclass c, jlong lib, jlong first, jlong count, jobject directRgb) {
    const uint8_t *px = (const uint8_t *)direct_bytes(env, directRgb, library_bytes(lib, count, 3), "directRgb is not a direct buffer of count * 3 * w * h bytes");
    if (!px) return;
    int rc = cds_library_upload(P(cds_library, lib), first, count, px);
    if (rc) throw_for(env, rc);
}
JNIEXPORT void JNICALL CLS(libraryUploadVariants)(JNIEnv *env, jclass c, jlong lib, jlong first, jlong count, jobject directGrad, jobject directZgap) {
    const uint16_t *g = NULL;
    const uint8_t *z = NULL;
    if (directGrad && !(g = (const uint16_t *)direct_bytes(env, directGrad, library_bytes(lib, count, 2), "directGrad is not a direct buffer of count * 2 * w * h bytes"))) return;
    if (directZgap && !(z = (const uint8_t *)direct_bytes(env, directZgap, library_bytes(lib, count, 3), "directZgapRgb is not a direct buffer of count * 3 * w * h bytes"))) return;
    int rc = cds_library_upload_variants(P(cds_library, lib), first, count, g, z);
    if (rc) throw_for(env, rc);
}
JNIEXPORT void JNICALL CLS(libraryUploadGradientU8)(JNIEnv *env, jclass c, jlong lib, jlong first, jlong count, jobject directGrad8) {
    const uint8_t *g = (const uint8_t *)direct_bytes(env, directGrad8, library_bytes(lib, count, 1), "directGrad8 is not a direct buffer of count * w * h bytes");
    if (!g) return;
    int rc = cds_library_upload_gradient_u8(P(cds_library, lib), first, count, g);
    if (rc) throw_for(env, rc);
}
JNIEXPORT void JNICALL CLS(libraryMakeZgapVariants)(JNIEnv *env, jclass c, jlong lib, jlong first, jlong count, jint threshold, jdouble radius,
                                                    jintArray rects) {
    cds_rect r[CDS_MAX_RECTS];
    int n = get_rects(env, rects, r);
    if (n < 0) return;
    int rc = cds_library_make_zgap_variants(P(cds_library, lib), first, count, threshold, radius, r, n);
    if (rc) throw_for(env, rc);
}
static void search_common(JNIEnv *env, int sparse, jlong ms, jlong lib, jlong first, jlong count, jintArray scores, jbyteArray mirrored) {
    const int64_t need = count < 0 ? -1 : (int64_t)cds_maskset_size(P(cds_maskset, ms)) * count;
    if (!has_len(env, scores, need, "scores is shorter than nMasks * count") || !has_len(env, mirrored, need, "mirrored is shorter than nMasks * count")) return;
    jint *s = (*env)->GetIntArrayElements(env, scores, NULL);
    jbyte *m = (*env)->GetByteArrayElements(env, mirrored, NULL);
    int rc = sparse ? cds_search_sparse(P(cds_maskset, ms), P(cds_sparse_library, lib), first, count, (int32_t *)s, (uint8_t *)m)
                    : cds_search(P(cds_maskset, ms), P(cds_library, lib), first, count, (int32_t *)s, (uint8_t *)m);
    (*env)->ReleaseIntArrayElements(env, scores, s, 0);
    (*env)->ReleaseByteArrayElements(env, mirrored, m, 0);
    if (rc) throw_for(env, rc);
}
JNIEXPORT void JNICALL CLS(search)(JNIEnv *env, jclass c, jlong ms, jlong lib, jlong first, jlong count, jintArray scores, jbyteArray mirrored) {
    search_common(env, 0, ms, lib, first, count, scores, mirrored);
}

/* ---- sparse resident library ---- */
JNIEXPORT jlong JNICALL CLS(sparseLibraryCreate)(JNIEnv *env, jclass c, jlong ctx, jint w, jint h, jlong capImages, jlong capPixels) {
    cds_sparse_library *sl = NULL;
    int rc = cds_sparse_library_create(P(cds_context, ctx), w, h, capImages, capPixels, &sl);
    if (rc) { throw_for(env, rc); return 0; }
    return (jlong)(intptr_t)sl;
}
JNIEXPORT void JNICALL CLS(sparseLibraryDestroy)(JNIEnv *env, jclass c, jlong sl) { cds_sparse_library_destroy(P(cds_sparse_library, sl)); }
JNIEXPORT void JNICALL CLS(sparseLibrarySetGrowable)(JNIEnv *env, jclass c, jlong sl, jboolean on) {
    int rc = cds_sparse_library_set_growable(P(cds_sparse_library, sl), on ? 1 : 0);
    if (rc) throw_for(env, rc);
}
JNIEXPORT void JNICALL CLS(sparseLibraryReserve)(JNIEnv *env, jclass c, jlong sl, jlong capImages, jlong capPixels) {
    int rc = cds_sparse_library_reserve(P(cds_sparse_library, sl), capImages, capPixels);
    if (rc) throw_for(env, rc);
}
JNIEXPORT void JNICALL CLS(sparseLibrarySetDensityLimit)(JNIEnv *env, jclass c, jlong sl, jdouble limit) {
    int rc = cds_sparse_library_set_density_limit(P(cds_sparse_library, sl), limit);
    if (rc) throw_for(env, rc);
}
JNIEXPORT void JNICALL CLS(sparseLibrarySetMinThreshold)(JNIEnv *env, jclass c, jlong sl, jint minDataThreshold) {
    int rc = cds_sparse_library_set_min_threshold(P(cds_sparse_library, sl), minDataThreshold);
    if (rc) throw_for(env, rc);
}
JNIEXPORT void JNICALL CLS(sparseLibraryAppend)(JNIEnv *env, jclass c, jlong sl, jlong count, jobject directRgb) {
    int32_t w = 0, h = 0;
    const int64_t need = (count < 0 || cds_sparse_library_dims(P(cds_sparse_library, sl), &w, &h) != CDS_OK) ? -1 : (int64_t)3 * w * h * count;
    const uint8_t *px = (const uint8_t *)direct_bytes(env, directRgb, need, "directRgb is not a direct buffer of count * 3 * w * h bytes");
    if (!px) return;
    int rc = cds_sparse_library_append_host(P(cds_sparse_library, sl), count, px);
    if (rc) throw_for(env, rc);
}
/* cds_sparse_library_append: dense resident slots [first, first + count) of `lib` converted on the device (what follows libraryLoadVariantImages when
 * the target library is filled from a variant shard) */
JNIEXPORT void JNICALL CLS(sparseLibraryAppendFrom)(JNIEnv *env, jclass c, jlong sl, jlong lib, jlong first, jlong count) {
    int rc = cds_sparse_library_append(P(cds_sparse_library, sl), P(cds_library, lib), first, count);
    if (rc) throw_for(env, rc);
}
/* a String[] as a malloc'ed array of strdup'ed UTF-8 paths (NULL elements stay NULL when allow_null); every local reference is dropped
 * inside the loop: a library has millions of files and the local-reference table holds a few hundred entries.  NULL on failure. */
static const char **copy_paths(JNIEnv *env, jobjectArray paths, jsize n, int allow_null) {
    const char **cp = (const char **)calloc((size_t)n + 1, sizeof(char *));
    if (!cp) return NULL;
    for (jsize i = 0; i < n; i++) {
        jstring str = (jstring)(*env)->GetObjectArrayElement(env, paths, i);
        const char *u = str ? (*env)->GetStringUTFChars(env, str, NULL) : NULL;
        if (u) {
            cp[i] = strdup(u);
            (*env)->ReleaseStringUTFChars(env, str, u);
        }
        if (str) (*env)->DeleteLocalRef(env, str);
        if (!cp[i] && !(allow_null && !str)) {
            for (jsize k = 0; k < i; k++) free((void *)cp[k]);
            free(cp);
            return NULL;
        }
    }
    return cp;
}
static void free_paths(const char **cp, jsize n) {
    if (!cp) return;
    for (jsize i = 0; i < n; i++) free((void *)cp[i]);
    free(cp);
}
/* cds_library_load_files: slots first.. from the targets' files and their GradientImage / ZGapImage files (either array, or single
 * elements, may be null: no such variant) */
JNIEXPORT void JNICALL CLS(libraryLoadFiles)(JNIEnv *env, jclass c, jlong lib, jlong first, jobjectArray targets, jobjectArray gradients, jobjectArray zgaps, jint threads) {
    if (!targets) { throw_iae(env, "targets is null"); return; }
    jsize n = (*env)->GetArrayLength(env, targets);
    if ((gradients && (*env)->GetArrayLength(env, gradients) != n) || (zgaps && (*env)->GetArrayLength(env, zgaps) != n)) {
        throw_iae(env, "gradients / zgaps must have one entry (or null) per target");
        return;
    }
    const char **tp = copy_paths(env, targets, n, 0);
    const char **gp = gradients ? copy_paths(env, gradients, n, 1) : NULL;
    const char **zp = zgaps ? copy_paths(env, zgaps, n, 1) : NULL;
    int rc = (!tp || (gradients && !gp) || (zgaps && !zp)) ? CDS_ERR_INVALID_ARGUMENT : cds_library_load_files(P(cds_library, lib), first, n, tp, gp, zp, threads);
    free_paths(tp, n);
    free_paths(gp, n);
    free_paths(zp, n);
    if (rc == CDS_ERR_INVALID_ARGUMENT && (!tp || (gradients && !gp) || (zgaps && !zp))) { throw_iae(env, "a path is null or not a valid string"); return; }
    if (rc) throw_for(env, rc);
}
/* ---- the variant shard (cds_variant_shard_*): kernel 2's inputs of a library in one file, decoded on the GPU ---- */
JNIEXPORT jlong JNICALL CLS(variantShardCreate)(JNIEnv *env, jclass c, jstring path, jint w, jint h) {
    const char *u = path ? (*env)->GetStringUTFChars(env, path, NULL) : NULL;
    if (!u) { throw_iae(env, "path is null"); return 0; }
    cds_variant_writer *wr = NULL;
    int rc = cds_variant_shard_create(u, w, h, &wr);
    (*env)->ReleaseStringUTFChars(env, path, u);
    if (rc) { throw_for(env, rc); return 0; }
    return (jlong)(intptr_t)wr;
}
JNIEXPORT void JNICALL CLS(variantShardAppendFiles)(JNIEnv *env, jclass c, jlong writer, jobjectArray targets, jobjectArray gradients, jobjectArray zgaps, jint threads) {
    if (!targets) { throw_iae(env, "targets is null"); return; }
    jsize n = (*env)->GetArrayLength(env, targets);
    if ((gradients && (*env)->GetArrayLength(env, gradients) != n) || (zgaps && (*env)->GetArrayLength(env, zgaps) != n)) {
        throw_iae(env, "gradients / zgaps must have one entry (or null) per target");
        return;
    }
    const char **tp = copy_paths(env, targets, n, 0);
    const char **gp = gradients ? copy_paths(env, gradients, n, 1) : NULL;
    const char **zp = zgaps ? copy_paths(env, zgaps, n, 1) : NULL;
    const int bad = !tp || (gradients && !gp) || (zgaps && !zp);
    int rc = bad ? CDS_ERR_INVALID_ARGUMENT : cds_variant_shard_append_files(P(cds_variant_writer, writer), n, tp, gp, zp, threads);
    free_paths(tp, n);
    free_paths(gp, n);
    free_paths(zp, n);
    if (bad) { throw_iae(env, "a path is null or not a valid string"); return; }
    if (rc) throw_for(env, rc);
}
JNIEXPORT void JNICALL CLS(variantShardFinish)(JNIEnv *env, jclass c, jlong writer) {
    int rc = cds_variant_shard_finish(P(cds_variant_writer, writer));
    if (rc) throw_for(env, rc);
}
JNIEXPORT jlong JNICALL CLS(variantShardOpen)(JNIEnv *env, jclass c, jstring path) {
    const char *u = path ? (*env)->GetStringUTFChars(env, path, NULL) : NULL;
    if (!u) { throw_iae(env, "path is null"); return 0; }
    cds_variant_shard *sh = NULL;
    int rc = cds_variant_shard_open(u, &sh);
    (*env)->ReleaseStringUTFChars(env, path, u);
    if (rc) { throw_for(env, rc); return 0; }
    return (jlong)(intptr_t)sh;
}
JNIEXPORT jlong JNICALL CLS(variantShardSize)(JNIEnv *env, jclass c, jlong shard) {
    int64_t n = 0;
    int rc = cds_variant_shard_info(P(cds_variant_shard, shard), NULL, NULL, &n, NULL);
    if (rc) { throw_for(env, rc); return 0; }
    return (jlong)n;
}
JNIEXPORT jint JNICALL CLS(variantShardTargetFlags)(JNIEnv *env, jclass c, jlong shard, jlong target) {
    uint32_t f = 0;
    int rc = cds_variant_shard_target_info(P(cds_variant_shard, shard), target, 1, &f, NULL);
    if (rc) { throw_for(env, rc); return 0; }
    return (jint)f;
}
JNIEXPORT void JNICALL CLS(variantShardClose)(JNIEnv *env, jclass c, jlong shard) {
    cds_variant_shard_close(P(cds_variant_shard, shard));
}
JNIEXPORT void JNICALL CLS(libraryLoadVariants)(JNIEnv *env, jclass c, jlong lib, jlong first, jlong shard, jlongArray targets, jint threads) {
    if (!targets) { throw_iae(env, "targets is null"); return; }
    jsize n = (*env)->GetArrayLength(env, targets);
    jlong *t = (*env)->GetLongArrayElements(env, targets, NULL);
    if (!t) return;
    int rc = cds_library_load_variants(P(cds_library, lib), first, P(cds_variant_shard, shard), (const int64_t *)t, n, threads);
    (*env)->ReleaseLongArrayElements(env, targets, t, JNI_ABORT);
    if (rc) throw_for(env, rc);
}
JNIEXPORT void JNICALL CLS(libraryLoadVariantImages)(JNIEnv *env, jclass c, jlong lib, jlong first, jlong shard, jlongArray targets, jint threads) {
    if (!targets) { throw_iae(env, "targets is null"); return; }
    jsize n = (*env)->GetArrayLength(env, targets);
    jlong *t = (*env)->GetLongArrayElements(env, targets, NULL);
    if (!t) return;
    int rc = cds_library_load_variant_images(P(cds_library, lib), first, P(cds_variant_shard, shard), (const int64_t *)t, n, threads);
    (*env)->ReleaseLongArrayElements(env, targets, t, JNI_ABORT);
    if (rc) throw_for(env, rc);
}
JNIEXPORT void JNICALL CLS(sparseLibraryAppendFiles)(JNIEnv *env, jclass c, jlong sl, jobjectArray paths, jint threads) {
    if (!paths) { throw_iae(env, "paths is null"); return; }
    jsize n = (*env)->GetArrayLength(env, paths);
    const char **cp = (const char **)calloc((size_t)n + 1, sizeof(char *));
    jstring *js = (jstring *)calloc((size_t)n + 1, sizeof(jstring));
    /* the paths are copied out and every local reference is dropped inside the loop: a library has millions of files and the
     * local-reference table holds a few hundred entries */
    int bad = 0;
    for (jsize i = 0; i < n && !bad; i++) {
        jstring str = (jstring)(*env)->GetObjectArrayElement(env, paths, i);
        const char *u = str ? (*env)->GetStringUTFChars(env, str, NULL) : NULL;
        if (u) {
            cp[i] = strdup(u);
            (*env)->ReleaseStringUTFChars(env, str, u);
        }
        if (str) (*env)->DeleteLocalRef(env, str);
        if (!cp[i]) bad = 1;
    }
    (void)js;
    int rc = bad ? CDS_ERR_INVALID_ARGUMENT : cds_sparse_library_append_files(P(cds_sparse_library, sl), cp, n, threads);
    for (jsize i = 0; i < n; i++) free((void *)cp[i]);
    free(cp);
    free(js);
    if (bad) throw_iae(env, "a path is null");
    else if (rc) throw_for(env, rc);
}
JNIEXPORT jlong JNICALL CLS(sparseLibrarySize)(JNIEnv *env, jclass c, jlong sl) { return cds_sparse_library_size(P(cds_sparse_library, sl)); }
JNIEXPORT jlong JNICALL CLS(sparseLibraryPixels)(JNIEnv *env, jclass c, jlong sl) { return cds_sparse_library_pixels(P(cds_sparse_library, sl)); }
JNIEXPORT void JNICALL CLS(sparseLibraryClear)(JNIEnv *env, jclass c, jlong sl) {
    int rc = cds_sparse_library_clear(P(cds_sparse_library, sl));
    if (rc) throw_for(env, rc);
}
JNIEXPORT void JNICALL CLS(searchSparse)(JNIEnv *env, jclass c, jlong ms, jlong sl, jlong first, jlong count, jintArray scores, jbyteArray mirrored) {
    search_common(env, 1, ms, sl, first, count, scores, mirrored);
}
JNIEXPORT jlong JNICALL CLS(searchSparseMatchList)(JNIEnv *env, jclass c, jlong ms, jlong sl, jlong first, jlong count, jdouble pct) {
    cds_match_list *list = NULL;
    int64_t n = 0;
    int rc = cds_search_sparse_match_list(P(cds_maskset, ms), P(cds_sparse_library, sl), first, count, pct, &list, &n);
    if (rc) { throw_for(env, rc); return 0; }
    return (jlong)(intptr_t)list;
}
JNIEXPORT jlong JNICALL CLS(matchListSize)(JNIEnv *env, jclass c, jlong list) { return cds_match_list_size(P(cds_match_list, list)); }
JNIEXPORT void JNICALL CLS(matchListFetch)(JNIEnv *env, jclass c, jlong list, jlong first, jlong count, jintArray triples) {
    if ((jlong)(*env)->GetArrayLength(env, triples) < 3 * count) { throw_for(env, CDS_ERR_INVALID_ARGUMENT); return; }
    jint *t = (*env)->GetIntArrayElements(env, triples, NULL);   /* cds_match = 3 x int32: the array IS the match buffer */
    int rc = cds_match_list_fetch(P(cds_match_list, list), first, count, (cds_match *)t);
    (*env)->ReleaseIntArrayElements(env, triples, t, 0);
    if (rc) throw_for(env, rc);
}
JNIEXPORT void JNICALL CLS(matchListDestroy)(JNIEnv *env, jclass c, jlong list) { cds_match_list_destroy(P(cds_match_list, list)); }

JNIEXPORT void JNICALL CLS(searchSparseTopK)(JNIEnv *env, jclass c, jlong ms, jlong sl, jlong first, jlong count, jint k, jintArray targets,
                                             jintArray scores) {
    const int64_t need = k < 1 ? -1 : (int64_t)cds_maskset_size(P(cds_maskset, ms)) * k;
    if (!has_len(env, targets, need, "targets is shorter than nMasks * k") || !has_len(env, scores, need, "scores is shorter than nMasks * k")) return;
    jint *t = (*env)->GetIntArrayElements(env, targets, NULL), *s = (*env)->GetIntArrayElements(env, scores, NULL);
    int rc = cds_search_sparse_topk(P(cds_maskset, ms), P(cds_sparse_library, sl), first, count, k, (int32_t *)t, (int32_t *)s);
    (*env)->ReleaseIntArrayElements(env, targets, t, 0);
    (*env)->ReleaseIntArrayElements(env, scores, s, 0);
    if (rc) throw_for(env, rc);
}

/* ---- several GPUs of one node ---- */
JNIEXPORT jlong JNICALL CLS(commCreate)(JNIEnv *env, jclass c, jlongArray contexts) {
    if (!contexts) { throw_iae(env, "contexts is null"); return 0; }
    jsize n = (*env)->GetArrayLength(env, contexts);
    cds_context **ctxs = (cds_context **)handle_array(env, contexts, n, "contexts");
    if (!ctxs) return 0;
    cds_comm *comm = NULL;
    int rc = cds_comm_create(ctxs, n, &comm);
    free(ctxs);
    if (rc) { throw_for(env, rc); return 0; }
    return (jlong)(intptr_t)comm;
}
JNIEXPORT void JNICALL CLS(commDestroy)(JNIEnv *env, jclass c, jlong comm) { cds_comm_destroy(P(cds_comm, comm)); }
JNIEXPORT void JNICALL CLS(searchSparseShardedTopK)(JNIEnv *env, jclass c, jlong comm, jlongArray masksets, jlongArray slibs, jint k, jintArray targets,
                                                    jintArray scores) {
    /* one mask set and one shard per context of the communicator: the C ABI reads exactly cds_comm_size() entries of each */
    const jsize n = (jsize)cds_comm_size(P(cds_comm, comm));
    const cds_maskset **ms = (const cds_maskset **)handle_array(env, masksets, n, "masksets must hold one handle per context of the communicator");
    if (!ms) return;
    const cds_sparse_library **sl = (const cds_sparse_library **)handle_array(env, slibs, n, "slibs must hold one handle per context of the communicator");
    if (!sl) { free(ms); return; }
    const int64_t need = (k < 1 || n < 1) ? -1 : (int64_t)cds_maskset_size(ms[0]) * k;
    if (!has_len(env, targets, need, "targets is shorter than nMasks * k") || !has_len(env, scores, need, "scores is shorter than nMasks * k")) { free(ms); free(sl); return; }
    jint *t = (*env)->GetIntArrayElements(env, targets, NULL), *s = (*env)->GetIntArrayElements(env, scores, NULL);
    int rc = cds_search_sparse_sharded_topk(P(cds_comm, comm), ms, sl, k, NULL, (int32_t *)t, (int32_t *)s);
    (*env)->ReleaseIntArrayElements(env, targets, t, 0);
    (*env)->ReleaseIntArrayElements(env, scores, s, 0);
    free(ms); free(sl);
    if (rc) throw_for(env, rc);
}
JNIEXPORT jintArray JNICALL CLS(searchSparseShardedMatches)(JNIEnv *env, jclass c, jlong comm, jlongArray masksets, jlongArray slibs, jdouble pct) {
    /* one mask set and one shard per context of the communicator: the C ABI reads exactly cds_comm_size() entries of each */
    const jsize n = (jsize)cds_comm_size(P(cds_comm, comm));
    const cds_maskset **ms = (const cds_maskset **)handle_array(env, masksets, n, "masksets must hold one handle per context of the communicator");
    if (!ms) return NULL;
    const cds_sparse_library **sl = (const cds_sparse_library **)handle_array(env, slibs, n, "slibs must hold one handle per context of the communicator");
    if (!sl) { free(ms); return NULL; }
    int64_t cap = 1 << 16, cnt = 0;
    cds_match *buf = (cds_match *)malloc((size_t)cap * sizeof(cds_match));
    int rc = buf ? cds_search_sparse_sharded_matches(P(cds_comm, comm), ms, sl, pct, NULL, buf, cap, &cnt) : CDS_ERR_OUT_OF_MEMORY;
    if (buf && rc == CDS_ERR_OUT_OF_MEMORY && cnt > cap) {   /* longer than the first guess: once more, exactly sized */
        free(buf);
        cap = cnt;
        buf = (cds_match *)malloc((size_t)cap * sizeof(cds_match));
        rc = buf ? cds_search_sparse_sharded_matches(P(cds_comm, comm), ms, sl, pct, NULL, buf, cap, &cnt) : CDS_ERR_OUT_OF_MEMORY;
    }
    jintArray out = NULL;
    if (rc) {
        throw_for(env, rc);
    } else {
        out = (*env)->NewIntArray(env, (jsize)(3 * cnt));
        if (out) (*env)->SetIntArrayRegion(env, out, 0, (jsize)(3 * cnt), (const jint *)buf);   /* a cds_match IS three int32 */
    }
    free(buf); free(ms); free(sl);
    return out;
}
JNIEXPORT jlongArray JNICALL CLS(searchSparseSharded)(JNIEnv *env, jclass c, jlong comm, jlongArray masksets, jlongArray slibs, jintArray scores,
                                                      jbyteArray mirrored) {
    /* one mask set and one shard per context of the communicator: the C ABI reads exactly cds_comm_size() entries of each */
    const jsize n = (jsize)cds_comm_size(P(cds_comm, comm));
    const cds_maskset **ms = (const cds_maskset **)handle_array(env, masksets, n, "masksets must hold one handle per context of the communicator");
    if (!ms) return NULL;
    const cds_sparse_library **sl = (const cds_sparse_library **)handle_array(env, slibs, n, "slibs must hold one handle per context of the communicator");
    if (!sl) { free(ms); return NULL; }
    int64_t total = 0;
    for (jsize i = 0; i < n; i++) total += cds_sparse_library_size(sl[i]);
    const int64_t need = n < 1 ? -1 : (int64_t)cds_maskset_size(ms[0]) * total;
    if (!has_len(env, scores, need, "scores is shorter than nMasks * (sum of the shards' sizes)") ||
        !has_len(env, mirrored, need, "mirrored is shorter than nMasks * (sum of the shards' sizes)")) { free(ms); free(sl); return NULL; }
    int64_t *offs = (int64_t *)calloc((size_t)n + 1, sizeof(int64_t));
    jint *s = (*env)->GetIntArrayElements(env, scores, NULL);
    jbyte *m = (*env)->GetByteArrayElements(env, mirrored, NULL);
    int rc = cds_search_sparse_sharded(P(cds_comm, comm), ms, sl, offs, (int32_t *)s, (uint8_t *)m);
    (*env)->ReleaseIntArrayElements(env, scores, s, 0);
    (*env)->ReleaseByteArrayElements(env, mirrored, m, 0);
    jlongArray out = NULL;
    if (rc) {
        throw_for(env, rc);
    } else {
        out = (*env)->NewLongArray(env, n + 1);
        (*env)->SetLongArrayRegion(env, out, 0, n + 1, (const jlong *)offs);
    }
    free(ms); free(sl); free(offs);
    return out;
}

/* ---- GradientAreaGap scorer ---- */
JNIEXPORT jlong JNICALL CLS(shapesetCreate)(JNIEnv *env, jclass c, jlong ctx, jint nMasks, jbyteArray masksRgb, jint w, jint h, jint queryThr,
                                            jint border, jboolean mirror, jbyteArray roi, jintArray rects) {
    cds_rect r[CDS_MAX_RECTS];
    int n = get_rects(env, rects, r);
    if (n < 0) return 0;
    if (nMasks < 1 || !has_len(env, masksRgb, image_bytes(w, h) < 0 ? -1 : nMasks * image_bytes(w, h), "masksRgb is shorter than nMasks * 3 * w * h")) return 0;
    if (roi && !has_len(env, roi, (int64_t)w * h, "roi is shorter than w * h")) return 0;
    jbyte *px = (*env)->GetByteArrayElements(env, masksRgb, NULL);
    jbyte *rp = roi ? (*env)->GetByteArrayElements(env, roi, NULL) : NULL;
    cds_shapeset *ss = NULL;
    int rc = cds_shapeset_create_bordered(P(cds_context, ctx), nMasks, (const uint8_t *)px, w, h, queryThr, border, mirror ? 1 : 0,
                                          (const uint8_t *)rp, r, n, &ss);
    (*env)->ReleaseByteArrayElements(env, masksRgb, px, JNI_ABORT);
    if (rp) (*env)->ReleaseByteArrayElements(env, roi, rp, JNI_ABORT);
    if (rc) { throw_for(env, rc); return 0; }
    return (jlong)(intptr_t)ss;
}
JNIEXPORT void JNICALL CLS(shapesetDestroy)(JNIEnv *env, jclass c, jlong ss) { cds_shapeset_destroy(P(cds_shapeset, ss)); }
JNIEXPORT void JNICALL CLS(shapesetMaskStats)(JNIEnv *env, jclass c, jlong ss, jlongArray maskSizes, jlongArray highSizes) {
    const int64_t nm = cds_shapeset_size(P(cds_shapeset, ss));
    if (!has_len(env, maskSizes, nm, "maskSizes is shorter than the shape set") || !has_len(env, highSizes, nm, "highExpressionSizes is shorter than the shape set")) return;
    jlong *a = (*env)->GetLongArrayElements(env, maskSizes, NULL), *b = (*env)->GetLongArrayElements(env, highSizes, NULL);
    int rc = cds_shapeset_mask_stats(P(cds_shapeset, ss), (int64_t *)a, (int64_t *)b);
    (*env)->ReleaseLongArrayElements(env, maskSizes, a, 0);
    (*env)->ReleaseLongArrayElements(env, highSizes, b, 0);
    if (rc) throw_for(env, rc);
}
JNIEXPORT void JNICALL CLS(shapeScorePair)(JNIEnv *env, jclass c, jlong ss, jint maskIndex, jbyteArray targetRgb, jshortArray grad, jbyteArray zgapRgb,
                                           jint w, jint h, jlongArray out) {
    if (!has_len(env, targetRgb, image_bytes(w, h), "targetRgb is shorter than 3 * w * h") || !has_len(env, out, 3, "out needs 3 elements")) return;
    if (grad && !has_len(env, grad, (int64_t)w * h, "grad is shorter than w * h")) return;
    if (zgapRgb && !has_len(env, zgapRgb, image_bytes(w, h), "zgapRgb is shorter than 3 * w * h")) return;
    jbyte *t = (*env)->GetByteArrayElements(env, targetRgb, NULL);
    jshort *g = grad ? (*env)->GetShortArrayElements(env, grad, NULL) : NULL;
    jbyte *z = zgapRgb ? (*env)->GetByteArrayElements(env, zgapRgb, NULL) : NULL;
    int64_t gap = -1, high = -1;
    uint8_t mirrored = 0;
    int rc = cds_shape_score_pair(P(cds_shapeset, ss), maskIndex, (const uint8_t *)t, (const uint16_t *)g, (const uint8_t *)z, w, h, &gap, &high, &mirrored);
    (*env)->ReleaseByteArrayElements(env, targetRgb, t, JNI_ABORT);
    if (g) (*env)->ReleaseShortArrayElements(env, grad, g, JNI_ABORT);
    if (z) (*env)->ReleaseByteArrayElements(env, zgapRgb, z, JNI_ABORT);
    if (rc) { throw_for(env, rc); return; }
    jlong v[3] = {gap, high, mirrored};
    (*env)->SetLongArrayRegion(env, out, 0, 3, v);
}
JNIEXPORT void JNICALL CLS(gradscore)(JNIEnv *env, jclass c, jlong ss, jlong lib, jintArray pairMask, jlongArray pairSlot, jlongArray gap,
                                      jlongArray high, jbyteArray mirrored) {
    if (!pairMask) { throw_iae(env, "pairMask is null"); return; }
    jsize n = (*env)->GetArrayLength(env, pairMask);
    if (!has_len(env, pairSlot, n, "pairSlot is shorter than pairMask") || !has_len(env, gap, n, "gradientAreaGap is shorter than pairMask") ||
        !has_len(env, high, n, "highExpressionArea is shorter than pairMask") || !has_len(env, mirrored, n, "mirrored is shorter than pairMask")) return;
    jint *pm = (*env)->GetIntArrayElements(env, pairMask, NULL);
    jlong *ps = (*env)->GetLongArrayElements(env, pairSlot, NULL), *g = (*env)->GetLongArrayElements(env, gap, NULL),
          *hi = (*env)->GetLongArrayElements(env, high, NULL);
    jbyte *m = (*env)->GetByteArrayElements(env, mirrored, NULL);
    int rc = cds_gradscore(P(cds_shapeset, ss), P(cds_library, lib), n, (const int32_t *)pm, (const int64_t *)ps, (int64_t *)g, (int64_t *)hi, (uint8_t *)m);
    (*env)->ReleaseIntArrayElements(env, pairMask, pm, JNI_ABORT);
    (*env)->ReleaseLongArrayElements(env, pairSlot, ps, JNI_ABORT);
    (*env)->ReleaseLongArrayElements(env, gap, g, 0);
    (*env)->ReleaseLongArrayElements(env, high, hi, 0);
    (*env)->ReleaseByteArrayElements(env, mirrored, m, 0);
    if (rc) throw_for(env, rc);
}
/* kernel 2 over dense shards: pairs of (mask index, LIBRARY index), every pair scored on the GPU owning its target */
JNIEXPORT jlongArray JNICALL CLS(gradscoreSharded)(JNIEnv *env, jclass c, jlong comm, jlongArray shapesets, jlongArray libs, jlongArray shardSizes,
                                                   jintArray pairMask, jlongArray pairTarget, jlongArray gap, jlongArray high, jbyteArray mirrored) {
    const jsize n = (jsize)cds_comm_size(P(cds_comm, comm));
    if (!pairMask) { throw_iae(env, "pairMask is null"); return NULL; }
    const jsize np = (*env)->GetArrayLength(env, pairMask);
    if (!has_len(env, pairTarget, np, "pairTarget is shorter than pairMask") || !has_len(env, gap, np, "gradientAreaGap is shorter than pairMask") ||
        !has_len(env, high, np, "highExpressionArea is shorter than pairMask") || !has_len(env, mirrored, np, "mirrored is shorter than pairMask")) return NULL;
    if (shardSizes && !has_len(env, shardSizes, n, "shardSizes must hold one size per context of the communicator")) return NULL;
    const cds_shapeset **ss = (const cds_shapeset **)handle_array(env, shapesets, n, "shapesets must hold one handle per context of the communicator");
    if (!ss) return NULL;
    const cds_library **lb = (const cds_library **)handle_array(env, libs, n, "libs must hold one handle per context of the communicator");
    if (!lb) { free(ss); return NULL; }
    int64_t *offs = (int64_t *)calloc((size_t)n + 1, sizeof(int64_t));
    jlong *sz = shardSizes ? (*env)->GetLongArrayElements(env, shardSizes, NULL) : NULL;
    jint *pm = (*env)->GetIntArrayElements(env, pairMask, NULL);
    jlong *pt = (*env)->GetLongArrayElements(env, pairTarget, NULL), *g = (*env)->GetLongArrayElements(env, gap, NULL),
          *hi = (*env)->GetLongArrayElements(env, high, NULL);
    jbyte *m = (*env)->GetByteArrayElements(env, mirrored, NULL);
    int rc = cds_gradscore_sharded(P(cds_comm, comm), ss, lb, (const int64_t *)sz, np, (const int32_t *)pm, (const int64_t *)pt, offs, (int64_t *)g,
                                   (int64_t *)hi, (uint8_t *)m);
    if (sz) (*env)->ReleaseLongArrayElements(env, shardSizes, sz, JNI_ABORT);
    (*env)->ReleaseIntArrayElements(env, pairMask, pm, JNI_ABORT);
    (*env)->ReleaseLongArrayElements(env, pairTarget, pt, JNI_ABORT);
    (*env)->ReleaseLongArrayElements(env, gap, g, 0);
    (*env)->ReleaseLongArrayElements(env, high, hi, 0);
    (*env)->ReleaseByteArrayElements(env, mirrored, m, 0);
    jlongArray out = NULL;
    if (rc) {
        throw_for(env, rc);
    } else {
        out = (*env)->NewLongArray(env, n + 1);
        if (out) (*env)->SetLongArrayRegion(env, out, 0, n + 1, (const jlong *)offs);
    }
    free(ss); free(lb); free(offs);
    return out;
}
JNIEXPORT void JNICALL CLS(makeZgap)(JNIEnv *env, jclass c, jlong ctx, jbyteArray targetRgb, jint w, jint h, jint threshold, jdouble radius,
                                     jintArray rects, jbyteArray zgapOut) {
    cds_rect r[CDS_MAX_RECTS];
    int n = get_rects(env, rects, r);
    if (n < 0) return;
    if (!has_len(env, targetRgb, image_bytes(w, h), "targetRgb is shorter than 3 * w * h") || !has_len(env, zgapOut, image_bytes(w, h), "zgapRgbOut is shorter than 3 * w * h")) return;
    jbyte *t = (*env)->GetByteArrayElements(env, targetRgb, NULL), *z = (*env)->GetByteArrayElements(env, zgapOut, NULL);
    int rc = cds_make_zgap(P(cds_context, ctx), (const uint8_t *)t, w, h, threshold, radius, r, n, (uint8_t *)z);
    (*env)->ReleaseByteArrayElements(env, targetRgb, t, JNI_ABORT);
    (*env)->ReleaseByteArrayElements(env, zgapOut, z, 0);
    if (rc) throw_for(env, rc);
}
