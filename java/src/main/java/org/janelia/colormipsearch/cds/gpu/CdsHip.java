package org.janelia.colormipsearch.cds.gpu;

import java.nio.ByteBuffer;

/**
 * The JNI surface of libcds_hip.so (include/cds_hip.h): one native method per C entry point the plugin uses; the
 * implementations are in src/main/c/cds_hip_jni.c.  Handles are opaque pointers carried as long.  A failing call throws:
 * CDS_ERR_INVALID_ARGUMENT becomes IllegalArgumentException (where the reference throws it too), everything else
 * IllegalStateException with cds_last_error() as the message.  There is no CPU fallback behind any of these.
 */
final class CdsHip {
    static {
        System.loadLibrary("cds_hip_jni");   // links libcds_hip.so
    }

    private CdsHip() {
    }

    // ---- context ----
    static native long contextCreate(int device);
    static native void contextDestroy(long ctx);
    static native void contextSynchronize(long ctx);
    /** Idle device blocks of the cache of the context go back to the driver; returns the bytes freed. */
    static native long contextTrim(long ctx);
    static native void contextSetCacheLimit(long ctx, long bytes);

    // ---- pixel-match mask sets (n x PixelMatchColorDepthSearchAlgorithm) ----
    /** masksRgb: nMasks x 3*w*h bytes; rectsXYXY: x0,y0,x1,y1 per excluded rectangle (at most 8). */
    static native long masksetCreate(long ctx, int nMasks, byte[] masksRgb, int w, int h, int maskThreshold, int dataThreshold,
                                     double zTolerance, int xyShift, boolean mirror, int[] rectsXYXY);
    static native void masksetDestroy(long maskset);
    static native void masksetQuerySizes(long maskset, int[] sizes);
    /** calculateMatchingScore of one host image: score | mirrored << 32. */
    static native long scorePair(long maskset, int maskIndex, byte[] targetRgb, int w, int h);

    // ---- dense resident library (targets + gradient / zgap variants; what the shape scorer reads) ----
    static native long libraryCreate(long ctx, int w, int h, long capacity);
    static native void libraryDestroy(long lib);
    /** directRgb: count x 3*w*h bytes in a direct buffer. */
    static native void libraryUpload(long lib, long first, long count, ByteBuffer directRgb);
    /** directGrad (count x w*h unsigned shorts, native order) and directZgapRgb may each be null. */
    static native void libraryUploadVariants(long lib, long first, long count, ByteBuffer directGrad, ByteBuffer directZgapRgb);
    /** a GRAY8 gradient image (imageprocessing/ByteImageArray): widened into the 16-bit plane the scorer reads. */
    static native void libraryUploadGradientU8(long lib, long first, long count, ByteBuffer directGrad8);
    /**
     * Slots first.. straight from files: the targets, their GradientImage and ZGapImage files (arrays or single entries may be null: the
     * slot has no such variant), decoded on native threads while the previous 64 slots are uploaded -- the batch form of the per-match
     * NeuronMIPUtils.loadComputeFile calls of cmd/CalculateGradientScoresCmd.java:541-586.
     */
    static native void libraryLoadFiles(long lib, long first, String[] targets, String[] gradients, String[] zgaps, int threads);
    /**
     * The variant shard: every target's MIP, GradientImage and ZGapImage of a library losslessly in one file (written once from the files
     * -- no GPU needed), of which libraryLoadVariants decodes any subset ON THE DEVICE into dense slots: 77 K targets/s where inflating
     * the three files of a target on 16 host cores gives 5 K (cmd/CalculateGradientScoresCmd.java:541-586 loads them per match).
     */
    static native long variantShardCreate(String path, int width, int height);
    static native void variantShardAppendFiles(long writer, String[] targets, String[] gradients, String[] zgaps, int threads);
    static native void variantShardFinish(long writer);
    static native long variantShardOpen(String path);
    static native long variantShardSize(long shard);
    /** bit 0: the target has a gradient, bit 2: it has a zgap image */
    static native int variantShardTargetFlags(long shard, long target);
    static native void variantShardClose(long shard);
    static native void libraryLoadVariants(long lib, long first, long shard, long[] targets, int threads);
    /** the targets' MIPs alone (the slots have no variants): what filling a sparse library from a shard reads */
    static native void libraryLoadVariantImages(long lib, long first, long shard, long[] targets, int threads);
    static native void libraryMakeZgapVariants(long lib, long first, long count, int threshold, double radius, int[] rectsXYXY);
    static native void search(long maskset, long lib, long first, long count, int[] scores, byte[] mirrored);

    // ---- sparse resident library (signal-pixel records: the pixel-match scan's layout) ----
    static native long sparseLibraryCreate(long ctx, int w, int h, long capacityImages, long capacityPixels);
    static native void sparseLibraryDestroy(long slib);
    /** appends enlarge the library when a capacity is lacking (the reference's MIP cache sizes itself with use). */
    static native void sparseLibrarySetGrowable(long slib, boolean on);
    static native void sparseLibraryReserve(long slib, long capacityImages, long capacityPixels);
    static native void sparseLibrarySetDensityLimit(long slib, double maxSignalFraction);
    static native void sparseLibrarySetMinThreshold(long slib, int minDataThreshold);
    static native void sparseLibraryAppend(long slib, long count, ByteBuffer directRgb);
    static native void sparseLibraryAppendFiles(long slib, String[] paths, int threads);
    /** dense resident slots first.. of a library (libraryCreate), converted on the device */
    static native void sparseLibraryAppendFrom(long slib, long lib, long first, long count);
    static native long sparseLibrarySize(long slib);
    static native long sparseLibraryPixels(long slib);
    static native void sparseLibraryClear(long slib);
    static native void searchSparse(long maskset, long slib, long first, long count, int[] scores, byte[] mirrored);
    /**
     * Scan + ColorMIPSearch.isMatch + compaction on the device: the match list is built there to its exact length (one scan);
     * the call returns its handle, matchListSize its length, matchListFetch copies triples (mask, target, score | mirrored << 31)
     * [first, first + count) into triples[0 .. 3 * count).
     */
    static native long searchSparseMatchList(long maskset, long slib, long first, long count, double pctPositivePixels);
    static native long matchListSize(long list);
    static native void matchListFetch(long list, long first, long count, int[] triples);
    static native void matchListDestroy(long list);
    /** per mask the k best-scoring targets, selected on the device: targets / scores are nMasks x k (score | mirrored << 31; -1 = no entry). */
    static native void searchSparseTopK(long maskset, long slib, long first, long count, int k, int[] targets, int[] scores);

    // ---- several GPUs of one node ----
    static native long commCreate(long[] contexts);
    static native void commDestroy(long comm);
    /** scores / mirrored: nMasks x (sum of the shards' sizes), library order; returns shard offsets (n + 1). */
    static native long[] searchSparseSharded(long comm, long[] masksets, long[] slibs, int[] scores, byte[] mirrored);
    /** The shards' match lists (isMatch + compaction on every GPU) as mask, target, score | mirrored << 31 triples, by mask, then library index. */
    static native int[] searchSparseShardedMatches(long comm, long[] masksets, long[] slibs, double pctPositivePixels);
    /** Per mask the k best targets over all shards: targets[m * k + i] (library index, -1 behind the last), scores likewise (| mirrored << 31). */
    static native void searchSparseShardedTopK(long comm, long[] masksets, long[] slibs, int k, int[] targets, int[] scores);

    // ---- GradientAreaGap scorer (n x Shape2DMatchColorDepthSearchAlgorithm) ----
    static native long shapesetCreate(long ctx, int nMasks, byte[] masksRgb, int w, int h, int queryThreshold, int queryBorderSize,
                                      boolean mirror, byte[] roiOrNull, int[] rectsXYXY);
    static native void shapesetDestroy(long shapeset);
    static native void shapesetMaskStats(long shapeset, long[] maskSizes, long[] highExpressionSizes);
    /** out: gradientAreaGap, highExpressionArea, mirrored (0/1); grad / zgapRgb null -> (-1, -1, 0). */
    static native void shapeScorePair(long shapeset, int maskIndex, byte[] targetRgb, short[] grad, byte[] zgapRgb, int w, int h, long[] out);
    static native void gradscore(long shapeset, long lib, int[] pairMask, long[] pairSlot, long[] gradientAreaGap, long[] highExpressionArea,
                                 byte[] mirrored);
    /**
     * The same over dense shards (one library with variants and one replicated shape set per context of the communicator): pairTarget holds
     * LIBRARY indices, shard r owning [offsets[r], offsets[r] + shardSizes[r]) in its slots 0.. (shardSizesOrNull == null: the libraries'
     * capacities); every pair is scored on the GPU that owns its target, all GPUs at once; results in pair order; returns the offsets (n + 1).
     */
    static native long[] gradscoreSharded(long comm, long[] shapesets, long[] libs, long[] shardSizesOrNull, int[] pairMask, long[] pairTarget,
                                          long[] gradientAreaGap, long[] highExpressionArea, byte[] mirrored);
    static native void makeZgap(long ctx, byte[] targetRgb, int w, int h, int threshold, double radius, int[] rectsXYXY, byte[] zgapRgbOut);
}
