package org.janelia.colormipsearch.cds.gpu;

import java.util.List;

/**
 * The gradient-score step's inputs of a whole library in one file: every target's MIP, its GradientImage and its ZGapImage, lossless, in
 * codes the GPU expands (CdsHip.variantShard*).  CalculateGradientScoresCmd loads these three images per match through its variant
 * suppliers (NeuronMIPUtils.loadComputeFile: a file open and an inflate per image on a host core); with a shard the selected targets
 * of a batch are read with one pread each and decoded on the device into the dense slots GpuShape2DMatchColorDepthSearchAlgorithm
 * scores -- bit-equal to what the files decode to.
 *
 * Written once per library (no GPU needed):
 * <pre>
 *   GpuVariantShard.write(path, width, height, targetFiles, gradientFiles, zgapFiles, threads);
 * </pre>
 * and read by any number of runs:
 * <pre>
 *   try (GpuVariantShard shard = GpuVariantShard.open(path)) {
 *       shard.load(denseLibraryHandle, firstSlot, selectedTargets, threads);   // selectedTargets[i] -> slot firstSlot + i
 *   }
 * </pre>
 * Target numbers are positions in the lists given to write().  A target written without a gradient leaves its slot without one (its
 * pairs score -1, as in Shape2DMatchColorDepthSearchAlgorithm.java:155-158); one written without a zgap image gets it from the device
 * (CdsHip.libraryMakeZgapVariants), as the reference computes it from the target when the file is missing.
 */
public final class GpuVariantShard implements AutoCloseable {
    public static final int HAS_GRADIENT = 1;
    public static final int HAS_ZGAP = 4;

    private long handle;
    private final long size;

    private GpuVariantShard(long handle) {
        this.handle = handle;
        this.size = CdsHip.variantShardSize(handle);
    }

    /** Packs the targets' files (entries of gradientFiles / zgapFiles, or the lists themselves, may be null: no such variant). */
    public static void write(String path, int width, int height, List<String> targetFiles, List<String> gradientFiles, List<String> zgapFiles, int threads) {
        long writer = CdsHip.variantShardCreate(path, width, height);
        try {
            final int batch = 4096;   // (bounds the arrays handed to the native side; the shard's target numbers count on across calls)
            for (int first = 0; first < targetFiles.size(); first += batch) {
                int last = Math.min(targetFiles.size(), first + batch);
                CdsHip.variantShardAppendFiles(writer, targetFiles.subList(first, last).toArray(new String[0]),
                        gradientFiles == null ? null : gradientFiles.subList(first, last).toArray(new String[0]),
                        zgapFiles == null ? null : zgapFiles.subList(first, last).toArray(new String[0]), threads);
            }
        } catch (RuntimeException e) {
            try {
                CdsHip.variantShardFinish(writer);   // releases the writer; the file stays without what failed
            } catch (RuntimeException ignored) {
                // the first error is the one to report
            }
            throw e;
        }
        CdsHip.variantShardFinish(writer);
    }

    public static GpuVariantShard open(String path) {
        return new GpuVariantShard(CdsHip.variantShardOpen(path));
    }

    public long size() {
        return size;
    }

    public boolean hasGradient(long target) {
        return (CdsHip.variantShardTargetFlags(handle, target) & HAS_GRADIENT) != 0;
    }

    public boolean hasZgap(long target) {
        return (CdsHip.variantShardTargetFlags(handle, target) & HAS_ZGAP) != 0;
    }

    /** Dense slots firstSlot.. of the library (a CdsHip.libraryCreate handle) from the shard's targets, decoded on the device. */
    public void load(long denseLibrary, long firstSlot, long[] targets, int threads) {
        CdsHip.libraryLoadVariants(denseLibrary, firstSlot, handle, targets, threads);
    }

    /** The targets' MIPs alone into dense slots (no variants): the staging step of filling a GpuTargetLibrary from the shard. */
    public void loadImages(long denseLibrary, long firstSlot, long[] targets, int threads) {
        CdsHip.libraryLoadVariantImages(denseLibrary, firstSlot, handle, targets, threads);
    }

    @Override
    public synchronized void close() {
        if (handle != 0) {
            CdsHip.variantShardClose(handle);
            handle = 0;
        }
    }
}
