package org.janelia.colormipsearch.cds.gpu;

import java.nio.ByteBuffer;
import java.util.ArrayList;
import java.util.List;

import org.janelia.colormipsearch.imageprocessing.ImageArray;
import org.janelia.colormipsearch.imageprocessing.ImageArrayBytes;

/**
 * The resident target library of a colorDepthSearch run: a cds_sparse_library (signal-pixel records, ~4 % of the packed
 * RGB bytes, so a million MIPs stay in one GPU's HBM) plus, per resident image, the caller's own key (the neuron entity).
 * It replaces the host-side Guava cache of decoded MIPs (cmd/CachedMIPsUtils.java:60-100) for the pixel-match scan.
 * Images are handed over in batches through one direct staging buffer.
 */
public final class GpuTargetLibrary<K> implements AutoCloseable {
    private final long handle;
    private final int width, height;
    private final List<K> keys = new ArrayList<>();
    private final ByteBuffer staging;
    private final int stagingImages;
    private final List<K> pendingKeys = new ArrayList<>();

    public GpuTargetLibrary(long ctx, int width, int height, long capacityImages, long capacityRecords, int stagingImages) {
        this.width = width;
        this.height = height;
        this.stagingImages = Math.max(1, stagingImages);
        this.handle = CdsHip.sparseLibraryCreate(ctx, width, height, capacityImages, capacityRecords);
        CdsHip.sparseLibrarySetGrowable(handle, true);   // the capacities are a first guess: like the reference's MIP cache
                                                         // (cmd/CachedMIPsUtils.java:60-100) the library sizes itself with use
        this.staging = ByteBuffer.allocateDirect(this.stagingImages * 3 * width * height);
    }

    long nativeHandle() {
        return handle;
    }

    /**
     * Pixels whose brightest channel is at or below the run's target threshold are never looked at by the scan
     * (PixelMatchColorDepthSearchAlgorithm: "if (red2 > targetThreshold || ...)"): a library built for one --dataThreshold need not
     * hold them.  Call before the first add(); a search with a smaller threshold is then refused by the native library.
     */
    public void setMinDataThreshold(int targetThreshold) {
        CdsHip.sparseLibrarySetMinThreshold(handle, targetThreshold);
    }

    public int size() {
        return keys.size();
    }

    public int getWidth() {
        return width;
    }

    public int getHeight() {
        return height;
    }

    public K key(int index) {
        return keys.get(index);
    }

    /** queue one target; it becomes resident (and searchable) at the next flush(). */
    public synchronized void add(K key, ImageArray<?> image) {
        if (image.getWidth() != width || image.getHeight() != height) {
            throw new IllegalArgumentException(String.format(
                    "Invalid image size - target's image size (%d, %d) must match query's image size: (%d, %d)",
                    image.getWidth(), image.getHeight(), width, height));
        }
        staging.put(ImageArrayBytes.rgb(image));
        pendingKeys.add(key);
        if (pendingKeys.size() == stagingImages) {
            flush();
        }
    }

    public synchronized void flush() {
        if (!pendingKeys.isEmpty()) {
            CdsHip.sparseLibraryAppend(handle, pendingKeys.size(), staging);
            keys.addAll(pendingKeys);
            pendingKeys.clear();
            staging.clear();
        }
    }

    /**
     * Targets straight from a variant shard (GpuVariantShard: the library's MIPs and variants in one file): their images are decoded on
     * the device into a dense staging library of this library's context and converted from there -- no ImageJ decode, no Java heap:
     * 100 K targets per second where the per-file path manages the loader's few thousand.  keys.get(i) names shard target targets[i].
     */
    public synchronized void addFromShard(long ctx, GpuVariantShard shard, long[] targets, List<K> targetKeys, int threads) {
        if (targets.length != targetKeys.size()) {
            throw new IllegalArgumentException("one key per target");
        }
        flush();
        final int batch = (int) Math.min(1024, Math.max(1, targets.length));
        long dense = CdsHip.libraryCreate(ctx, width, height, batch);
        try {
            for (int first = 0; first < targets.length; first += batch) {
                int n = Math.min(batch, targets.length - first);
                shard.loadImages(dense, 0, java.util.Arrays.copyOfRange(targets, first, first + n), threads);
                CdsHip.sparseLibraryAppendFrom(handle, dense, 0, n);
                keys.addAll(targetKeys.subList(first, first + n));
            }
        } finally {
            CdsHip.libraryDestroy(dense);
        }
    }

    @Override
    public void close() {
        CdsHip.sparseLibraryDestroy(handle);
    }
}
