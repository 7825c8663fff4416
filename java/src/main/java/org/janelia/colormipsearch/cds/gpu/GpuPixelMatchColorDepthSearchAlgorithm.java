package org.janelia.colormipsearch.cds.gpu;

import java.util.Collections;
import java.util.Map;
import java.util.Set;
import java.util.function.Supplier;

import javax.annotation.Nonnull;

import org.janelia.colormipsearch.cds.ColorDepthSearchAlgorithm;
import org.janelia.colormipsearch.cds.PixelMatchScore;
import org.janelia.colormipsearch.imageprocessing.ImageArray;
import org.janelia.colormipsearch.imageprocessing.ImageArrayBytes;
import org.janelia.colormipsearch.model.ComputeFileType;

/**
 * Drop-in for cds/PixelMatchColorDepthSearchAlgorithm (the pixel-match scorer of one query) computing on an MI355X
 * through libcds_hip.so.  calculateMatchingScore keeps the reference's per-pair contract (one upload + one launch per call,
 * ~0.16 ms); throughput comes from the batch entry points GpuColorMIPSearchProcessor uses on the SAME native mask set
 * ({@link #nativeMaskSet()}).  The native handle is transient: the reference serialises algorithm objects for Spark
 * (cmd/cdsprocess/SparkColorMIPSearchProcessor.java:27-28), so it is re-created lazily from the retained query.
 */
public class GpuPixelMatchColorDepthSearchAlgorithm implements ColorDepthSearchAlgorithm<PixelMatchScore> {
    private static final long serialVersionUID = 1L;

    private final ImageArray<?> queryImage;
    private final int queryThreshold;
    private final int targetThreshold;
    private final boolean mirrorQuery;
    private final double zTolerance;
    private final int xyShift;
    private final int[] excludedRects;
    private transient long handle;
    private transient int querySize = -1;

    public GpuPixelMatchColorDepthSearchAlgorithm(ImageArray<?> queryImage, int queryThreshold, boolean mirrorQuery,
                                                  int targetThreshold, double zTolerance, int xyShift, int[] excludedRectsXYXY) {
        if ((xyShift & 1) == 1) {
            throw new IllegalArgumentException("XY shift parameter must be an even number.");
        }
        this.queryImage = queryImage;
        this.queryThreshold = queryThreshold;
        this.mirrorQuery = mirrorQuery;
        this.targetThreshold = targetThreshold;
        this.zTolerance = zTolerance;
        this.xyShift = xyShift;
        this.excludedRects = excludedRectsXYXY;
    }

    /** the cds_maskset of this query (one mask), created on first use. */
    public synchronized long nativeMaskSet() {
        if (handle == 0) {
            handle = CdsHip.masksetCreate(GpuContext.get(), 1, ImageArrayBytes.rgb(queryImage), queryImage.getWidth(), queryImage.getHeight(),
                    queryThreshold, targetThreshold, zTolerance, xyShift, mirrorQuery, excludedRects);
            int[] size = new int[1];
            CdsHip.masksetQuerySizes(handle, size);
            querySize = size[0];
        }
        return handle;
    }

    @Override
    public ImageArray<?> getQueryImage() {
        return queryImage;
    }

    @Override
    public int getQuerySize() {
        nativeMaskSet();
        return querySize;
    }

    @Override
    public int getQueryFirstPixelIndex() {
        return scanQuery(true);
    }

    @Override
    public int getQueryLastPixelIndex() {
        return scanQuery(false);
    }

    /** first / last query position above the threshold outside the excluded rectangles (row-major index), -1 if none. */
    private int scanQuery(boolean first) {
        byte[] px = ImageArrayBytes.rgb(queryImage);
        int w = queryImage.getWidth(), n = queryImage.getPixelCount();
        for (int k = 0; k < n; k++) {
            int i = first ? k : n - 1 - k;
            int x = i % w, y = i / w;
            boolean excluded = false;
            for (int r = 0; r + 3 < excludedRects.length && !excluded; r += 4) {
                excluded = x >= excludedRects[r] && x < excludedRects[r + 2] && y >= excludedRects[r + 1] && y < excludedRects[r + 3];
            }
            if (!excluded && ((px[3 * i] & 0xFF) > queryThreshold || (px[3 * i + 1] & 0xFF) > queryThreshold || (px[3 * i + 2] & 0xFF) > queryThreshold)) {
                return i;
            }
        }
        return -1;
    }

    @Override
    public Set<ComputeFileType> getRequiredTargetVariantTypes() {
        return Collections.emptySet();
    }

    @Override
    public PixelMatchScore calculateMatchingScore(@Nonnull ImageArray<?> targetImageArray,
                                                  Map<ComputeFileType, Supplier<ImageArray<?>>> variantImageSuppliers) {
        int size = getQuerySize();
        if (size == 0) {
            return new PixelMatchScore(0, 0, false);
        }
        // a size mismatch comes back as IllegalArgumentException with the reference's message
        long r = CdsHip.scorePair(nativeMaskSet(), 0, ImageArrayBytes.rgb(targetImageArray), targetImageArray.getWidth(), targetImageArray.getHeight());
        int score = (int) r;
        boolean mirrored = (r >>> 32) != 0;
        return new PixelMatchScore(score, (double) score / (double) size, mirrored);
    }

    @Override
    protected void finalize() {
        if (handle != 0) {
            CdsHip.masksetDestroy(handle);
            handle = 0;
        }
    }
}
