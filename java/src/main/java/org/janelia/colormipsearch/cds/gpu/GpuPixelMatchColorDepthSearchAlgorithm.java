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
    private transient int[] strip;

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
        return stripBounds()[0];
    }

    @Override
    public int getQueryLastPixelIndex() {
        return stripBounds()[1];
    }

    /**
     * The two values the reference returns, quirks included (the accessors are informational: nothing in the reference reads them):
     * the base class MULTIPLIES row, width and column of the query's bounding box -- miny * width * minx and maxy * width * maxx
     * with maxx / maxy one past the last pixel (cds/AbstractColorDepthSearchAlgorithm.java:76-84, :96-126) -- and the pixel-match
     * class lowers / raises those by the FIRST / LAST entry of every shifted (and mirrored) position list, an entry shifted
     * out of the image being -1 (cds/PixelMatchColorDepthSearchAlgorithm.java:59-100, :113-158).  Same result as
     * colormipsearch_amd/cds.py (PixelMatchColorDepthSearchAlgorithm._strip).
     */
    private synchronized int[] stripBounds() {
        if (strip != null) {
            return strip;
        }
        byte[] px = ImageArrayBytes.rgb(queryImage);
        int w = queryImage.getWidth(), h = queryImage.getHeight(), n = queryImage.getPixelCount();
        int minx = w, miny = h, maxx = 0, maxy = 0, firstPos = -1, lastPos = -1;
        for (int i = 0; i < n; i++) {
            int x = i % w, y = i / w;
            boolean excluded = false;
            for (int r = 0; r + 3 < excludedRects.length && !excluded; r += 4) {
                excluded = x >= excludedRects[r] && x < excludedRects[r + 2] && y >= excludedRects[r + 1] && y < excludedRects[r + 3];
            }
            if (excluded || !((px[3 * i] & 0xFF) > queryThreshold || (px[3 * i + 1] & 0xFF) > queryThreshold || (px[3 * i + 2] & 0xFF) > queryThreshold)) {
                continue;
            }
            if (firstPos < 0) {
                firstPos = i;
            }
            lastPos = i;
            minx = Math.min(minx, x);
            maxx = Math.max(maxx, x + 1);
            miny = Math.min(miny, y);
            maxy = Math.max(maxy, y + 1);
        }
        int first = miny * w * minx, last = maxy * w * maxx;
        if (firstPos >= 0) {
            // every position list keeps the order of the query positions, so its first / last entry is the shifted (mirrored)
            // first / last query position, or -1 when the shift leaves the image
            for (int dx = -xyShift; dx <= xyShift; dx += Math.max(xyShift, 1)) {
                for (int dy = -xyShift; dy <= xyShift; dy += Math.max(xyShift, 1)) {
                    int a = shifted(firstPos, dx, dy, w, h), b = shifted(lastPos, dx, dy, w, h);
                    first = Math.min(first, a);
                    last = Math.max(last, b);
                    if (mirrorQuery) {
                        first = Math.min(first, a == -1 ? -1 : a + (w - 1) - 2 * (a % w));
                        last = Math.max(last, b == -1 ? -1 : b + (w - 1) - 2 * (b % w));
                    }
                }
            }
        }
        strip = new int[]{first, last};
        return strip;
    }

    private static int shifted(int pos, int dx, int dy, int w, int h) {
        int x = pos % w + dx, y = pos / w + dy;
        return x >= 0 && x < w && y >= 0 && y < h ? y * w + x : -1;
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
