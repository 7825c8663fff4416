package org.janelia.colormipsearch.cds.gpu;

import java.util.EnumSet;
import java.util.Map;
import java.util.Set;
import java.util.function.Supplier;

import javax.annotation.Nonnull;

import org.janelia.colormipsearch.cds.ColorDepthSearchAlgorithm;
import org.janelia.colormipsearch.cds.GpuScoreAccess;
import org.janelia.colormipsearch.cds.ShapeMatchScore;
import org.janelia.colormipsearch.imageprocessing.ImageArray;
import org.janelia.colormipsearch.imageprocessing.ImageArrayBytes;
import org.janelia.colormipsearch.model.ComputeFileType;

/**
 * Drop-in for cds/Shape2DMatchColorDepthSearchAlgorithm (GradientAreaGap scorer of one query, prepared as
 * ColorDepthSearchAlgorithmProviderFactory.createShapeMatchCDSAlgorithmProvider does, queryBorderSize included).  The mask
 * preparation (two circular max filters, query mask, high-expression mask) runs on the GPU when the native set is created.
 */
public class GpuShape2DMatchColorDepthSearchAlgorithm implements ColorDepthSearchAlgorithm<ShapeMatchScore> {
    private static final long serialVersionUID = 1L;

    private final ImageArray<?> queryImage;
    private final int queryThreshold;
    private final int queryBorderSize;
    private final boolean mirrorQuery;
    private final byte[] roiMask;          // w*h bytes, non-zero = inside, or null
    private final int[] excludedRects;
    private transient long handle;

    public GpuShape2DMatchColorDepthSearchAlgorithm(ImageArray<?> queryImage, int queryThreshold, int queryBorderSize, boolean mirrorQuery,
                                                    byte[] roiMask, int[] excludedRectsXYXY) {
        this.queryImage = queryImage;
        this.queryThreshold = queryThreshold;
        this.queryBorderSize = queryBorderSize;
        this.mirrorQuery = mirrorQuery;
        this.roiMask = roiMask;
        this.excludedRects = excludedRectsXYXY;
    }

    public synchronized long nativeShapeSet() {
        if (handle == 0) {
            handle = CdsHip.shapesetCreate(GpuContext.get(), 1, ImageArrayBytes.rgb(queryImage), queryImage.getWidth(), queryImage.getHeight(),
                    queryThreshold, queryBorderSize, mirrorQuery, roiMask, excludedRects);
        }
        return handle;
    }

    @Override
    public ImageArray<?> getQueryImage() {
        return queryImage;
    }

    private boolean visited(int x, int y) {   // what a fold over the bordered, region-cleared query image visits
        int w = queryImage.getWidth(), h = queryImage.getHeight(), b = queryBorderSize;
        if (x < b || x >= w - b || y < b || y >= h - b) {
            return false;
        }
        for (int r = 0; r + 3 < excludedRects.length; r += 4) {
            if (x >= excludedRects[r] && x < excludedRects[r + 2] && y >= excludedRects[r + 1] && y < excludedRects[r + 3]) {
                return false;
            }
        }
        return true;
    }

    private boolean signal(byte[] px, int i) {
        return (px[3 * i] & 0xFF) > queryThreshold || (px[3 * i + 1] & 0xFF) > queryThreshold || (px[3 * i + 2] & 0xFF) > queryThreshold;
    }

    @Override
    public int getQuerySize() {
        byte[] px = ImageArrayBytes.rgb(queryImage);
        int w = queryImage.getWidth(), n = queryImage.getPixelCount(), size = 0;
        for (int i = 0; i < n; i++) {
            if (visited(i % w, i / w) && signal(px, i)) {
                size++;
            }
        }
        return size;
    }

    @Override
    public int getQueryFirstPixelIndex() {
        byte[] px = ImageArrayBytes.rgb(queryImage);
        int w = queryImage.getWidth(), n = queryImage.getPixelCount();
        for (int i = 0; i < n; i++) {
            if (visited(i % w, i / w) && signal(px, i)) {
                return i;
            }
        }
        return -1;
    }

    @Override
    public int getQueryLastPixelIndex() {
        byte[] px = ImageArrayBytes.rgb(queryImage);
        int w = queryImage.getWidth(), n = queryImage.getPixelCount();
        for (int i = n - 1; i >= 0; i--) {
            if (visited(i % w, i / w) && signal(px, i)) {
                return i;
            }
        }
        return -1;
    }

    @Override
    public Set<ComputeFileType> getRequiredTargetVariantTypes() {
        return EnumSet.of(ComputeFileType.GradientImage, ComputeFileType.ZGapImage);
    }

    @Override
    public ShapeMatchScore calculateMatchingScore(@Nonnull ImageArray<?> targetImageArray,
                                                  Map<ComputeFileType, Supplier<ImageArray<?>>> variantImageSuppliers) {
        ImageArray<?> gradient = variant(variantImageSuppliers, ComputeFileType.GradientImage);
        if (gradient == null) {
            return GpuScoreAccess.shapeMatchScore(-1, -1, -1, false);      // "not computed", as the reference signals it
        }
        int w = targetImageArray.getWidth(), h = targetImageArray.getHeight();
        byte[] target = ImageArrayBytes.rgb(targetImageArray);
        ImageArray<?> zgap = variant(variantImageSuppliers, ComputeFileType.ZGapImage);
        byte[] zgapRgb;
        if (zgap != null) {
            zgapRgb = ImageArrayBytes.rgb(zgap);
        } else {                                                           // no ZGapImage file: dilate the thresholded target (radius 10)
            zgapRgb = new byte[target.length];
            CdsHip.makeZgap(GpuContext.get(), target, w, h, queryThreshold, 10, excludedRects, zgapRgb);
        }
        long[] out = new long[3];
        CdsHip.shapeScorePair(nativeShapeSet(), 0, target, ImageArrayBytes.gray16(gradient), zgapRgb, w, h, out);
        return GpuScoreAccess.shapeMatchScore(out[0], out[1], -1, out[2] != 0);
    }

    private static ImageArray<?> variant(Map<ComputeFileType, Supplier<ImageArray<?>>> suppliers, ComputeFileType type) {
        Supplier<ImageArray<?>> s = suppliers == null ? null : suppliers.get(type);
        return s == null ? null : s.get();
    }

    @Override
    protected void finalize() {
        if (handle != 0) {
            CdsHip.shapesetDestroy(handle);
            handle = 0;
        }
    }
}
