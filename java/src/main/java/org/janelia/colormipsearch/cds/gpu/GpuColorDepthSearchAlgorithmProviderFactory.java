package org.janelia.colormipsearch.cds.gpu;

import org.janelia.colormipsearch.cds.ColorDepthSearchAlgorithm;
import org.janelia.colormipsearch.cds.ColorDepthSearchAlgorithmProvider;
import org.janelia.colormipsearch.cds.ColorDepthSearchParams;
import org.janelia.colormipsearch.cds.PixelMatchScore;
import org.janelia.colormipsearch.cds.ShapeMatchScore;
import org.janelia.colormipsearch.imageprocessing.ImageArray;
import org.janelia.colormipsearch.imageprocessing.ImageRegionDefinition;

/**
 * Same two factory methods, same signatures and parameter handling as cds/ColorDepthSearchAlgorithmProviderFactory (:30-35,
 * :76-79), producing the GPU algorithms.  A caller switches back ends by calling this class instead of the reference's
 * (cmd/ColorDepthSearchCmd.java:217-223, cmd/CalculateGradientScoresCmd.java:466-480).
 */
public final class GpuColorDepthSearchAlgorithmProviderFactory {
    private GpuColorDepthSearchAlgorithmProviderFactory() {
    }

    public static ColorDepthSearchAlgorithmProvider<PixelMatchScore> createPixMatchCDSAlgorithmProvider(
            boolean mirrorMask,
            int targetThreshold,
            double pixColorFluctuation,
            int xyShiftParam,
            ImageRegionDefinition ignoredRegionsProvider) {
        return new ColorDepthSearchAlgorithmProvider<PixelMatchScore>() {
            private final ColorDepthSearchParams defaultCDSParams = new ColorDepthSearchParams()
                    .setParam("mirrorMask", mirrorMask)
                    .setParam("dataThreshold", targetThreshold)
                    .setParam("pixColorFluctuation", pixColorFluctuation)
                    .setParam("xyShift", xyShiftParam);

            @Override
            public ColorDepthSearchParams getDefaultCDSParams() {
                return defaultCDSParams;
            }

            @Override
            public ColorDepthSearchAlgorithm<PixelMatchScore> createColorDepthSearchAlgorithm(ImageArray<?> queryImageArray, int queryThreshold,
                                                                                              int queryBorderSize, ColorDepthSearchParams cdsParams) {
                Double fluctuation = cdsParams.getDoubleParam("pixColorFluctuation", pixColorFluctuation);
                double zTolerance = fluctuation == null ? 0. : fluctuation / 100;
                int xyShift = cdsParams.getIntParam("xyShift", xyShiftParam);
                if ((xyShift & 0x1) == 1) {
                    throw new IllegalArgumentException("XY shift parameter must be an even number.");
                }
                return new GpuPixelMatchColorDepthSearchAlgorithm(
                        queryImageArray, queryThreshold, cdsParams.getBoolParam("mirrorMask", mirrorMask),
                        cdsParams.getIntParam("dataThreshold", targetThreshold), zTolerance, xyShift,
                        RegionRects.fromRegion(ignoredRegionsProvider, queryImageArray));   // (the pixel-match scorer ignores queryBorderSize, as the reference does)
            }
        };
    }

    public static ColorDepthSearchAlgorithmProvider<ShapeMatchScore> createShapeMatchCDSAlgorithmProvider(
            boolean mirrorMask,
            ImageArray<?> roiMaskImageArray,
            ImageRegionDefinition excludedRegions) {
        return new ColorDepthSearchAlgorithmProvider<ShapeMatchScore>() {
            private final ColorDepthSearchParams defaultCDSParams = new ColorDepthSearchParams().setParam("mirrorMask", mirrorMask);

            @Override
            public ColorDepthSearchParams getDefaultCDSParams() {
                return defaultCDSParams;
            }

            @Override
            public ColorDepthSearchAlgorithm<ShapeMatchScore> createColorDepthSearchAlgorithm(ImageArray<?> queryImageArray, int queryThreshold,
                                                                                             int queryBorderSize, ColorDepthSearchParams cdsParams) {
                byte[] roi = null;
                if (roiMaskImageArray != null) {                 // non-black ROI pixels = inside (ColorTransformation.mask semantics)
                    int n = roiMaskImageArray.getPixelCount();
                    roi = new byte[n];
                    for (int i = 0; i < n; i++) {
                        roi[i] = (byte) ((roiMaskImageArray.get(i) & 0xFFFFFF) != 0 ? 1 : 0);
                    }
                }
                return new GpuShape2DMatchColorDepthSearchAlgorithm(
                        queryImageArray, queryThreshold, queryBorderSize, cdsParams.getBoolParam("mirrorMask", mirrorMask), roi,
                        RegionRects.fromRegion(excludedRegions, queryImageArray));
            }
        };
    }
}
