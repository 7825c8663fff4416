package org.janelia.colormipsearch.cds;

/**
 * ShapeMatchScore's constructors are package-private in the reference (cds/ShapeMatchScore.java:12-26); the GPU plugin
 * (package cds.gpu) creates its scores through this factory.
 */
public final class GpuScoreAccess {
    private GpuScoreAccess() {
    }

    public static ShapeMatchScore shapeMatchScore(long gradientAreaGap, long highExpressionArea, long maxGradientAreaGap, boolean mirrored) {
        return new ShapeMatchScore(gradientAreaGap, highExpressionArea, maxGradientAreaGap, mirrored);
    }
}
