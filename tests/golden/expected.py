"""Known-answer values copied (as data) from the reference's own JUnit tests.

APIT = colormipsearch-api/src/test/java/org/janelia/colormipsearch/
Keys of the image names refer to tests/golden/cds_fixtures.npz (see make_fixtures.py).
"""

# APIT/cds/PixelMatchColorDepthSearchAlgorithmTest.java:33-53 -- direct constructor,
# maskThr 20, mirror, dataThr 20, zTol 0.01, xyShift 2, custom regions (x >= W-260 && y < 90 || x < 330 && y < 100)
PIXEL_MATCH_DIRECT = dict(
    mask="em_LPLC2", target="lm_GMR", mask_thr=20, data_thr=20, ztol=0.01, xy_shift=2, mirror=True,
    rects=[(1210 - 260, 0, 1210, 90), (0, 0, 330, 100)], score=87, mirrored=False)

# APIT/cds/PixelMatchColorDepthSearchAlgorithmTest.java:56-153 -- through the provider:
# mirror, thr 20/20, pixColorFluctuation 1 (zTol = 1/100), xyShift 2, ImageTestUtils.getExcludedRegions()
STD_RECTS = [(1210 - 270, 0, 1210, 90), (0, 0, 330, 100)]   # APIT/ImageTestUtils.java:12-24
PIXEL_MATCH_PROVIDER = [
    # (mask, target, expected score, mirrored)                         test file line
    ("em_12191", "lm_VT033614", 439, False),                           # :72-77
    ("em_12191", "lm_BJD", 414, False),                                # :78-83
    ("em_12191_FL", "lm_VT033614", 515, False),                        # :84-89
    ("em_12191_FL", "lm_VT016795", 483, False),                        # :90-95
    ("em_12191", "lm_VT016795", 426, True),                            # :96-101
]

# APIT/cds/Shape2DMatchColorDepthSearchAlgorithmTest.java:32-60
SHAPE_MASK_PREP = dict(mask="em_12191_FL", mask_size=17340, high_expr_size=70640)

# APIT/cds/Shape2DMatchColorDepthSearchAlgorithmTest.java:63-202 and :205-371
# (mask, target, grad, zgap-or-None, gaps, highExpression, score, mirrored); queryThreshold 20, mirror on.
# zgap None => synthesised as clearRegion -> mask(20) -> unsafeMaxFilter(10) of the target (:171-174)
SHAPE_SCORES = [
    ("em_12191", "lm_VT033614", "grad_VT033614", None, 21365, 731, 21608, False),      # :86-94, :230-239
    ("em_12191", "lm_BJD", "grad_BJD", None, 23359, 523, 23533, False),                # :95-103
    ("em_12191", "lm_BJD", "grad_BJD", "zgap_BJD", 33884, 523, 34058, False),          # :240-249
    ("em_12191", "lm_BJD", "grad_VT033614", None, 23367, 523, 23541, False),           # :250-259
    ("em_12191", "lm_VT016795", "grad_VT016795", None, 40696, 17253, 46447, True),     # :104-112, :260-269
    ("em_12191_FL", "lm_VT033614", "grad_VT033614", None, 65381, 677, 65606, False),   # :113-121, :270-279
    ("em_12191_FL", "lm_VT016795", "grad_VT016795", None, 104449, 16803, 110050, True),# :122-130, :280-289
]

# APIT/cds/GradientAreaGapUtilsTest.java:31-39: (pix, gap, highExpr, maxPix, maxNeg, shapeScore, normalized +-0.1)
NORMALIZED = [
    (636, 156, 1897, 679, 1114361, 788, 46833.58),
    (636, 233, 1644, 679, 1107088, 781, 46833.58),
    (636, 0, 1644, 679, 1114361, 548, 46833.58),
    (795, 123, 93, 875, 1606182, 154, 45428.57),
]
