"""Pins the CPU oracle (oracle/cds_oracle.c) to every known-answer value the reference's own
JUnit tests hold for the hot path, on the reference's own fixture images."""
import numpy as np
import pytest

from tests.golden import expected as E


def test_pixel_match_direct_constructor(fixtures, oracle):
    # PixelMatchColorDepthSearchAlgorithmTest.pixelMatchScore
    e = E.PIXEL_MATCH_DIRECT
    q = oracle.PixelMatchQuery(fixtures[e["mask"]], e["mask_thr"], e["data_thr"], e["ztol"], e["xy_shift"],
                               e["mirror"], e["rects"])
    score, mirrored, counts = q.score(fixtures[e["target"]])
    assert (score, mirrored) == (e["score"], e["mirrored"])
    assert counts[:9].max() == score


@pytest.mark.parametrize("mask,target,score,mirrored", E.PIXEL_MATCH_PROVIDER)
def test_pixel_match_provider(fixtures, oracle, mask, target, score, mirrored):
    # PixelMatchColorDepthSearchAlgorithmTest.multiplePixelScores: pixColorFluctuation 1 -> zTol 1/100
    q = oracle.PixelMatchQuery(fixtures[mask], 20, 20, 1 / 100, 2, True, E.STD_RECTS)
    s, m, _ = q.score(fixtures[target])
    assert (s, m) == (score, mirrored)


def test_mask_sizes_appendix_b(fixtures, oracle):
    # SURVEY.md appendix B (P20 column) -- query sizes with the standard label regions
    for key, P in (("em_12191", 10299), ("em_12191_FL", 17340), ("em_LPLC2", 1897)):
        pos, bbox = oracle.mask_positions(fixtures[key], 20, E.STD_RECTS)
        assert len(pos) == P
    pos, bbox = oracle.mask_positions(fixtures["em_12191_FL"], 20, E.STD_RECTS)
    assert list(bbox) == [316, 165, 894, 335]


def test_shape_mask_preparation(fixtures, oracle):
    # Shape2DMatchColorDepthSearchAlgorithmTest.overExpressesMaskExpression
    e = E.SHAPE_MASK_PREP
    sq = oracle.ShapeQuery(fixtures[e["mask"]], 20, True, E.STD_RECTS)
    assert int(sq.QM.sum()) == e["mask_size"]
    assert int(sq.QH.sum()) == e["high_expr_size"]


@pytest.fixture(scope="module")
def shape_queries(fixtures, oracle):
    return {k: oracle.ShapeQuery(fixtures[k], 20, True, E.STD_RECTS) for k in ("em_12191", "em_12191_FL")}


@pytest.mark.parametrize("mask,target,grad,zgap,gaps,high,score,mirrored", E.SHAPE_SCORES)
def test_shape_scores(fixtures, oracle, shape_queries, mask, target, grad, zgap, gaps, high, score, mirrored):
    # Shape2DMatchColorDepthSearchAlgorithmTest.computeShapeScoreUsing{AlgorithmProvider,DirectAlgorithmConstructor}
    z = fixtures[zgap] if zgap else oracle.make_zgap(fixtures[target], 20, 10, E.STD_RECTS)
    got = shape_queries[mask].score(fixtures[target], fixtures[grad], z)
    assert got == (gaps, high, score, mirrored)


def test_shape_missing_variants(fixtures, oracle, shape_queries):
    # Shape2DMatchColorDepthSearchAlgorithm.java:155-158
    got = shape_queries["em_12191"].score(fixtures["lm_BJD"], None, None)
    assert got[:2] == (-1, -1) and got[3] is False


@pytest.mark.parametrize("pix,gap,high,max_pix,max_neg,shape,norm", E.NORMALIZED)
def test_normalized_scores(oracle, pix, gap, high, max_pix, max_neg, shape, norm):
    # GradientAreaGapUtilsTest.scoreCalculator
    s = oracle.shape_score_value(gap, high)
    assert s == shape
    assert abs(oracle.normalized_score(pix, s, max_pix, max_neg) - norm) <= 0.1


def test_slice_numbers_of_lut_entries(oracle):
    # every LUT colour is its own slice (exact-ratio hit, GradientAreaGapUtils.java:185-187), except where two
    # entries of one sextant share a ratio (the first wins)
    lut = oracle.depth_lut()
    got = np.array([oracle.slice_number(c) for c in lut])
    assert got[0] == 1 and got[255] == 256
    assert (np.abs(got - np.arange(1, 257)) <= 1).all()
    assert oracle.slice_number((0, 0, 0)) == 0


def test_pixel_gap_quirks(oracle):
    assert oracle.pixel_gap((10, 10, 200), (10, 10, 200)) == 0.0      # BG (r<=g), same ratio
    assert oracle.pixel_gap((200, 200, 10), (200, 200, 10)) == 10000  # tie at the top: no class
    assert oracle.pixel_gap((0, 0, 200), (0, 0, 100)) == 10000        # ratio 0: same class never matches
    # adjacent classes may give a NEGATIVE gap (no abs), which passes <= zTol
    g = oracle.pixel_gap((50, 0, 255), (0, 50, 255))                  # BR 0.196 vs BG 0.196
    assert g < 0


def test_oracle_query_border_restatement(fixtures, oracle):
    """queryBorderSize in the oracle (no reference test exercises it: a restatement of LImage.fold's loop bounds,
    LImage.java:89-113, and of which images inherit the borders, LImageUtils.java:27-49 / LImageImpl.java:76-104): border 0
    is the pinned behaviour; with a border the masks are empty in the frame, unchanged inside, and a gap under the frame
    no longer counts."""
    mask = fixtures["em_12191"]
    q0 = oracle.ShapeQuery(mask, 20, True, E.STD_RECTS)
    qb = oracle.ShapeQuery(mask, 20, True, E.STD_RECTS, border=200)
    assert np.array_equal(q0.Q, qb.Q)
    inner = np.zeros(q0.QM.shape, dtype=bool)
    inner[200:-200, 200:-200] = True
    assert np.array_equal(qb.QM, q0.QM * inner) and np.array_equal(qb.QH, q0.QH * inner)
    assert qb.QM.sum() < q0.QM.sum()
    t, g = fixtures["lm_VT033614"], fixtures["grad_VT033614"]
    z = oracle.make_zgap(t, 20, 10, E.STD_RECTS)
    g0, h0, _, _ = q0.score(t, g, z)
    gb, hb, _, _ = qb.score(t, g, z)
    assert 0 <= gb < g0 and 0 <= hb <= h0
