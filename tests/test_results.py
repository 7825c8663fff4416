"""Result files: the JSON layout of PERSIST/dataio/fs/JSONNeuronMatchesWriter.java, checked
the way the reference's own PERSIST test JSONReadWriteTest.java:81-116 checks it, on that test's data file
(tests/golden/testcdsmatches.json = PERSIST/src/test/resources/cdsmatches/testcdsmatches.json, 18 EM->LM matches)."""
import json
import os

import numpy as np
import pytest

from colormipsearch_amd import results
from colormipsearch_amd.processor import CDMatch

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "testcdsmatches.json")


def _test_matches():
    ms = results.read_matches_list(GOLDEN)
    ms.sort(key=results.by_matching_pixels_desc)     # JSONReadWriteTest.readTestMatches :131
    return ms


def _same_entity(a, b):
    """AbstractNeuronEntity.equals (:283-296): entity id, MIP id, source and input image."""
    cf = lambda e, k: (e.get("computeFiles") or {}).get(k)
    return (a.get("entityId") == b.get("entityId") and a["mipId"] == b["mipId"] and
            cf(a, "SourceColorDepthImage") == cf(b, "SourceColorDepthImage") and
            cf(a, "InputColorDepthImage") == cf(b, "InputColorDepthImage"))


def test_read_write_cds_results(tmp_path):
    matches = _test_matches()
    assert len(matches) == 18
    em2lm, lm2em = str(tmp_path / "em2lm"), str(tmp_path / "lm2em")
    w = results.JSONNeuronMatchesWriter(matchOrdering=results.by_matching_pixels_desc, perMasksOutputDir=em2lm, perMatchesOutputDir=lm2em)
    n = w.write(matches)
    mask_ids = {m["maskImage"]["mipId"] for m in matches}
    target_ids = {m["image"]["mipId"] for m in matches}
    assert n == len(mask_ids) + len(target_ids)
    assert {f[:-5] for f in os.listdir(em2lm)} == mask_ids          # checkResultFiles :118-126
    assert {f[:-5] for f in os.listdir(lm2em)} == target_ids
    rd = results.JSONNeuronMatchesReader()
    for mid in mask_ids:
        back = rd.readMatchesByMask(em2lm, [mid])
        want = [m for m in matches if m["maskImage"]["mipId"] == mid]   # already sorted by -matchingPixels
        assert len(back) == len(want) > 0
        for b, x in zip(back, want):
            assert _same_entity(b["maskImage"], x["maskImage"]) and _same_entity(b["image"], x["image"])
            assert b["mirrored"] == x["mirrored"] and b["matchComputeFiles"] == {}
            assert b["matchingPixels"] == x["matchingPixels"]
            assert b["matchingPixelsRatio"] == x["matchingPixelsRatio"] and b["normalizedScore"] == x["normalizedScore"]
            # the three compute files travelled through matchComputeFiles and came back
            for k in ("InputColorDepthImage", "GradientImage", "ZGapImage", "SourceColorDepthImage"):
                assert b["maskImage"]["computeFiles"].get(k) == x["maskImage"]["computeFiles"].get(k)
    for tid in target_ids:
        back = rd.readMatchesByTarget(lm2em, [tid])
        want = [m for m in matches if m["image"]["mipId"] == tid]
        assert len(back) == len(want) > 0
        for b, x in zip(back, want):
            assert _same_entity(b["maskImage"], x["maskImage"]) and _same_entity(b["image"], x["image"])
            assert b["matchingPixels"] == x["matchingPixels"]


def test_file_layout(tmp_path):
    matches = _test_matches()
    out = str(tmp_path / "bymask")
    results.JSONNeuronMatchesWriter(perMasksOutputDir=out).write(matches)
    mid = matches[0]["maskImage"]["mipId"]
    doc = json.load(open(os.path.join(out, mid + ".json")))
    assert set(doc) == {"inputImage", "results"}                       # GroupedMatchedEntities.java:12-33
    key = doc["inputImage"]
    assert key["mipId"] == mid and key["class"].endswith("EMNeuronEntity")
    # the key keeps SourceColorDepthImage only (MatchEntitiesGrouping.java:73-80)
    assert set(key["computeFiles"]) == {"SourceColorDepthImage"}
    px = [r["matchingPixels"] for r in doc["results"]]
    assert px == sorted(px, reverse=True)
    for r in doc["results"]:
        assert "maskImage" not in r and r["image"]["class"].endswith("LMNeuronEntity")
        assert set(r["matchComputeFiles"]) == {"MaskColorDepthImage", "MaskGradientImage", "MaskZGapImage"}
        assert "files" not in r["image"]                              # not a property of the entity
        assert isinstance(r["mirrored"], bool)


def test_non_empty_inclusion_and_float32(tmp_path):
    mask = {"class": "org.janelia.colormipsearch.model.EMNeuronEntity", "mipId": "m1", "publishedName": "",
            "computeFiles": {"InputColorDepthImage": "/a/m1.tif"}, "tags": [], "neuronTerms": None}
    tgt = {"class": "org.janelia.colormipsearch.model.LMNeuronEntity", "mipId": "t1", "slideCode": "20200101_1_A1",
           "computeFiles": {"InputColorDepthImage": "/a/t1.tif"}}
    recs = [CDMatch(0, 0, 241, float(np.float32(241 / 6198)), False),
            CDMatch(0, 0, 300, 0.1, True, gradientAreaGap=1234, highExpressionArea=56, normalizedScore=float(np.float32(12.5)))]
    docs = results.match_entities(recs, [mask], [tgt])
    out = str(tmp_path / "o")
    results.JSONNeuronMatchesWriter(perMasksOutputDir=out, prettyPrint=False).write(docs)
    text = open(os.path.join(out, "m1.json")).read()
    doc = json.loads(text)
    assert "publishedName" not in doc["inputImage"] and "tags" not in doc["inputImage"] and "neuronTerms" not in doc["inputImage"]
    assert "computeFiles" not in doc["inputImage"]                   # emptied by the grouping, so NON_EMPTY drops it
    r0, r1 = doc["results"]                                            # 300 before 241
    assert r0["matchingPixels"] == 300 and r0["mirrored"] is True and r0["gradientAreaGap"] == 1234 and r0["highExpressionArea"] == 56
    assert r0["normalizedScore"] == 12.5
    assert "gradientAreaGap" not in r1 and "normalizedScore" not in r1
    assert '"matchingPixelsRatio":0.03888351' in text                  # the float32's shortest decimal, as in the reference's file
    assert r0["matchComputeFiles"] == {"MaskColorDepthImage": "/a/m1.tif"}


def test_grouping_is_by_first_mask_of_the_key_and_stable():
    a1 = {"mipId": "A", "computeFiles": {"InputColorDepthImage": "A_1.tif"}}
    a2 = {"mipId": "A", "computeFiles": {"InputColorDepthImage": "A_2.tif"}}
    t = lambda i: {"mipId": "T%d" % i}
    ms = [{"maskImage": a1, "image": t(0), "matchingPixels": 5}, {"maskImage": a2, "image": t(1), "matchingPixels": 9},
          {"maskImage": a1, "image": None, "matchingPixels": 99}, {"maskImage": a2, "image": t(2), "matchingPixels": 5}]
    g = results.group_by_mask(ms, ordering=results.by_matching_pixels_desc)
    assert len(g) == 1
    assert [r["image"]["mipId"] for r in g[0]["results"]] == ["T1", "T0", "T2"]      # no-image match dropped; ties keep input order
    assert [r["matchComputeFiles"]["MaskColorDepthImage"] for r in g[0]["results"]] == ["A_2.tif", "A_1.tif", "A_2.tif"]
    back = results.expand_results_by_mask(g[0])
    assert [m["maskImage"]["computeFiles"]["InputColorDepthImage"] for m in back] == ["A_2.tif", "A_1.tif", "A_2.tif"]


def test_compute_files_in_both_json_forms():
    """FileData as the reference writes and reads it (FileDataSerializer.java:13-26, FileDataDeserializer.java:14-36): a string, a zipEntry
    object, a file object with a directory and a name; on the reference's own test matches the three variant lists come out."""
    from colormipsearch_amd import cds
    assert results.file_data("/a/b.tif") == "/a/b.tif" and results.file_data("  ") is None and results.file_data(None) is None
    assert results.file_data({"dataType": "zipEntry", "fileName": "/lib/mips.zip", "entryName": "cdm/x.tif"}) == ("/lib/mips.zip", "cdm/x.tif")
    assert results.file_data({"dataType": "file", "fileName": "/lib/cdm", "entryName": "x.tif"}) == "/lib/cdm/x.tif"
    assert results.file_data({"fileName": "/lib/x.tif"}) == "/lib/x.tif" and results.file_data({"dataType": "zipEntry", "fileName": "/lib/mips.zip"}) is None
    assert cds.native_path(results.file_data({"dataType": "zipEntry", "fileName": "/lib/mips.zip", "entryName": "cdm/x.tif"})) == b"/lib/mips.zip!cdm/x.tif"
    with pytest.raises(ValueError):
        results.file_data(17)
    matches = results.read_matches_list(GOLDEN)
    targets, grads, zgaps = results.variant_files([m["image"] for m in matches])
    assert len(targets) == len(matches) and all(t and t.endswith(".tif") for t in targets)
    assert zgaps[0].endswith("Split2_segmented_RGB20px/SS01491-20140729_31_B4-m_20x_CH1_03.tif") and all(cds.is_native_path(x) for x in targets + grads + zgaps)
    assert results.compute_file(matches[0]["maskImage"], "GradientImage").endswith("grad/1002507170-PFL1-RT_18U_FL.png")
    assert results.compute_file(None, "GradientImage") is None and results.compute_file({}, "ZGapImage") is None
