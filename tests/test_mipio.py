"""Host I/O glue: image decode, packed shards, result records."""
import json
import zipfile

import numpy as np
import pytest

from colormipsearch_amd import mipio


def test_image_files_and_zip_entries_round_trip(tmp_path, fixtures):
    PIL = pytest.importorskip("PIL.Image")
    img = fixtures["lm_GMR"]
    grad = fixtures["grad_BJD"]
    for name, kw in (("a.tif", dict(compression="packbits")), ("b.tif", dict(compression="tiff_lzw")), ("c.tif", {}), ("d.png", {})):
        p = tmp_path / name
        PIL.fromarray(img, "RGB").save(p, **kw)
        assert np.array_equal(mipio.read_mip(str(p)), img)
    gp = tmp_path / "g.png"
    PIL.fromarray(grad).save(gp)
    assert np.array_equal(mipio.read_gradient(str(gp)), grad)
    zp = tmp_path / "mips.zip"
    with zipfile.ZipFile(zp, "w") as z:
        z.write(tmp_path / "a.tif", "dir/a.tif")
    assert np.array_equal(mipio.read_mip((str(zp), "dir/a.tif")), img)


def test_shard_round_trip(tmp_path, fixtures):
    imgs = np.stack([fixtures[k] for k in ("lm_GMR", "lm_BJD", "em_LPLC2")])
    path = tmp_path / "lib.shard"
    mipio.write_shard(path, imgs, meta=[{"mipId": str(i)} for i in range(3)])
    arr, meta = mipio.open_shard(path)
    assert arr.shape == imgs.shape and np.array_equal(np.asarray(arr), imgs) and meta[2]["mipId"] == "2"
    with open(tmp_path / "bad", "wb") as f:
        f.write(b'{"magic": "x"}\n')
    with pytest.raises(ValueError):
        mipio.open_shard(tmp_path / "bad")


def test_match_records_per_mask(tmp_path):
    from colormipsearch_amd.processor import CDMatch
    matches = [CDMatch(0, 1, 50, 0.05, False), CDMatch(0, 0, 80, 0.08, True, 120, 30, 77.5), CDMatch(1, 1, 5, 0.01, False)]
    masks = [{"mipId": "m0", "publishedName": "em0"}, {"mipId": "m1"}]
    targets = [{"mipId": "t0", "publishedName": "lineA"}, {"mipId": "t1", "publishedName": "lineB"}]
    paths = mipio.write_matches_per_mask(tmp_path / "out", matches, masks, targets)
    assert sorted(p.split("/")[-1] for p in paths) == ["m0.json", "m1.json"]
    recs = json.load(open([p for p in paths if p.endswith("m0.json")][0]))
    assert [r["matchingPixels"] for r in recs] == [80, 50]
    assert recs[0]["image"]["mipId"] == "t0" and recs[0]["mirrored"] is True
    assert recs[0]["gradientAreaGap"] == 120 and recs[0]["highExpressionArea"] == 30 and recs[0]["normalizedScore"] == 77.5
    assert "gradientAreaGap" not in recs[1] and recs[1]["class"].endswith("CDMatchEntity")


@pytest.mark.gpu
def test_shard_upload_and_search(tmp_path, fixtures, oracle):
    from colormipsearch_amd import cds
    from tests.golden import expected as E
    keys = ("lm_VT033614", "lm_BJD", "lm_VT016795")
    path = tmp_path / "lms.shard"
    mipio.write_shard(path, np.stack([fixtures[k] for k in keys]))
    arr, _ = mipio.open_shard(path)
    lib = cds.Library(1210, 566, len(keys))
    assert mipio.upload_shard(lib, arr, batch=2) == len(keys)
    alg = cds.PixelMatchColorDepthSearchAlgorithm(fixtures["em_12191"], 20, True, None, 0, False, 20, 0.01, 2, cds.label_regions)
    scores, mir = alg.searchLibrary(lib)
    assert scores[0].tolist() == [439, 414, 426] and mir[0].tolist() == [False, False, True]   # the reference's goldens
