"""Host I/O glue: image decode, packed shards, result records."""
import json
import os
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


def test_native_decoder_equals_pillow(tmp_path, fixtures):
    """csrc/cds_imageio.hip against Pillow on every container / compression / pixel type it claims, on real MIP content
    (the reference's fixtures) and on noise (long literal runs, LZW table resets), plus its refusals."""
    PIL = pytest.importorskip("PIL.Image")
    from colormipsearch_amd.capi import CdsError
    rng = np.random.default_rng(8)
    rgb = [fixtures["lm_VT016795"], fixtures["em_LPLC2"], rng.integers(0, 256, size=(97, 131, 3)).astype(np.uint8),
           np.zeros((33, 40, 3), dtype=np.uint8), np.repeat(rng.integers(0, 256, size=(50, 1, 3)), 300, axis=1).astype(np.uint8)]
    k = 0
    for img in rgb:
        for fmt, kw in (("tif", dict(compression="raw")), ("tif", dict(compression="packbits")), ("tif", dict(compression="tiff_lzw")),
                        ("tif", dict(compression="tiff_adobe_deflate")), ("tif", dict(compression="tiff_lzw", tiffinfo={317: 2})),
                        ("png", {}), ("png", dict(compress_level=1))):
            p = tmp_path / f"i{k}.{fmt}"
            k += 1
            PIL.fromarray(img, "RGB").save(p, **kw)
            got = mipio.decode_image(open(p, "rb").read())
            assert got.dtype == np.uint8 and np.array_equal(got, img), (p, kw)
            assert np.array_equal(mipio.read_mip(str(p)), img)
    # gray 8 / 16 (gradient images), RGBA and palette inputs
    g16, g8 = fixtures["grad_BJD"], (fixtures["grad_BJD"] >> 2).astype(np.uint8)
    for arr, fmt, kw in ((g16, "png", {}), (g16, "tif", dict(compression="tiff_lzw")), (g16, "tif", dict(compression="raw")),
                         (g16, "tif", dict(compression="tiff_adobe_deflate", tiffinfo={317: 2})), (g8, "png", {}), (g8, "tif", dict(compression="packbits"))):
        p = tmp_path / f"g{k}.{fmt}"
        k += 1
        PIL.fromarray(arr).save(p, **kw)
        got = mipio.decode_image(open(p, "rb").read())
        assert got.dtype == arr.dtype and np.array_equal(got, arr), (p, kw)
        assert np.array_equal(mipio.read_gradient(str(p)), arr.astype(np.uint16))
    rgba = np.dstack([rgb[2], rng.integers(0, 256, size=rgb[2].shape[:2]).astype(np.uint8)])
    for fmt in ("png", "tif"):
        p = tmp_path / f"a.{fmt}"
        PIL.fromarray(rgba, "RGBA").save(p)
        assert np.array_equal(mipio.decode_image(open(p, "rb").read()), rgb[2])
    pal = PIL.fromarray(rgb[4], "RGB").quantize(32)
    pal.save(tmp_path / "p.png")
    assert np.array_equal(mipio.decode_image(open(tmp_path / "p.png", "rb").read()), np.asarray(pal.convert("RGB")))
    # refusals: not an image, truncated data, JPEG (Pillow takes over in read_mip)
    with pytest.raises(CdsError):
        mipio.decode_image(b"hello world, this is not an image")
    data = open(tmp_path / "i1.tif", "rb").read()
    with pytest.raises(CdsError):
        mipio.decode_image(data[: len(data) // 2])
    PIL.fromarray(rgb[2], "RGB").save(tmp_path / "j.jpg")
    with pytest.raises(CdsError):
        mipio.decode_image(open(tmp_path / "j.jpg", "rb").read())
    assert mipio.read_mip(str(tmp_path / "j.jpg")).shape == rgb[2].shape


def test_native_decoder_survives_corrupt_files(tmp_path):
    """Seeded corruption of valid files (flipped bytes, truncations, garbage tails): the decoder answers with an error or
    with pixels, never with a crash or an out-of-bounds access (the headers are untrusted input)."""
    PIL = pytest.importorskip("PIL.Image")
    from colormipsearch_amd.capi import CdsError
    rng = np.random.default_rng(2026)
    img = rng.integers(0, 256, size=(40, 52, 3)).astype(np.uint8)
    img[10:30, 5:40] = 0
    files = []
    for name, kw in (("a.tif", dict(compression="packbits")), ("b.tif", dict(compression="tiff_lzw")), ("c.tif", dict(compression="raw")),
                     ("d.tif", dict(compression="tiff_adobe_deflate")), ("e.png", {})):
        PIL.fromarray(img, "RGB").save(tmp_path / name, **kw)
        files.append(open(tmp_path / name, "rb").read())
    outcomes = {"ok": 0, "error": 0}
    for data in files:
        for trial in range(250):
            b = bytearray(data)
            kind = trial % 3
            if kind == 0:
                for _ in range(int(rng.integers(1, 6))):
                    b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
            elif kind == 1:
                b = b[: int(rng.integers(0, len(b)))]
            else:
                pos = int(rng.integers(0, len(b)))
                b[pos:pos + 8] = bytes(rng.integers(0, 256, size=8).astype(np.uint8))
            try:
                out = mipio.decode_image(bytes(b))
                assert out.size > 0
                outcomes["ok"] += 1
            except (CdsError, MemoryError, ValueError):
                outcomes["error"] += 1
    assert outcomes["ok"] > 50 and outcomes["error"] > 50, outcomes


def test_native_decoder_on_the_reference_files(fixtures):
    """The reference's own test images (when the reference tree is present), decoded natively == the committed pixel fixtures."""
    import glob, os
    root = "/root/reference/colormipsearch-api/src/test/resources/colormipsearch/api"
    if not os.path.isdir(root):
        pytest.skip("reference tree not present")
    want = {"12191_JRC2018U.tif": "em_12191", "12191_JRC2018U_FL.tif": "em_12191_FL", "1752016801-LPLC2-RT_18U.tif": "em_LPLC2"}
    for f, key in want.items():
        assert np.array_equal(mipio.decode_image(open(os.path.join(root, "cdsearch/ems", f), "rb").read()), fixtures[key]), f
    for f in glob.glob(os.path.join(root, "cdsearch/grad/BJD*.png")):
        assert np.array_equal(mipio.decode_image(open(f, "rb").read()), fixtures["grad_BJD"])
    PIL = pytest.importorskip("PIL.Image")
    for f in glob.glob(os.path.join(root, "imageprocessing/*.tif")) + glob.glob(os.path.join(root, "cdsearch/*/*.tif")):
        with PIL.open(f) as im:
            ref = np.asarray(im.convert("RGB"))
        assert np.array_equal(mipio.read_mip(f), ref), f
        assert np.array_equal(mipio.decode_image(open(f, "rb").read()), ref), f   # (all of them are within the native decoder's range)


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
    doc = json.load(open([p for p in paths if p.endswith("m0.json")][0]))
    assert doc["inputImage"] == {"mipId": "m0", "publishedName": "em0"}
    recs = doc["results"]
    assert [r["matchingPixels"] for r in recs] == [80, 50]
    assert recs[0]["image"]["mipId"] == "t0" and recs[0]["mirrored"] is True
    assert recs[0]["gradientAreaGap"] == 120 and recs[0]["highExpressionArea"] == 30 and recs[0]["normalizedScore"] == 77.5
    assert "gradientAreaGap" not in recs[1] and recs[1]["class"].endswith("CDMatchEntity")
    # a hostile id stays a file inside the output directory
    paths = mipio.write_matches_per_mask(tmp_path / "out2", matches[:1], [{"mipId": "../../x/..m"}], targets)
    assert [os.path.relpath(p, tmp_path / "out2") for p in paths] == ["m.json"]


def _variant_cases(fixtures):
    """(image, gradient | None, zgap | None): the reference's images and the shapes that stress the two codes -- a gradient that is
    no distance ramp (kept raw), runs longer than 255 pixels, a row of pixels that all differ, a frame without any run of black"""
    rng = np.random.default_rng(5)
    H, W = fixtures["lm_BJD"].shape[:2]
    noise16 = rng.integers(0, 65536, (H, W)).astype(np.uint16)
    noise8 = rng.integers(0, 256, (H, W, 3)).astype(np.uint8)
    flat = np.full((H, W, 3), 7, np.uint8)
    steps = (np.arange(W)[None, :] // 3 + np.arange(H)[:, None]).astype(np.uint16)      # a ramp with its first pixel far from 0
    return [(fixtures["lm_BJD"], fixtures["grad_BJD"], fixtures["zgap_BJD"]),
            (fixtures["lm_VT016795"], fixtures["grad_VT016795"], np.ascontiguousarray(fixtures["zgap_BJD"][::-1])),
            (fixtures["lm_VT033614"], noise16, None),
            (fixtures["lm_GMR"], None, None),
            (noise8, steps + 40000, flat),
            (flat, None, noise8)]


def test_variant_shard_codes_on_the_host(tmp_path, fixtures):
    """The packer of the variant shard (csrc/cds_variants.hip; needs no GPU) against a numpy reading of the file format
    (tests/variant_shard_ref.py): every image comes back bit for bit, from arrays and from files; the reference's gradients take the
    two-bit code; a file that was not finished, is cut short or has another size does not open."""
    PIL = pytest.importorskip("PIL.Image")
    from colormipsearch_amd import cds
    from tests.variant_shard_ref import read_shard
    cases = _variant_cases(fixtures)
    H, W = cases[0][0].shape[:2]
    path = tmp_path / "lib.cdsv"
    with cds.VariantShardWriter(path, W, H) as w:
        w.append(np.stack([c[0] for c in cases[:2]]), np.stack([c[1] for c in cases[:2]]), np.stack([c[2] for c in cases[:2]]), threads=2)
        for img, g, z in cases[2:]:
            w.append(img, None if g is None else g[None], None if z is None else z[None])
        tf, gf = tmp_path / "t.tif", tmp_path / "g.png"
        PIL.fromarray(cases[0][0], "RGB").save(tf, compression="packbits")
        PIL.fromarray(cases[0][1]).save(gf)
        w.append_files([str(tf), str(tf)], [str(gf), None], None, threads=2)
        with pytest.raises(cds.CdsError, match="nope"):
            w.append_files([str(tf), str(tmp_path / "nope.tif")])
        with pytest.raises(cds.CdsError, match="nope"):       # ... also when the failing target comes batches after good ones: nothing of the call stays
            w.append_files([str(tf)] * 40 + [str(tmp_path / "nope.tif")], threads=2)
        assert w.count == len(cases) + 2
    shard = cds.VariantShard(path)
    assert (shard.width, shard.height, len(shard)) == (W, H, len(cases) + 2) and shard.file_bytes == os.path.getsize(path)
    assert shard.flags.tolist() == [29, 29, 17, 16, 29, 36, 25, 16]       # 1 has gradient | 8 second differences (2: two bits per pixel) | 4 has zgap | 16 / 32 the MIP / the zgap image as spans
    assert shard.packed_bytes[0] < 250_000                                # (the reference's target: 3.4 MB + 1.4 MB raw)
    w2, h2, got = read_shard(path)
    assert (w2, h2) == (W, H)
    for (flags, rgb, g, z), (ergb, eg, ez) in zip(got, cases + [(cases[0][0], cases[0][1], None), (cases[0][0], None, None)]):
        assert np.array_equal(rgb, ergb)
        assert (g is None) == (eg is None) and (eg is None or np.array_equal(g, eg))
        assert (z is None) == (ez is None) and (ez is None or np.array_equal(z, ez))
    shard.close()
    raw = path.read_bytes()
    cut = tmp_path / "cut.cdsv"
    cut.write_bytes(raw[:len(raw) - 40])
    with pytest.raises(ValueError, match="not a variant shard"):
        cds.VariantShard(cut)
    unfinished = tmp_path / "unfinished.cdsv"
    unfinished.write_bytes(b"\0" * 64 + raw[64:])
    with pytest.raises(ValueError, match="magic"):
        cds.VariantShard(unfinished)
    with pytest.raises(cds.CdsError, match="image size"):
        cds.VariantShardWriter(tmp_path / "x.cdsv", 4, 4)
    empty = tmp_path / "empty.cdsv"
    cds.VariantShardWriter(empty, W, H).finish()
    assert len(cds.VariantShard(empty)) == 0


def test_native_loaders_read_zip_entries(tmp_path, fixtures):
    """The reference's FileData of type zipEntry (an archive + an entry name, NeuronMIPUtils.java:177-199) through the native loaders:
    `archive.zip!entry` -- stored and deflated entries, nested names, the file-name-only lookup the reference falls back to
    (:163-170), an archive with more than 65,535 entries (zip64 end records), errors that name the entry.  Checked through the variant
    shard's packer, which needs no GPU (the GPU loaders share the reader: tests below)."""
    PIL = pytest.importorskip("PIL.Image")
    import io
    from colormipsearch_amd import cds
    from tests.variant_shard_ref import read_shard
    img, grad = fixtures["lm_GMR"], fixtures["grad_BJD"]
    H, W = img.shape[:2]

    def encoded(a, fmt, **kw):
        b = io.BytesIO()
        PIL.fromarray(a, "RGB" if a.ndim == 3 else None).save(b, fmt, **kw)
        return b.getvalue()

    zp = tmp_path / "library.zip"
    with zipfile.ZipFile(zp, "w") as z:
        z.writestr("mips/a.tif", encoded(img, "TIFF", compression="packbits"), compress_type=zipfile.ZIP_DEFLATED)
        z.writestr("mips/deep/b.png", encoded(img[::-1].copy(), "PNG"), compress_type=zipfile.ZIP_STORED)
        z.writestr("gradients/a.png", encoded(grad, "PNG"), compress_type=zipfile.ZIP_DEFLATED)
        z.writestr("notes.txt", b"not an image")
    big = tmp_path / "many.ZIP"                                # (the suffix is matched without regard to case)
    with zipfile.ZipFile(big, "w") as z:
        for i in range(65600):
            z.writestr(f"pad/{i}.txt", b"")
        z.writestr("mips/last.tif", encoded(img, "TIFF", compression="tiff_lzw"), compress_type=zipfile.ZIP_DEFLATED)
    path = tmp_path / "z.cdsv"
    with cds.VariantShardWriter(path, W, H) as w:
        w.append_files([(str(zp), "mips/a.tif"), (str(zp), "mips/deep/b.png"), (str(zp), "elsewhere/b.png"), str(zp) + "!/mips/a.tif", (str(big), "mips/last.tif")],
                       [(str(zp), "gradients/a.png"), None, None, None, None], None, threads=2)
        for bad, what in (((str(zp), "mips/none.tif"), "none.tif"), ((str(zp), "notes.txt"), "notes.txt"), ((str(tmp_path / "absent.zip"), "a.tif"), "absent.zip")):
            with pytest.raises(cds.CdsError, match=what):
                w.append_files([bad])
    _, _, got = read_shard(path)
    assert [f for f, _, _, _ in got] == [25, 16, 16, 16, 16]
    assert np.array_equal(got[0][1], img) and np.array_equal(got[0][2], grad)
    assert np.array_equal(got[1][1], img[::-1]) and np.array_equal(got[2][1], img[::-1])     # found under its full name / by its file name alone
    assert np.array_equal(got[3][1], img) and np.array_equal(got[4][1], img)
    raw = bytearray(zp.read_bytes())
    at = raw.find(b"mips/deep/b.png") + len(b"mips/deep/b.png") + 2000                      # inside the stored entry's data
    raw[at] ^= 0x55
    (tmp_path / "damaged.zip").write_bytes(bytes(raw))
    with cds.VariantShardWriter(tmp_path / "x.cdsv", W, H) as w:
        with pytest.raises(cds.CdsError, match="CRC"):
            w.append_files([(str(tmp_path / "damaged.zip"), "mips/deep/b.png")])
    assert cds.is_native_path((str(zp), "x")) and cds.is_native_path("a.tif") and cds.is_native_path(None)
    assert not cds.is_native_path(("archive.tar", "x")) and not cds.is_native_path(io.BytesIO(b""))


def _random_variant_images(rng, W, H, n):
    """n (image, gradient | None, zgap | None) of W x H: sparse strokes, runs of every length, ramps and non-ramps"""
    out = []
    for i in range(n):
        img = np.zeros((H, W, 3), np.uint8)
        for _ in range(rng.integers(0, 6)):
            y0, x0 = rng.integers(0, H), rng.integers(0, W)
            h, w = rng.integers(1, H + 1), rng.integers(1, min(W, 300) + 1)
            img[y0:y0 + h, x0:x0 + w] = rng.integers(0, 256, 3) if rng.random() < 0.5 else rng.integers(0, 256, (min(h, H - y0), min(w, W - x0), 3))
        kind = i % 4
        if kind == 0:
            grad = None
        elif kind == 1:       # a ramp: steps of -1 / 0 / +1 from a random start, never leaving 0 .. 65535
            steps = rng.integers(-1, 2, (H, W))
            steps[:, 0] = 0
            grad = np.clip(rng.integers(0, 65536, (H, 1)) + np.cumsum(steps, axis=1), 0, 65535).astype(np.uint16)   # (clipping keeps the steps within 1)
        elif kind == 2:       # no ramp
            grad = rng.integers(0, 65536, (H, W)).astype(np.uint16)
        else:                 # flat, at the top of the range
            grad = np.full((H, W), 65535, np.uint16)
        zgap = None if i % 3 == 0 else np.roll(img, 1, axis=1)
        out.append((img, grad, zgap))
    return out


@pytest.mark.parametrize("W,H", [(8, 1), (9, 3), (16, 2), (17, 5), (33, 4), (255, 3), (256, 2), (257, 2), (511, 1), (1210, 3), (2000, 2), (573, 1119)])
def test_variant_shard_codes_at_other_image_sizes(tmp_path, W, H):
    """the packer at the sizes where its words end unevenly (W not a multiple of 16, of 2; one row; the widest image) against the numpy reading"""
    from colormipsearch_amd import cds
    from tests.variant_shard_ref import read_shard
    cases = _random_variant_images(np.random.default_rng(W * 1000 + H), W, H, 8)
    path = tmp_path / "s.cdsv"
    with cds.VariantShardWriter(path, W, H) as w:
        for img, g, z in cases:
            w.append(img, None if g is None else g[None], None if z is None else z[None], threads=1)
    w2, h2, got = read_shard(path)
    assert (w2, h2, len(got)) == (W, H, len(cases))
    for (flags, rgb, g, z), (eimg, eg, ez) in zip(got, cases):
        assert np.array_equal(rgb, eimg)
        assert (g is None) == (eg is None) and (eg is None or np.array_equal(g, eg))
        assert (z is None) == (ez is None) and (ez is None or np.array_equal(z, ez))
    coded = [bool(f & 10) for f, _, _, _ in got]          # two bits per pixel (2) or second differences (8), whichever is shorter
    assert coded[1] and coded[3] and not coded[2]
    assert W * H < 64 or (got[1][0] & 2 and got[3][0] & 8)   # a ramp of random steps has too many exceptions; a flat image has one


@pytest.mark.gpu
@pytest.mark.parametrize("W,H", [(8, 1), (9, 3), (16, 2), (17, 5), (33, 4), (255, 3), (257, 2), (511, 1), (1023, 2), (2000, 3)])
def test_variant_shard_device_decoders_at_other_image_sizes(tmp_path, W, H):
    """the device decoders where a row's last word, the row pitch and the 16-byte stores end unevenly: slots equal the packed images"""
    from colormipsearch_amd import cds
    cases = _random_variant_images(np.random.default_rng(W * 1000 + H), W, H, 12)
    path = tmp_path / "s.cdsv"
    with cds.VariantShardWriter(path, W, H) as w:
        for img, g, z in cases:
            w.append(img, None if g is None else g[None], None if z is None else z[None], threads=1)
    shard = cds.VariantShard(path)
    lib = cds.Library(W, H, len(cases) + 1)
    lib.upload(0, np.full((len(cases) + 1, H, W, 3), 200, np.uint8))                                  # stale content everywhere first
    lib.upload_variants(0, gradients=np.full((len(cases) + 1, H, W), 77, np.uint16), zgaps=np.full((len(cases) + 1, H, W, 3), 9, np.uint8))
    lib.load_variants(1, shard, np.arange(len(cases)))
    imgs = lib.download(0, len(cases) + 1)
    grads, zgaps = lib.download_variants(0, len(cases) + 1)
    assert (imgs[0] == 200).all() and (grads[0] == 77).all() and (zgaps[0] == 9).all()               # the slot in front is untouched
    for k, (img, g, z) in enumerate(cases):
        assert np.array_equal(imgs[k + 1], img), k
        if g is not None:
            assert np.array_equal(grads[k + 1], g), k
        if z is not None:
            assert np.array_equal(zgaps[k + 1], z), k


@pytest.mark.gpu
def test_variant_shard_slots_equal_the_images(tmp_path, fixtures):
    """cds_library_load_variants: targets of a shard decoded ON THE DEVICE into dense slots -- any order, repeats, into a library
    whose slots held other images -- are the images, gradients and zgaps that were packed, bit for bit; the slots' variant flags
    follow the shard's; a damaged blob is refused."""
    from colormipsearch_amd import cds
    cases = _variant_cases(fixtures)
    H, W = cases[0][0].shape[:2]
    path = tmp_path / "lib.cdsv"
    with cds.VariantShardWriter(path, W, H) as w:
        for img, g, z in cases:
            w.append(img, None if g is None else g[None], None if z is None else z[None])
    shard = cds.VariantShard(path)
    lib = cds.Library(W, H, 12)
    lib.generate_synthetic(0xC0FFEE, 0, 0, 12, kind=0, with_variants=True)           # every slot holds something else first
    order = [5, 0, 3, 2, 0, 4, 1, 1, 2]
    flags = lib.load_variants(2, shard, order, threads=3)
    assert flags.tolist() == [int(shard.flags[t]) for t in order]
    imgs = lib.download(2, len(order))
    grads, zgaps = lib.download_variants(2, len(order))
    for k, t in enumerate(order):
        img, g, z = cases[t]
        assert np.array_equal(imgs[k], img), (k, t)
        if g is not None:
            assert np.array_equal(grads[k], g), (k, t)
        if z is not None:
            assert np.array_equal(zgaps[k], z), (k, t)
    # a slot loaded without a gradient scores (-1, -1); with one, the same as the uploaded images
    alg = cds.Shape2DMatchColorDepthSearchAlgorithm(fixtures["em_12191"][None], 20, True, None, cds.label_regions)
    ref = cds.Library(W, H, 2)
    ref.upload(0, np.stack([cases[0][0], cases[1][0]]))
    ref.upload_variants(0, gradients=np.stack([cases[0][1], cases[1][1]]), zgaps=np.stack([cases[0][2], cases[1][2]]))
    g_ref, h_ref, _ = alg.scoreLibraryPairs(ref, np.zeros(2, np.int32), np.array([0, 1]))
    g_got, h_got, _ = alg.scoreLibraryPairs(lib, np.zeros(3, np.int32), np.array([3, 8, 4]))   # targets 0, 1 and 3 (no gradient)
    assert (g_got[:2].tolist(), h_got[:2].tolist()) == (g_ref.tolist(), h_ref.tolist()) and (g_got[2], h_got[2]) == (-1, -1)
    # the MIPs alone (what filling a sparse library reads): the images come, the slots have no variants
    only = cds.Library(W, H, 4)
    only.upload_variants(0, gradients=np.full((4, H, W), 5, np.uint16), zgaps=np.full((4, H, W, 3), 5, np.uint8))
    assert not only.load_variants(1, shard, [4, 0, 5], images_only=True).any()
    got = only.download(1, 3)
    assert np.array_equal(got[0], cases[4][0]) and np.array_equal(got[1], cases[0][0]) and np.array_equal(got[2], cases[5][0])
    g_only, h_only, _ = alg.scoreLibraryPairs(only, np.zeros(2, np.int32), np.array([2, 0]))
    assert (g_only[0], h_only[0]) == (-1, -1) and g_only[1] >= 0                  # slot 2 lost its variants to the load; slot 0 kept the uploaded ones
    bare = cds.Library(W, H, 2)                                                   # (a library that never had variant planes)
    bare.load_variants(0, shard, [1, 2], images_only=True)
    assert np.array_equal(bare.download(0, 2), np.stack([cases[1][0], cases[2][0]]))
    with pytest.raises(ValueError):
        lib.load_variants(0, shard, [len(cases)])
    with pytest.raises(ValueError):
        lib.load_variants(8, shard, order)                                         # does not fit
    shard.close()
    raw = bytearray(path.read_bytes())
    index_off = int(np.frombuffer(raw, "<u8", 1, 24)[0])
    off0 = int(np.frombuffer(raw, "<u8", 1, index_off)[0])
    raw[off0 + 16 + 4 * 300 + 2] ^= 0xFF                                           # a row offset of target 0's image, far off
    bad = tmp_path / "bad.cdsv"
    bad.write_bytes(bytes(raw))
    with pytest.raises(ValueError, match="damaged"):
        lib.load_variants(0, cds.VariantShard(bad), [0])


@pytest.mark.gpu
def test_variant_shard_random_subsets_of_a_thousand_targets(tmp_path, fixtures):
    """At a size where batches, double buffering and the staging limit matter: a shard of 1,024 DIFFERENT targets (the reference's images
    and variants rolled by a target-specific number of columns and rows) and random subsets of it -- any size, any order, repeats --
    loaded into random slot ranges; every loaded slot is compared (image, gradient, zgap) with what was packed."""
    from colormipsearch_amd import cds
    base = [(fixtures["lm_BJD"], fixtures["grad_BJD"], fixtures["zgap_BJD"]), (fixtures["lm_VT016795"], fixtures["grad_VT016795"], None),
            (fixtures["lm_VT033614"], fixtures["grad_VT033614"], fixtures["zgap_BJD"][:, ::-1]), (fixtures["lm_GMR"], None, None)]
    H, W = base[0][0].shape[:2]
    n = 1024

    def target(t):
        img, g, z = base[t % 4]
        dx, dy = (t * 37) % W, (t * 11) % H
        roll = lambda a: None if a is None else np.roll(np.roll(a, dx, axis=1), dy, axis=0)
        return roll(img), roll(g), roll(z)

    path = tmp_path / "thousand.cdsv"
    with cds.VariantShardWriter(path, W, H) as w:
        for t in range(n):
            img, g, z = target(t)
            w.append(img, None if g is None else g[None], None if z is None else z[None], threads=1)
    shard = cds.VariantShard(path)
    assert len(shard) == n and (shard.flags & shard.HAS_GRADIENT != 0).sum() == 768 and (shard.flags & shard.HAS_ZGAP != 0).sum() == 512
    lib = cds.Library(W, H, 700)
    rng = np.random.default_rng(11)
    for size in (1, 700, 513, 64, 300):
        first = int(rng.integers(0, 700 - size + 1))
        picks = rng.integers(0, n, size)
        flags = lib.load_variants(first, shard, picks, threads=int(rng.integers(1, 9)))
        check = rng.choice(size, min(size, 24), replace=False)             # (downloading all 700 slots every time would take the test's time)
        for k in check:
            img, g, z = target(int(picks[k]))
            assert np.array_equal(lib.download(first + int(k), 1)[0], img), (size, k)
            gg, zz = lib.download_variants(first + int(k), 1)
            assert (flags[k] & 1 != 0) == (g is not None) and (flags[k] & 4 != 0) == (z is not None)
            if g is not None:
                assert np.array_equal(gg[0], g), (size, k)
            if z is not None:
                assert np.array_equal(zz[0], z), (size, k)


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


@pytest.mark.gpu
def test_append_files_to_sparse_library(tmp_path, fixtures):
    """cds_sparse_library_append_files: a directory of MIP files (mixed compressions, more than two batches) -> sparse resident
    library -> the reference's golden scores; a broken file stops the ingest with its path in the message."""
    PIL = pytest.importorskip("PIL.Image")
    from colormipsearch_amd import cds
    keys = ("lm_VT033614", "lm_BJD", "lm_VT016795", "lm_GMR")
    comps = ("packbits", "tiff_lzw", "raw", None)
    paths = []
    for i in range(150):
        k = i % 4
        p = tmp_path / (f"t{i}.png" if comps[k] is None else f"t{i}.tif")
        PIL.fromarray(fixtures[keys[k]], "RGB").save(p, **({} if comps[k] is None else dict(compression=comps[k])))
        paths.append(str(p))
    sl = cds.SparseLibrary(1210, 566, 160, 160 * 70000)
    sl.append_files(paths, threads=8)
    assert len(sl) == 150
    alg = cds.PixelMatchColorDepthSearchAlgorithm(fixtures["em_12191"], 20, True, None, 0, False, 20, 0.01, 2, cds.label_regions)
    scores, mir = alg.searchSparseLibrary(sl)
    want = {0: (439, False), 1: (414, False), 2: (426, True)}
    for i in range(150):
        if i % 4 in want:
            assert (int(scores[0, i]), bool(mir[0, i])) == want[i % 4], i
    # round 5: the decode threads build the records on the host (the device's reciprocal table, the device's arithmetic); they are
    # bit-identical to the records the device makes from the same frames, here on the reference's four LM images and on dense-uniform
    # noise (every hue class, ties, thresholds) at a minimum data threshold
    dev = cds.SparseLibrary(1210, 566, 8, 8 * 700000, density_limit=2.0, min_threshold=7)
    rng = np.random.default_rng(3)
    noise = rng.integers(0, 256, size=(2, 566, 1210, 3), dtype=np.uint8)
    noise[0, ::3] = 0
    frames = np.concatenate([np.stack([fixtures[k] for k in keys]), noise])
    dev.append(frames)
    npaths = []
    for i in range(len(frames)):
        p = tmp_path / f"cmp{i}.tif"
        PIL.fromarray(frames[i], "RGB").save(p, compression="packbits")
        npaths.append(str(p))
    host = cds.SparseLibrary(1210, 566, 8, 8 * 700000, density_limit=2.0, min_threshold=7)
    host.append_files(npaths, threads=3)
    assert len(host) == len(dev) == 6 and host.pixels == dev.pixels
    for i in range(6):
        (ra, ca), (rb, cb) = host.image_records(i), dev.image_records(i)
        assert np.array_equal(ra, rb) and np.array_equal(ca, cb), i
    # the same files as entries of a zip archive (the reference's FileData of type zipEntry): the same records; and as the dense
    # slots + gradient of kernel 2's loader
    zp = tmp_path / "lib.zip"
    with zipfile.ZipFile(zp, "w", zipfile.ZIP_DEFLATED) as z:
        for i, q in enumerate(npaths):
            z.write(q, f"mips/{i}.tif", compress_type=zipfile.ZIP_STORED if i % 2 else zipfile.ZIP_DEFLATED)
        gq = tmp_path / "g.png"
        PIL.fromarray(fixtures["grad_BJD"]).save(gq)
        z.write(gq, "grad/1.png")
    zipped = cds.SparseLibrary(1210, 566, 8, 8 * 700000, density_limit=2.0, min_threshold=7)
    zipped.append_files([(str(zp), f"mips/{i}.tif") for i in range(len(npaths))], threads=3)
    assert len(zipped) == 6 and zipped.pixels == dev.pixels
    for i in range(6):
        (ra, ca), (rb, cb) = zipped.image_records(i), dev.image_records(i)
        assert np.array_equal(ra, rb) and np.array_equal(ca, cb), i
    # ... and through a variant shard written from the same archive (decoded on the device, converted from dense staging slots)
    with cds.VariantShardWriter(tmp_path / "lib.cdsv", 1210, 566) as w:
        w.append_files([(str(zp), f"mips/{i}.tif") for i in range(len(npaths))], threads=3)
    sharded = cds.SparseLibrary(1210, 566, 8, 8 * 700000, density_limit=2.0, min_threshold=7)
    sharded.append_from_shard(cds.VariantShard(tmp_path / "lib.cdsv"), batch=4)
    assert len(sharded) == 6 and sharded.pixels == dev.pixels
    for i in range(6):
        (ra, ca), (rb, cb) = sharded.image_records(i), dev.image_records(i)
        assert np.array_equal(ra, rb) and np.array_equal(ca, cb), i
    dl = cds.Library(1210, 566, 2)
    dl.load_files(0, [(str(zp), "mips/1.tif"), (str(zp), "mips/4.tif")], [(str(zp), "grad/1.png"), None])
    assert np.array_equal(dl.download(0, 2), frames[[1, 4]]) and np.array_equal(dl.download_variants(0, 1)[0][0], fixtures["grad_BJD"])
    bad = tmp_path / "bad.tif"
    bad.write_bytes(open(paths[0], "rb").read()[:5000])
    with pytest.raises(cds.CdsError, match="bad.tif"):
        sl.append_files([paths[1], str(bad)])
    with pytest.raises(cds.CdsError, match="size"):
        p2 = tmp_path / "small.png"
        PIL.fromarray(np.zeros((10, 10, 3), dtype=np.uint8), "RGB").save(p2)
        sl.append_files([str(p2)])


@pytest.mark.gpu
def test_sparse_shard_round_trip(tmp_path, fixtures, oracle):
    """A converted library written to a sparse shard file and loaded again (in several batches, into a library that already
    holds images): identical record arrays and identical scores; inconsistent offset tables are refused."""
    from colormipsearch_amd import cds
    keys = ("lm_VT033614", "lm_BJD", "lm_VT016795", "lm_GMR")
    imgs = np.stack([fixtures[k] for k in keys] + [np.zeros_like(fixtures["lm_GMR"]), oracle.synth_image(0xC0FFEE, 5, 0)])
    sl = cds.SparseLibrary(1210, 566, 8, 8 * 70000)
    sl.append(imgs)
    path = tmp_path / "lib.sparse"
    hd = mipio.write_sparse_shard(path, sl, meta=[{"mipId": str(i)} for i in range(len(imgs))])
    assert hd["n"] == len(imgs) and hd["records"] == sl.pixels
    sl2 = cds.SparseLibrary(1210, 566, 12, 12 * 70000)
    sl2.append(imgs[2:3])                                   # not empty: the import appends behind it
    assert mipio.load_sparse_shard(sl2, path, batch=4) == len(imgs)
    assert len(sl2) == len(imgs) + 1 and sl2.pixels == sl.pixels + len(sl.image_records(2)[0])
    for i in range(len(imgs)):
        a, b = sl.image_records(i), sl2.image_records(i + 1)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), i
    alg = cds.PixelMatchColorDepthSearchAlgorithm(fixtures["em_12191"], 20, True, None, 0, False, 20, 0.01, 2, cds.label_regions)
    s1, m1 = alg.searchSparseLibrary(sl)
    s2, m2 = alg.searchSparseLibrary(sl2, 1, len(imgs))
    assert np.array_equal(s1, s2) and np.array_equal(m1, m2) and s1[0, :3].tolist() == [439, 414, 426]
    offs, cells, rec = sl.export(1, 3)
    assert offs[0] == 0 and offs[3] == len(rec) and cells.shape == (3, sl.n_strips, 567)
    bad = cells.copy()
    bad[1, 0, 5] = bad[1, 0, 4] + 10**6                  # not monotone any more
    with pytest.raises(ValueError):
        sl2.import_arrays(offs, bad, rec)
    with pytest.raises(ValueError):
        sl2.import_arrays(offs, cells, rec, n_strips=2, strip_width=608)
    with pytest.raises(ValueError):
        mipio.open_shard(path)                            # (a sparse shard is not a dense shard)
    # corrupt RECORDS behind consistent offset tables are refused too (validated on the device), and nothing is added
    n_before, px_before = len(sl2), sl2.pixels
    first_cell = int(cells[0, 1, 300])                     # some cell of the first exported image with signal pixels
    sig = np.flatnonzero((rec[: int(offs[1])] & 0x7FFFFFFF) != 0)
    k = int(sig[len(sig) // 2])                            # a signal record of image 0
    for what, mutate in (("row flag on a later record", lambda r: r.__setitem__(k + (1 if (r[k] >> 31) else 0), r[k + (1 if (r[k] >> 31) else 0)] | np.uint32(1 << 31))),
                         ("band 0", lambda r: r.__setitem__(k, r[k] & np.uint32(~(63 << 16) & 0xFFFFFFFF))),
                         ("band that second / max do not give", lambda r: r.__setitem__(k, (r[k] & np.uint32(~(63 << 16) & 0xFFFFFFFF)) |
                                                                              np.uint32((lambda b: b + 1 if b % 5 < 4 else b - 1)((int(r[k]) >> 16) & 63) << 16))),
                         ("maximum not above the second channel", lambda r: r.__setitem__(k, (r[k] & np.uint32(0xFFFF0000)) | np.uint32(0x5050))),
                         ("x outside the strip", lambda r: r.__setitem__(k, r[k] | np.uint32(511 << 22))),
                         ("filler inside a cell", lambda r: r.__setitem__(k, np.uint32(1 << 31) if not (r[k] >> 31) else r[k] & np.uint32(0x7FFFFFFF)))):
        bad_rec = rec.copy()
        mutate(bad_rec)
        assert not np.array_equal(bad_rec, rec), what
        with pytest.raises(ValueError, match="inconsistent"):
            sl2.import_arrays(offs, cells, bad_rec)
        assert (len(sl2), sl2.pixels) == (n_before, px_before), what
    sl2.import_arrays(offs, cells, rec)                    # the untouched arrays still import
    assert len(sl2) == n_before + 3
    del first_cell
