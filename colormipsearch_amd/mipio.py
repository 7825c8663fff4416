"""Data formats on either side of the hot path -- host I/O glue only.

* reading MIPs: the reference decodes TIFF (PackBits / LZW / raw) and PNG/JPG files or zip entries into ImageArrays
  (API/imageprocessing/ImageArrayUtils.java:98-121, API/mips/NeuronMIPUtils.java:177-199); here the native decoder of
  libcds_hip.so (csrc/cds_imageio.hip: strip TIFF raw / PackBits / LZW / Deflate, PNG) does it -- pixel-identical to Pillow
  and to ImageJ on the reference's fixtures -- with Pillow as the reader of everything else; whole directories of MIP files
  go into a sparse resident library through `cds.SparseLibrary.append_files` (threaded decode overlapped with the upload);
* a packed shard format: one header line (JSON) + the raw uint8[n,H,W,3] pixels, written once and memory-mapped for
  large sequential uploads into a `cds.Library` -- for real libraries decode + PCIe dominate, not the scan;
* match results as JSON records with the field names of the reference's CDMatchEntity files
  (PERSIST/dataio/fs/JSONNeuronMatchesWriter.java; API/model/CDMatchEntity.java:13-20), one file per mask, sorted by
  descending matchingPixels (tools cmd/ColorDepthSearchCmd.java:403-409).
"""
import io
import json
import os
import re
import zipfile

import numpy as np

SHARD_MAGIC = "cds-mip-shard-v1"


def _pil():
    try:
        from PIL import Image
    except ImportError as e:  # pragma: no cover
        raise ImportError("reading MIP image files needs Pillow") from e
    return Image


def _open(file_data):
    """file_data: a path, or (zip_path, entry_name) like the reference's FileData zipEntry."""
    if isinstance(file_data, (tuple, list)):
        zpath, entry = file_data
        with zipfile.ZipFile(zpath) as z:
            return io.BytesIO(z.read(entry))
    return file_data


def _bytes(file_data):
    src = _open(file_data)
    if isinstance(src, io.BytesIO):
        return src.getvalue()
    with open(src, "rb") as f:
        return f.read()


def decode_image(data):
    """Native decode (libcds_hip.so: cds_image_probe / cds_image_decode -- strip TIFF raw / PackBits / LZW / Deflate, PNG):
    bytes -> uint8[H,W,3], uint8[H,W] or uint16[H,W].  Raises ValueError for anything the native decoder refuses."""
    import ctypes
    from . import capi
    lib = capi.load()
    buf = np.frombuffer(data, dtype=np.uint8)
    info = capi.ImageInfo()
    capi.check(lib.cds_image_probe(capi.as_u8p(buf), buf.size, ctypes.byref(info)))
    shape = (info.height, info.width, 3) if info.channels == 3 else (info.height, info.width)
    out = np.empty(shape, dtype=np.uint16 if info.bits == 16 else np.uint8)
    capi.check(lib.cds_image_decode(capi.as_u8p(buf), buf.size, out.ctypes.data_as(ctypes.c_void_p), out.nbytes))
    return out


def read_mip(file_data, native=True):
    """-> uint8[H,W,3] (R,G,B), the byte layout of the reference's ColorImageArray.  The native decoder handles the formats
    the reference's libraries come in; other files (JPEG, tiled TIFF, ...) go through Pillow."""
    if native:
        from .capi import CdsError
        try:
            a = decode_image(_bytes(file_data))
            return a if a.ndim == 3 else np.ascontiguousarray(np.repeat(a.astype(np.uint8)[:, :, None], 3, axis=2))
        except (CdsError, ImportError, OSError):   # (OSError: the .so exists but its ROCm runtime does not load)
            pass
    with _pil().open(_open(file_data)) as im:
        return np.ascontiguousarray(np.asarray(im.convert("RGB"), dtype=np.uint8))


def read_gradient(file_data, native=True):
    """-> uint16[H,W]: GRAY16 as is, GRAY8 widened (ImageArray.get semantics)."""
    a = None
    if native:
        from .capi import CdsError
        try:
            a = decode_image(_bytes(file_data))
        except (CdsError, ImportError, OSError):   # (OSError: the .so exists but its ROCm runtime does not load)
            a = None
    if a is None:
        with _pil().open(_open(file_data)) as im:
            a = np.asarray(im)
    if a.ndim != 2:
        raise ValueError(f"gradient image must be grayscale, got shape {a.shape}")
    return np.ascontiguousarray(a.astype(np.uint16))


# ---- packed shards ------------------------------------------------------------------------------------
def write_shard(path, images, meta=None):
    images = np.ascontiguousarray(images, dtype=np.uint8)
    if images.ndim != 4 or images.shape[3] != 3:
        raise ValueError(f"expected uint8[n,H,W,3], got {images.shape}")
    n, h, w, _ = images.shape
    header = json.dumps({"magic": SHARD_MAGIC, "n": n, "height": h, "width": w, "meta": meta or []}).encode() + b"\n"
    with open(path, "wb") as f:
        f.write(header)
        f.write(images.tobytes())


def open_shard(path):
    """-> (memory-mapped uint8[n,H,W,3], meta list). The pixels are not read until they are uploaded."""
    with open(path, "rb") as f:
        header = f.readline()
    hd = json.loads(header)
    if hd.get("magic") != SHARD_MAGIC:
        raise ValueError(f"{path} is not a {SHARD_MAGIC} file")
    arr = np.memmap(path, dtype=np.uint8, mode="r", offset=len(header), shape=(hd["n"], hd["height"], hd["width"], 3))
    return arr, hd.get("meta", [])


def upload_shard(library, shard_pixels, first_slot=0, batch=256):
    """Stream a (memory-mapped) shard into a resident library in batches."""
    n = shard_pixels.shape[0]
    for a in range(0, n, batch):
        b = min(n, a + batch)
        library.upload(first_slot + a, np.ascontiguousarray(shard_pixels[a:b]))
    return n


# ---- sparse shards: a converted library on disk -------------------------------------------------------------
SPARSE_MAGIC = "cds-sparse-shard-v2"   # v2: records carry the hue band and a strip-relative column (cds_internal.hpp); v1 files are refused


def write_sparse_shard(path, sparse_library, first=0, count=None, meta=None):
    """Images [first, first+count) of a cds.SparseLibrary as one file: a JSON header line, then image offsets (u64), cell
    offsets (u32) and records (u32), each 64-byte aligned -- ~82 KB per synthetic target instead of 2 MB of pixels."""
    offs, cells, rec = sparse_library.export(first, count)
    hd = {"magic": SPARSE_MAGIC, "n": len(offs) - 1, "width": sparse_library.width, "height": sparse_library.height,
          "n_strips": sparse_library.n_strips, "strip_width": sparse_library.strip_width, "records": int(rec.size),
          "min_threshold": sparse_library.min_threshold, "meta": meta or []}
    header = json.dumps(hd).encode()
    header += b" " * (-(len(header) + 1) % 64) + b"\n"
    with open(path, "wb") as f:
        f.write(header)
        for a in (offs, cells, rec):
            f.write(a.tobytes())
            f.write(b"\0" * (-a.nbytes % 64))
    return hd


def open_sparse_shard(path):
    """-> (header dict, image_offsets, cell_offsets, records) with the three arrays memory-mapped."""
    with open(path, "rb") as f:
        header = f.readline()
    hd = json.loads(header)
    if hd.get("magic") != SPARSE_MAGIC:
        raise ValueError(f"{path} is not a {SPARSE_MAGIC} file")
    n, cells = hd["n"], hd["n_strips"] * (hd["height"] + 1)
    pos = len(header)
    offs = np.memmap(path, dtype=np.uint64, mode="r", offset=pos, shape=(n + 1,))
    pos += (n + 1) * 8 + (-(n + 1) * 8) % 64
    co = np.memmap(path, dtype=np.uint32, mode="r", offset=pos, shape=(n, hd["n_strips"], hd["height"] + 1))
    pos += n * cells * 4 + (-(n * cells * 4)) % 64
    rec = np.memmap(path, dtype=np.uint32, mode="r", offset=pos, shape=(hd["records"],)) if hd["records"] else np.zeros(0, dtype=np.uint32)
    return hd, offs, co, rec


def load_sparse_shard(sparse_library, path, batch=65536):
    """Append a sparse shard file to a resident cds.SparseLibrary, `batch` images per upload.  Returns the image count."""
    hd, offs, co, rec = open_sparse_shard(path)
    if (hd["width"], hd["height"]) != (sparse_library.width, sparse_library.height):
        raise ValueError("image size of the shard differs from the library's")
    if int(hd.get("min_threshold", 0)) > sparse_library.min_threshold:    # (the file does not hold the pixels in between)
        raise ValueError(f"the shard was written with minimum data threshold {hd['min_threshold']}, the library's is "
                         f"{sparse_library.min_threshold}: raise it first (SparseLibrary.set_min_threshold)")
    for a in range(0, hd["n"], batch):
        b = min(hd["n"], a + batch)
        o = np.asarray(offs[a:b + 1])
        sparse_library.import_arrays(o - o[0], co[a:b], rec[int(o[0]):int(o[-1])], hd["n_strips"], hd["strip_width"],
                                     min_data_threshold=int(hd.get("min_threshold", 0)))   # (the C ABI checks it too: any caller's import is covered)
    return hd["n"]


# ---- results ------------------------------------------------------------------------------------------
def write_matches_per_mask(out_dir, matches, mask_infos, target_infos, ordering=None):
    """One `<mask mipId>.json` per mask in the reference's result-file layout (`{"inputImage": ..., "results": [...]}`,
    results by descending matchingPixels unless `ordering` says otherwise): see colormipsearch_amd/results.py.
    Returns the written paths."""
    from . import results
    docs = results.match_entities(matches, mask_infos, target_infos)
    w = results.JSONNeuronMatchesWriter(matchOrdering=ordering or results.by_matching_pixels_desc, perMasksOutputDir=str(out_dir))
    w.write(docs)
    return list(w.written)
