"""Small seed images for tools/asan_fuzz/run.sh: every TIFF / PNG variant the native decoder accepts."""
import struct

import numpy as np
from PIL import Image

rng = np.random.default_rng(5)
a = np.zeros((48, 64, 3), np.uint8)
a[10:30, 5:50] = rng.integers(0, 255, (20, 45, 3))
a[35:40] = (200, 10, 30)
for comp in ("raw", "packbits", "tiff_lzw", "tiff_adobe_deflate"):
    Image.fromarray(a).save(f"seed_rgb_{comp}.tif", compression=None if comp == "raw" else comp)
    Image.fromarray(a).save(f"seed_strips_{comp}.tif", compression=None if comp == "raw" else comp, tiffinfo={278: 8})
Image.fromarray(a).save("seed_rgb_lzw_pred.tif", compression="tiff_lzw", tiffinfo={317: 2})
Image.fromarray(a).save("seed_rgb.png")
Image.fromarray(a).convert("P").save("seed_pal.png")
Image.fromarray(np.dstack([a, a[..., :1]])).save("seed_rgba.png")
Image.fromarray(a[..., 0]).save("seed_g8.png")
Image.fromarray(a[..., 0]).save("seed_g8_lzw.tif", compression="tiff_lzw")
g16 = a[..., 0].astype(np.uint16) * 257
Image.fromarray(g16).save("seed_g16.png")
Image.fromarray(g16).save("seed_g16.tif")
# a hand-made big-endian uncompressed RGB TIFF (Pillow only writes little-endian)
h, w = 16, 24
px = rng.integers(0, 255, (h, w, 3), dtype=np.uint8).tobytes()
ents = [(256, 3, w), (257, 3, h), (258, 3, None), (259, 3, 1), (262, 3, 2), (273, 4, None), (277, 3, 3), (278, 3, h), (279, 4, len(px))]
ifd = 8
bps = ifd + 2 + 12 * len(ents) + 4
strip = bps + 6
out = b"MM" + struct.pack(">HI", 42, ifd) + struct.pack(">H", len(ents))
for tag, typ, val in ents:
    if tag == 258:
        out += struct.pack(">HHII", tag, typ, 3, bps)
    elif tag == 273:
        out += struct.pack(">HHII", tag, typ, 1, strip)
    elif typ == 3:
        out += struct.pack(">HHIHH", tag, typ, 1, val, 0)
    else:
        out += struct.pack(">HHII", tag, typ, 1, val)
out += struct.pack(">I", 0) + struct.pack(">HHH", 8, 8, 8) + px
open("seed_mm_raw.tif", "wb").write(out)

# a zip archive of two of the seeds (one deflated, one stored) and a variant shard of the 64 x 48 seed image: the archive reader and the
# shard reader see mutated copies of these (fuzz_main.cpp)
import zipfile
with zipfile.ZipFile("seed_archive.zip", "w") as z:
    z.write("seed_rgb_packbits.tif", "mips/a.tif", compress_type=zipfile.ZIP_DEFLATED)
    z.write("seed_rgb.png", "mips/b.png", compress_type=zipfile.ZIP_STORED)
    z.write("seed_g16.png", "grad/a.png", compress_type=zipfile.ZIP_DEFLATED)
