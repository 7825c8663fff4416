#!/usr/bin/env python3
"""Regenerate tests/golden/cds_fixtures.npz from the reference's own test resources.

Run in the build container only (needs /root/reference and Pillow):
    python tests/golden/make_fixtures.py

The npz holds *data only*: the decoded pixels of the image fixtures under
colormipsearch-api/src/test/resources/colormipsearch/api/cdsearch/{ems,lms,grad,zgap}
(BSD-3 test resources of the reference) as numpy arrays.  RGB images are stored as
uint8[H,W,3] (R,G,B order = the byte order of the reference's ColorImageArray,
imageprocessing/ColorImageArray.java:6-15), gradient images as uint16[H,W].
The expected scores live in tests/golden/expected.py (copied as *values* from the
reference's JUnit tests, file:line cited there).
"""
import os
import sys
import numpy as np
from PIL import Image

REF = "/root/reference/colormipsearch-api/src/test/resources/colormipsearch/api/cdsearch"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cds_fixtures.npz")

FILES = {
    # key -> (subdir, file name)
    "em_12191": ("ems", "12191_JRC2018U.tif"),
    "em_12191_FL": ("ems", "12191_JRC2018U_FL.tif"),
    "em_LPLC2": ("ems", "1752016801-LPLC2-RT_18U.tif"),
    "lm_BJD": ("lms", "BJD_127B01_AE_01-20171124_64_H6-40x-Brain-JRC2018_Unisex_20x_HR-2483089192251293794-CH2-01_CDM.tif"),
    "lm_GMR": ("lms", "GMR_31G04_AE_01-20190813_66_F3-40x-Brain-JRC2018_Unisex_20x_HR-2704505419467849826-CH2-07_CDM.tif"),
    "lm_VT016795": ("lms", "VT016795_115C08_AE_01-20200221_61_I2-m-CH1_01.tif"),
    "lm_VT033614": ("lms", "VT033614_127B01_AE_01-20171124_64_H6-f-CH2_01.tif"),
    "grad_BJD": ("grad", "BJD_127B01_AE_01-20171124_64_H6-40x-Brain-JRC2018_Unisex_20x_HR-2483089192251293794-CH2-01_CDM.png"),
    "grad_VT016795": ("grad", "VT016795_115C08_AE_01-20200221_61_I2-m-CH1_01.png"),
    "grad_VT033614": ("grad", "VT033614_127B01_AE_01-20171124_64_H6-f-CH2_01.png"),
    "zgap_BJD": ("zgap", "BJD_127B01_AE_01-20171124_64_H6-40x-Brain-JRC2018_Unisex_20x_HR-2483089192251293794-CH2-01_CDM.tif"),
}


def main():
    arrays = {}
    for key, (sub, name) in FILES.items():
        im = Image.open(os.path.join(REF, sub, name))
        if key.startswith("grad_"):
            a = np.asarray(im)
            assert a.ndim == 2, (key, a.shape, im.mode)
            assert a.max() < 65536
            a = a.astype(np.uint16)
        else:
            a = np.asarray(im.convert("RGB"), dtype=np.uint8)
            assert a.ndim == 3 and a.shape[2] == 3
        assert a.shape[:2] == (566, 1210), (key, a.shape)
        arrays[key] = np.ascontiguousarray(a)
        print(key, a.dtype, a.shape, "nonzero px:", int((a.reshape(a.shape[0], a.shape[1], -1).max(axis=2) > 0).sum()))
    np.savez_compressed(OUT, **arrays)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    sys.exit(main())
