#!/usr/bin/env python3
"""CPU model of stage-1 filters of the sparse scan on the benchmark's own synthetic data (the oracle's generator: mask = kind 2, seed
0xBADC0DE; targets = kind 0, seed 0xC0FFEE): the share of VISITED target records (signal pixels in the mask's row window) that a
(cell rows x cell columns) x (bins per hue class) map passes on to the evaluation rounds, plain orientation, and the share that
really matches under some shift (the ideal filter).  Same acceptance sets as tools/filter_model.py.  A model for sizing only."""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import filter_model as fm   # noqa: E402
from oracle import cds_oracle as O   # noqa: E402


def main():
    n_t = int(os.environ.get("TARGETS", "24"))
    mask = O.synth_image(0xBADC0DE, 0, 2)
    H, W = mask.shape[:2]
    mcls, mrho, mmax = fm.classify(mask)
    mpix = (mmax > fm.THR) & (mcls > 0)
    mcls = np.where(mpix, mcls, 0)
    ys = np.nonzero(mpix.any(axis=1))[0]
    ylo, yhi = ys[0], ys[-1]
    sets = fm.accept_sets(mcls, mrho)
    targets = [O.synth_image(0xC0FFEE, i, 0) for i in range(n_t)]
    print(f"mask pixels {int(mpix.sum())}, rows {ylo}..{yhi}")
    for B in (5, 8, 10, 16, 64):
        acc = fm.band_bits(mcls, sets[0], sets[1], B) | fm.band_bits(np.where(mpix, sets[2], 0), sets[3], sets[4], B)
        uni = np.zeros_like(acc)
        for dy in (-2, 0, 2):
            for dx in (-2, 0, 2):
                sh = np.zeros_like(acc)
                ysl = slice(max(dy, 0), H + min(dy, 0)); ysr = slice(max(-dy, 0), H + min(-dy, 0))
                xsl = slice(max(dx, 0), W + min(dx, 0)); xsr = slice(max(-dx, 0), W + min(-dx, 0))
                sh[ysl, xsl] = acc[ysr, xsr]
                uni |= sh
        for ch, cw in ((4, 8), (2, 8), (4, 4), (2, 4), (1, 8), (1, 1)):
            Hc, Wc = (H + ch - 1) // ch, (W + cw - 1) // cw
            pad = np.zeros((Hc * ch, Wc * cw, 6 * B), bool); pad[:H, :W] = uni
            cell = pad.reshape(Hc, ch, Wc, cw, 6 * B).any(axis=(1, 3))
            tot = pas = 0
            for t in targets:
                tcls, trho, tmax = fm.classify(t)
                sig = t.max(axis=2) > 0
                vis = sig.copy(); vis[: max(ylo - 2, 0)] = False; vis[yhi + 3:] = False
                ok = vis & (tmax > fm.THR) & (tcls > 0)
                tb = (np.clip(tcls, 1, 6) - 1) * B + np.clip(np.floor(trho * B).astype(int), 0, B - 1)
                yy, xx = np.nonzero(ok)
                tot += int(vis.sum()); pas += int(cell[yy // ch, xx // cw, tb[yy, xx]].sum())
            print(f"bins {B:2d}  cell {ch}x{cw}: pass {pas / tot:7.4f}  ({tot // n_t} visited per target)", flush=True)


if __name__ == "__main__":
    main()
