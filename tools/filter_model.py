#!/usr/bin/env python3
"""CPU model of stage-1 filters of the sparse scan on the reference's own images (tests/golden/cds_fixtures.npz): for every
(EM mask, LM target) pair, the share of VISITED target records (signal pixels in the mask's row window) that a filter passes
on to the evaluation rounds, per orientation.  Filters: the cell x band map of the kernel (4 x 8-pixel cells, 5 bands per
class), an exact per-PIXEL union of the nine neighbours' bands with B bins per class, and the ideal filter (some shift
really matches).  Acceptance sets follow AbstractColorDepthSearchAlgorithm.java:195-381 (as colormipsearch_amd/csrc/cds_host.hip
restates them) in double arithmetic -- a model for sizing, not the product path."""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ZTOL = float(os.environ.get("ZTOL", "0.01"))
THR = 20
BrBg, BgGb, GbGr, GrRg, RgRb = 0.354862745, 0.996078431, 0.505882353, 0.996078431, 0.505882353
NB = {1: [(2, True, 0.44, 0.54, BrBg)], 2: [(1, True, 0.54, 0.44, BrBg), (3, False, 0.8, 0.8, BgGb)],
      3: [(2, False, 0.8, 0.8, BgGb), (4, True, 0.7, 0.7, GbGr)], 4: [(3, True, 0.7, 0.7, GbGr), (5, False, 0.8, 0.8, GrRg)],
      5: [(4, False, 0.8, 0.8, GrRg), (6, True, 0.7, 0.7, RgRb)], 6: [(5, True, 0.7, 0.7, RgRb)]}


def classify(img):
    r, g, b = (img[..., i].astype(np.int32) for i in range(3))
    cls = np.zeros(r.shape, np.int32)
    num = np.zeros(r.shape, np.float64)
    den = np.ones(r.shape, np.float64)
    bm = (b > r) & (b > g)
    gm = (g > b) & (g > r) & ~bm
    rm = (r > b) & (r > g) & ~bm & ~gm
    for m, c_hi, c_lo, hi, lo, mx in ((bm, 1, 2, r, g, b), (gm, 3, 4, b, r, g), (rm, 5, 6, g, b, r)):
        first = m & (hi > lo)
        cls[first] = c_hi; num[first] = hi[first]; den[first] = mx[first]
        second = m & ~first
        cls[second] = c_lo; num[second] = lo[second]; den[second] = mx[second]
    return cls, num / den, np.maximum(np.maximum(r, g), b)


def accept_sets(cls, rho):
    """per pixel: own-class [lo, hi] and (neighbour class, lo, hi) in ratio space (empty = lo > hi)"""
    own_lo = np.where(rho > 0, np.maximum(rho - ZTOL, 1e-9), 2.0)
    own_hi = np.where(rho > 0, rho + ZTOL, -1.0)
    ncls = np.zeros(cls.shape, np.int32)
    nlo = np.full(cls.shape, 2.0)
    nhi = np.full(cls.shape, -1.0)
    for c1, rules in NB.items():
        for (c2, low, la, lb, c) in rules:
            if low:
                m = (cls == c1) & (rho < la)
                hi = np.minimum(ZTOL + 2 * c - rho, lb - 1e-12)
                ok = m & (hi >= 0)
                ncls[ok] = c2; nlo[ok] = 0.0; nhi[ok] = hi[ok]
            else:
                m = (cls == c1) & (rho > la)
                lo = np.maximum(2 * c - rho - ZTOL, lb + 1e-12)
                ok = m & (lo <= 1.0)
                ncls[ok] = c2; nlo[ok] = lo[ok]; nhi[ok] = 1.0
    return own_lo, own_hi, ncls, nlo, nhi


def band_bits(cls, lo, hi, B):
    """bit sets (as python-int object arrays are slow: use a bool array [.., 6*B]) of bins touched by [lo, hi] of class cls"""
    out = np.zeros(cls.shape + (6 * B,), bool)
    valid = (cls > 0) & (lo <= hi)
    b0 = np.clip(np.floor(lo * B).astype(int), 0, B - 1)
    b1 = np.clip(np.floor(hi * B).astype(int), 0, B - 1)
    ys, xs = np.nonzero(valid)
    for y, x in zip(ys, xs):
        base = (cls[y, x] - 1) * B
        out[y, x, base + b0[y, x]: base + b1[y, x] + 1] = True
    return out


def main():
    fx = np.load(os.path.join(ROOT, "tests", "golden", "cds_fixtures.npz"))
    lms = [k for k in fx.keys() if k.startswith("lm_")]
    ems = [k for k in fx.keys() if k.startswith("em_")]
    Bs = [int(b) for b in os.environ.get("BINS", "5,10,16,32").split(",")]
    H, W = fx[lms[0]].shape[:2]
    lab = np.zeros((H, W), bool)  # label regions are excluded from masks by the caller; ignored here
    for mk in ems:
        mcls, mrho, mmax = classify(fx[mk])
        mpix = (mmax > THR) & (mcls > 0)
        mcls = np.where(mpix, mcls, 0)
        ys = np.nonzero(mpix.any(axis=1))[0]
        ylo, yhi = ys[0], ys[-1]
        sets = accept_sets(mcls, mrho)
        for B in Bs:
            acc = band_bits(mcls, sets[0], sets[1], B) | band_bits(np.where(mpix, sets[2], 0), sets[3], sets[4], B)
            # union over the nine shifts: target pixel t is probed against mask pixels t - s
            uni = np.zeros_like(acc)
            for dy in (-2, 0, 2):
                for dx in (-2, 0, 2):
                    sh = np.zeros_like(acc)
                    ysl = slice(max(dy, 0), H + min(dy, 0)); ysr = slice(max(-dy, 0), H + min(-dy, 0))
                    xsl = slice(max(dx, 0), W + min(dx, 0)); xsr = slice(max(-dx, 0), W + min(-dx, 0))
                    sh[ysl, xsl] = acc[ysr, xsr]
                    uni |= sh
            # the kernel's cell map: OR over 4 x 8-pixel cells (only reported for B = 5)
            cell = None
            if B == 5:
                Hc, Wc = (H + 3) // 4, (W + 7) // 8
                pad = np.zeros((Hc * 4, Wc * 8, 6 * B), bool); pad[:H, :W] = uni
                cell = pad.reshape(Hc, 4, Wc, 8, 6 * B).any(axis=(1, 3))
            for orient, U in (("plain", uni), ("mirror", uni[:, ::-1])):
                tot = pas = pcell = 0
                for lk in lms:
                    tcls, trho, tmax = classify(fx[lk])
                    sig = (fx[lk].max(axis=2) > 0)
                    vis = sig.copy(); vis[: max(ylo - 2, 0)] = False; vis[yhi + 3:] = False
                    ok = vis & (tmax > THR) & (tcls > 0)
                    tb = (np.clip(tcls, 1, 6) - 1) * B + np.clip(np.floor(trho * B).astype(int), 0, B - 1)
                    yy, xx = np.nonzero(ok)
                    p = U[yy, xx, tb[yy, xx]]
                    tot += int(vis.sum()); pas += int(p.sum())
                    if cell is not None:
                        C = cell if orient == "plain" else cell  # (the mirrored cell map is built from the mirrored mask)
                        if orient == "mirror":
                            padm = np.zeros((Hc * 4, Wc * 8, 6 * B), bool); padm[:H, :W] = U
                            C = padm.reshape(Hc, 4, Wc, 8, 6 * B).any(axis=(1, 3))
                        pcell += int(C[yy // 4, xx // 8, tb[yy, xx]].sum())
                line = f"{mk:12s} {orient:6s} B={B:2d}: visited {tot:7d}  per-pixel pass {pas / tot:6.3f}"
                if cell is not None:
                    line += f"   cell(4x8) pass {pcell / tot:6.3f}"
                print(line, flush=True)


if __name__ == "__main__":
    main()
