#!/usr/bin/env python3
"""Extraction at the edges of what the library takes (run on a GPU box): widest and narrowest images, one level,
one feature, feature counts beyond the quad-tree's slots; every accepted case is compared with the oracle.
    python tools/check_extreme_shapes.py"""
import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
from oracle import oracle as O
from orb_slam2_amd import ORBextractor
from orb_slam2_amd._lib import OrbHipError
from orb_slam2_amd.synth import synth_image
bad = 0
for (rows, cols, nf, nlev, sf) in [(200, 4127, 3000, 4, 1.2), (4127, 2100, 5000, 8, 1.2), (62, 100, 50, 1, 1.2), (63, 124, 80, 1, 1.2),
                                   (376, 1241, 20000, 8, 1.2), (480, 640, 1, 8, 1.2), (480, 640, 7, 3, 1.44), (100, 199, 300, 2, 1.1),
                                   (2160, 3840, 30000, 8, 1.2)]:
    img = synth_image(rows, cols, rows + cols)
    try:
        k, d = ORBextractor(nf, sf, nlev, 20, 7)(img)
    except OrbHipError as e:
        print(rows, cols, nf, nlev, "refused:", str(e)[:80]); continue
    rk, rd = O.Extractor(nf, sf, nlev, 20, 7)(img)
    ok = len(k) == len(rk) and all(np.array_equal(k[f], rk[f]) for f in ("x", "y", "octave", "response", "size"))
    ok = ok and (len(k) == 0 or (np.abs(k["angle"] - rk["angle"]).max() < 1e-4 * 180 / np.pi and (d == rd).all(1).mean() >= 0.999))
    print(rows, cols, nf, nlev, sf, "ok" if ok else "MISMATCH", len(k), len(rk), flush=True)
    bad += not ok
print("bad", bad)
