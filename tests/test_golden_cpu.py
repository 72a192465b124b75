"""The oracle reproduces the committed golden vectors (tests/golden, written by
tools/make_golden.py from this oracle: a regression pin, not a reference pin)."""
import os
import zlib

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ["small_240x320", "tum_480x640"]


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_golden(oracle, name):
    from orb_slam2_amd.synth import synth_image
    g = np.load(os.path.join(GOLD, name + ".npz"))
    rows, cols, seed, nf, nl = (int(v) for v in g["params"])
    img = synth_image(rows, cols, seed)
    assert zlib.crc32(img.tobytes()) == int(g["image_crc"]), "synthetic input generator changed"
    e = oracle.Extractor(nf, 1.2, nl, 20, 7)
    kps, desc = e(img)
    for f in ("x", "y", "size", "angle", "response", "octave"):
        assert np.array_equal(kps[f], g[f]), f
    assert np.array_equal(desc, g["desc"])
    for l in range(nl):
        assert zlib.crc32(e.pyramid_level(l).tobytes()) == int(g["pyr_crc_%d" % l])
        assert len(e.candidates(l)) == int(g["ncand_%d" % l])


def test_oracle_knn_matches_golden(oracle):
    from orb_slam2_amd.synth import synth_descriptors
    g = np.load(os.path.join(GOLD, "knn2_3000x800.npz"))
    db, q = synth_descriptors(3000, 800, seed=4)
    b, s, i, a = oracle.knn2_bruteforce(q, db, 0.7)
    assert np.array_equal(b, g["best"]) and np.array_equal(s, g["second"])
    assert np.array_equal(i, g["idx"]) and np.array_equal(a, g["accept"])
