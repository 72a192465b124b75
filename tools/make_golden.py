#!/usr/bin/env python3
"""Writes tests/golden/*.npz from the CPU oracle.

The reference ships no fixtures for this path and cannot be built in this image (it needs
OpenCV), so these vectors pin this repository's oracle -- they guard it against
regressions and let GPU tests run without the oracle; they do not pin the oracle to the
reference ("parity unpinned").  Inputs are regenerated from seeds, not stored.

    python tools/make_golden.py
"""
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from orb_slam2_amd.synth import synth_descriptors, synth_image  # noqa: E402

CASES = {
    # name: (rows, cols, seed, nfeatures, nlevels)
    "small_240x320": (240, 320, 5, 500, 8),
    "tum_480x640": (480, 640, 0, 1000, 8),       # benchmark configuration 0
    "kitti_376x1241": (376, 1241, 0, 2000, 8),   # benchmark configuration 1
    "euroc_480x752": (480, 752, 0, 1200, 8),     # benchmark configuration 2 (one eye)
}


def main():
    out_dir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)
    for name, (rows, cols, seed, nf, nl) in CASES.items():
        img = synth_image(rows, cols, seed)
        e = O.Extractor(nf, 1.2, nl, 20, 7)
        kps, desc = e(img)
        lv = {}
        for l in range(nl):
            lv["pyr_crc_%d" % l] = np.uint32(zlib.crc32(e.pyramid_level(l).tobytes()))
            b = e.blurred_level(l)
            lv["blur_crc_%d" % l] = np.uint32(zlib.crc32(b.tobytes()) if b is not None else 0)
            lv["ncand_%d" % l] = np.int32(len(e.candidates(l)))
            lv["cand_crc_%d" % l] = np.uint32(zlib.crc32(e.candidates(l).tobytes()))
        np.savez_compressed(os.path.join(out_dir, name + ".npz"), image_crc=np.uint32(zlib.crc32(img.tobytes())),
                            params=np.array([rows, cols, seed, nf, nl], np.int32),
                            x=kps["x"], y=kps["y"], size=kps["size"], angle=kps["angle"], response=kps["response"],
                            octave=kps["octave"], desc=desc, **lv)
        print(name, len(kps))
    db, q = synth_descriptors(3000, 800, seed=4)
    b, s, i, a = O.knn2_bruteforce(q, db, 0.7)
    np.savez_compressed(os.path.join(out_dir, "knn2_3000x800.npz"), best=b, second=s, idx=i, accept=a)


if __name__ == "__main__":
    main()
