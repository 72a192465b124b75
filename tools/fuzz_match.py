#!/usr/bin/env python3
"""Randomised parity sweep of the matchers against the oracle (run on a GPU box): the
parity tests of tests/test_match_gpu.py with random sizes, thresholds and seeds.
    python tools/fuzz_match.py [n_rounds] [seed]"""
import os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pytest
from oracle import oracle as O
import test_match_gpu as T

O.lib()
n_rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
t0 = time.time()


def run(name, fn, *args):
    global bad
    try:
        fn(O, *args)
    except AssertionError:
        # the tests also assert that the scene is rich enough ("> 100 matches"); only parity counts here
        tb = traceback.format_exc()
        if "scene too poor" not in tb:
            bad += 1
            print("MISMATCH %s%r\n%s" % (name, args, tb.splitlines()[-2]), flush=True)


for r in range(n_rounds):
    s = int(rng.integers(0, 1 << 30))
    ri = lambda a, b: int(rng.integers(a, b))
    run("frames", T.test_search_by_projection_frames, s, ri(50, 3000), ri(50, 3000), ri(0, 40), bool(ri(0, 2)))
    run("corner", T.test_search_by_projection_frames_in_one_crowded_corner, s, ri(200, 3000), ri(200, 3000), float(rng.uniform(8, 80)))
    run("mappoints", T.test_search_by_projection_mappoints, s, ri(50, 3000), ri(50, 2500), float(rng.choice([1.0, 3.0, 5.0])), float(rng.uniform(0.5, 0.95)))
    run("triang", T.test_search_for_triangulation, s, ri(100, 2500), ri(100, 2500), bool(ri(0, 2)))
    run("windowed", T.test_match_windowed_vs_oracle, s, ri(100, 3000), ri(100, 3500), int(rng.choice([50, 100])), bool(ri(0, 2)))
    run("bow", T.test_search_by_bow_both_overloads, s, ri(100, 2500), ri(100, 2500), float(rng.uniform(0.6, 0.95)), bool(ri(0, 2)), ri(2, 300))
    run("init", T.test_search_for_initialization, s, ri(200, 3000), ri(10, 300), float(rng.uniform(0.7, 0.99)), bool(ri(0, 2)))
    run("knn", T.test_knn2_vs_oracle, ri(1, 3000), ri(1, 3000))
    # frames beyond the on-chip grid (the second form of the kernel), every few rounds
    if r % 3 == 0:
        import test_operating_point_gpu as P
        run("frames_large", lambda O_, *a: P._projection_case(O_, *a), s, ri(500, 6000), ri(9000, 25000), ri(0, 200), bool(ri(0, 2)))
        run("init_large", P.test_search_for_initialization_at_the_shipped_size, s, ri(7400, 16000), ri(30, 200))
        run("triang_large", T.test_search_for_triangulation, s, ri(3000, 9000), ri(8200, 14000), bool(ri(0, 2)))
        run("windowed_large", T.test_match_windowed_vs_oracle, s, ri(8200, 14000), ri(1000, 8000), int(rng.choice([50, 100])), bool(ri(0, 2)))
        run("bow_large", T.test_search_by_bow_both_overloads, s, ri(8200, 12000), ri(8200, 12000), float(rng.uniform(0.6, 0.95)), bool(ri(0, 2)), ri(50, 600))
        run("mappoints_large", T.test_search_by_projection_mappoints, s, ri(1000, 9000), ri(8200, 14000), float(rng.choice([1.0, 3.0, 5.0])), float(rng.uniform(0.5, 0.95)))
    print("round %d done, %d mismatches, %.0fs" % (r, bad, time.time() - t0), flush=True)
print("done: %d rounds, %d mismatches" % (n_rounds, bad))
sys.exit(1 if bad else 0)
