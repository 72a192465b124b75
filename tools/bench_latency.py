#!/usr/bin/env python3
"""Latency of one operator() call (host image in, host keypoints out) and of small batches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from orb_slam2_amd.extractor import ORBextractor
from orb_slam2_amd.synth import synth_sequence
seq = synth_sequence(376, 1241, 32, seed=5, step=2)
ex = ORBextractor(2000, 1.2, 8, 20, 7)
for _ in range(5):
    ex(seq[0])
for B in (1, 2, 4, 8, 32):
    t0 = time.perf_counter(); K = 30
    for k in range(K):
        if B == 1:
            ex(seq[k % 32])
        else:
            ex.extract_batch(seq[:B])
    dt = (time.perf_counter() - t0) / K
    print("batch %2d host path: %.3f ms per call, %.0f frames/s" % (B, dt * 1e3, B / dt), flush=True)
