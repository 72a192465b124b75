#!/usr/bin/env python3
"""Latency of one operator() call (host image in, host keypoints out) and the rate of the
host-buffer batch entry point (orb_extract_batch called with preallocated outputs)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from orb_slam2_amd._lib import check, ptr
from orb_slam2_amd._lib import KP_DTYPE
from orb_slam2_amd.extractor import ORBextractor
from orb_slam2_amd.synth import synth_sequence
NMAX = 128
seq = np.ascontiguousarray(synth_sequence(376, 1241, 32, seed=5, step=2))
seq = np.concatenate([seq] * (NMAX // 32))
rows, cols = seq.shape[1:]
ex = ORBextractor(2000, 1.2, 8, 20, 7)
for _ in range(5):
    ex(seq[0])
cap = ex.max_keypoints(rows, cols)
kps = np.zeros((NMAX, cap), KP_DTYPE)
desc = np.zeros((NMAX, cap, 32), np.uint8)
nout = np.zeros(NMAX, np.int32)
for B in (1, 2, 4, 8, 32, 64, 128):
    K = 30 if B <= 32 else 10
    for it in range(K + 2):
        if it == 2:
            t0 = time.perf_counter()
        check(ex._lib.orb_extract_batch(ex._h, B, ptr(seq), rows, cols, cols, rows * cols, cap,
                                        ptr(kps), ptr(desc), ptr(nout)))
    dt = (time.perf_counter() - t0) / K
    print("batch %3d host path: %.3f ms per call, %.0f frames/s (%d keypoints in frame 0)"
          % (B, dt * 1e3, B / dt, nout[0]), flush=True)
