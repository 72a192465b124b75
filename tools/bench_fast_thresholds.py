#!/usr/bin/env python3
"""FAST stage time against the two thresholds (KITTI shape, 128 device-resident frames):
how much of the stage the second, lower-threshold pass over corner-free cells costs.
    python tools/bench_fast_thresholds.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from orb_slam2_amd import _lib
from orb_slam2_amd.extractor import ORBextractor
from orb_slam2_amd.synth import synth_sequence

rows, cols, nf, B = 376, 1241, 2000, 128
seq = synth_sequence(rows, cols, 8, seed=5, step=2)
seq = np.concatenate([seq] * (B // len(seq)), 0)
d_img = torch.from_numpy(seq).cuda()
for ini, mn in [(20, 7), (20, 20), (7, 7), (40, 40), (255, 255)]:
    ex = ORBextractor(nf, 1.2, 8, ini, mn, device=0)
    max_kp = ex.max_keypoints(rows, cols)
    d_kps = torch.zeros((B, max_kp, 6), dtype=torch.float32, device="cuda")
    d_desc = torch.zeros((B, max_kp, 32), dtype=torch.uint8, device="cuda")
    d_n = torch.zeros(B, dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    run = lambda: ex.extract_batch_device(d_img.data_ptr(), B, rows, cols, d_kps.data_ptr(), d_desc.data_ptr(), d_n.data_ptr(), max_kp, stream=st)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    L = _lib.lib()
    L.orb_extractor_set_profiling(ex._h, 1)
    for _ in range(10):
        run()
    torch.cuda.synchronize()
    ms = np.zeros(5, np.float32)
    _lib.check(L.orb_extractor_stage_times(ex._h, ms.ctypes.data))
    ncand = sum(len(ex.fast_candidates(l, 0)) for l in range(8))
    print("iniTh %3d minTh %3d: fast %.3f ms, quadtree %.3f ms; %d candidates, %.0f keypoints per frame"
          % (ini, mn, ms[1], ms[2], ncand, float(d_n.float().mean())), flush=True)
    del ex
