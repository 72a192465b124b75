#!/usr/bin/env python3
"""Device time of SearchByProjection(Cur, Last) for ONE frame pair and for a batch of pairs
(orb_search_by_projection_frames_device, HIP events around the launch), beside the host-to-host
time of the host-buffer entry point: how much of a tracking thread's call is the kernel."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from test_zz_no_alloc import _inputs
from orb_slam2_amd import _lib
from orb_slam2_amd._lib import FrameBounds, check

L = _lib.lib()
dev = torch.device("cuda", 0)
# usage: match_kernel_time.py [pairs ...]      default: 1 and 255 pairs of 2000 keypoints
for npairs, n in [(int(a), 2000) for a in sys.argv[1:]] or [(1, 2000), (255, 2000)]:
    host, sc = _inputs(np.random.default_rng(1), npairs, n, n, [n] * npairs, [n] * npairs)
    d = {k: torch.from_numpy(v).to(dev) for k, v in host.items()}
    K, sf, bounds = sc["K"], sc["sf"], FrameBounds(*sc["bounds"])
    nbytes = L.orb_match_scratch_bytes_for(npairs, n, n)
    scratch = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    m = torch.zeros((npairs, n), dtype=torch.int32, device=dev)
    nm = torch.zeros(npairs, dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    args = [0, npairs, n, n, d["tcw"].data_ptr(), float(K[0]), float(K[1]), float(K[2]), float(K[3]), bounds, sf.ctypes.data, 8,
            d["nl"].data_ptr(), d["lk"].data_ptr(), d["lobs"].data_ptr(), d["lxw"].data_ptr(), d["ldesc"].data_ptr(),
            d["nc"].data_ptr(), d["ck"].data_ptr(), d["cdesc"].data_ptr(), d["cobs"].data_ptr(), 7.0, 1,
            scratch.data_ptr(), nbytes, m.data_ptr(), nm.data_ptr(), st]
    for _ in range(3):
        check(L.orb_search_by_projection_frames_device(*args))
    torch.cuda.synchronize()
    ts = []
    for _ in range(30):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); check(L.orb_search_by_projection_frames_device(*args)); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print("%3d pair(s) x %4d keypoints: kernel %.3f ms (median of 30), %d matches in pair 0" % (npairs, n, float(np.median(ts)), int(nm[0])), flush=True)
