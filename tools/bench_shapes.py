#!/usr/bin/env python3
"""Extraction rate at the other configured shapes (device-resident batch, one GPU):
    python tools/bench_shapes.py [channels]      (1 grey, 3 RGB, 4 RGBA: converted while level 0 is written)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from orb_slam2_amd import _lib
from orb_slam2_amd.extractor import ORBextractor
from orb_slam2_amd.synth import synth_sequence

CH = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda", 0)
for name, rows, cols, nf, B in [("TUM 640x480/1000", 480, 640, 1000, 128), ("KITTI 1241x376/2000", 376, 1241, 2000, 128),
                                 ("EuRoC 752x480/1200", 480, 752, 1200, 128), ("4K 3840x2160/4000", 2160, 3840, 4000, 16),
                                 # the configuration the reference ships (Examples/Monocular/shenzhen_simple.yaml:28-36) and
                                 # its initialisation extractor with twice the features (src/Tracking.cc:174-182)
                                 ("shipped 1920x1440/4000", 1440, 1920, 4000, 32), ("shipped 1920x1440/8000", 1440, 1920, 8000, 32)]:
    seq = synth_sequence(rows, cols, min(B, 8), seed=5, step=2)
    seq = np.concatenate([seq] * (B // len(seq)), 0)
    if CH > 1:  # grey replicated into the colour channels: same corners, 3-4x the input bytes
        seq = np.ascontiguousarray(np.repeat(seq[..., None], CH, 3))
    d_img = torch.from_numpy(seq).to(dev)
    ex = ORBextractor(nf, 1.2, 8, 20, 7, device=0)
    ex.set_input_format(CH, True)
    max_kp = ex.max_keypoints(rows, cols)
    d_kps = torch.zeros((B, max_kp, 6), dtype=torch.float32, device=dev)
    d_desc = torch.zeros((B, max_kp, 32), dtype=torch.uint8, device=dev)
    d_n = torch.zeros(B, dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    run = lambda: ex.extract_batch_device(d_img.data_ptr(), B, rows, cols, d_kps.data_ptr(), d_desc.data_ptr(), d_n.data_ptr(), max_kp, stream=st)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    L = _lib.lib()
    L.orb_extractor_set_profiling(ex._h, 1)
    t0 = time.perf_counter()
    K = 10
    for _ in range(K):
        run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    ms = np.zeros(5, np.float32)
    _lib.check(L.orb_extractor_stage_times(ex._h, ms.ctypes.data))
    print("%-22s batch %3d: %9.1f frames/s  (%.3f ms/batch; pyramid %.3f fast %.3f quadtree %.3f blur %.3f orient_desc %.3f; %.0f kp/frame)"
          % (name, B, B / dt, dt * 1e3, *ms, float(d_n.float().mean())), flush=True)
    if name.startswith("shipped"):
        # one frame host to host, the way Frame::ExtractORB calls the reference (src/Frame.cc:220-228)
        L.orb_extractor_set_profiling(ex._h, 0)
        ex1 = ORBextractor(nf, 1.2, 8, 20, 7, device=0)
        img = np.ascontiguousarray(seq[0])
        hk = np.zeros(max_kp, _lib.KP_DTYPE); hd = np.zeros((max_kp, 32), np.uint8); hn = np.zeros(1, np.int32)
        call = lambda: _lib.check(L.orb_extract(ex1._h, img.ctypes.data, rows, cols, cols * CH, max_kp, hk.ctypes.data, hd.ctypes.data, hn.ctypes.data))
        ex1.set_input_format(CH, True)
        for _ in range(5):
            call()
        ts = []
        for _ in range(100):
            t = time.perf_counter(); call(); ts.append(time.perf_counter() - t)
        print("%-22s one frame host to host: %.3f ms median, %.3f ms p90 (%d keypoints)"
              % (name, 1e3 * float(np.median(ts)), 1e3 * float(np.percentile(ts, 90)), int(hn[0])), flush=True)
        del ex1
    del ex, d_img, d_kps, d_desc
    torch.cuda.empty_cache()
