#!/usr/bin/env python3
"""Stage times of the extractor (device-resident batch) for one build of the library.

    python tools/stage_times.py [path/to/liborbhip.so] [--batch 256] [--steps 10] [--rows 376 --cols 1241 --nfeat 2000]

Development aid for kernel experiments: builds of the library with other flags
(make -C orb_slam2_amd/csrc OBJDIR=... OUT=... EXTRA=-D...) are timed side by side."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("lib", nargs="?", default=None)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--rows", type=int, default=376)
    ap.add_argument("--cols", type=int, default=1241)
    ap.add_argument("--nfeat", type=int, default=2000)
    ap.add_argument("--check", type=int, default=1, help="print a checksum of the outputs (equal across exact variants)")
    a = ap.parse_args()
    import torch
    from orb_slam2_amd import _lib
    if a.lib:
        _lib.LIB_PATH = os.path.abspath(a.lib)
    from orb_slam2_amd._lib import check
    from orb_slam2_amd.extractor import ORBextractor
    from orb_slam2_amd.synth import synth_sequence
    L = _lib.lib()
    dev = torch.device("cuda", 0)
    seq = synth_sequence(a.rows, a.cols, a.batch, seed=1000, step=2)
    d_img = torch.from_numpy(seq).to(dev)
    ex = ORBextractor(a.nfeat, 1.2, 8, 20, 7, device=0)
    mk = ex.max_keypoints(a.rows, a.cols)
    kps = torch.zeros((a.batch, mk, 6), dtype=torch.float32, device=dev)
    desc = torch.zeros((a.batch, mk, 32), dtype=torch.uint8, device=dev)
    nout = torch.zeros(a.batch, dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream

    def run():
        ex.extract_batch_device(d_img.data_ptr(), a.batch, a.rows, a.cols, kps.data_ptr(), desc.data_ptr(),
                                nout.data_ptr(), mk, stream=st)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    L.orb_extractor_set_profiling(ex._h, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.steps):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = np.zeros(5, np.float32)
    check(L.orb_extractor_stage_times(ex._h, ms.ctypes.data))
    total = e0.elapsed_time(e1) / a.steps
    names = ["pyramid", "fast", "quadtree", "blur", "orient_desc"]
    line = " ".join("%s=%.4f" % (n, v) for n, v in zip(names, ms))
    extra = ""
    if a.check:
        n = nout.cpu().numpy()
        k = kps.cpu().numpy()
        d = desc.cpu().numpy()
        s = 0
        for f in range(a.batch):
            s = (s * 1000003 + int(k[f, :n[f]].view(np.uint32).astype(np.uint64).sum()) + int(d[f, :n[f]].astype(np.uint64).sum())) % (1 << 61)
        extra = " nkp=%d checksum=%x" % (int(n.sum()), s)
    print("%-40s total=%.4f %s%s" % (os.path.basename(a.lib or "liborbhip.so"), total, line, extra), flush=True)


if __name__ == "__main__":
    main()
