#!/usr/bin/env python3
"""Experiment: one batch of frames extracted as `parts` sub-batches, each through an extractor handle
and a stream of its own, so kernels of different stages run side by side.

    python tools/split_batch_time.py [--batch 256] [--parts 1 2 4] [--steps 20]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--parts", type=int, nargs="+", default=[1, 2, 4])
    ap.add_argument("--steps", type=int, default=20)
    a = ap.parse_args()
    import torch
    from orb_slam2_amd.extractor import ORBextractor
    from orb_slam2_amd.synth import synth_sequence
    rows, cols = 376, 1241
    dev = torch.device("cuda", 0)
    seq = synth_sequence(rows, cols, a.batch, seed=1000, step=2)
    d_img = torch.from_numpy(seq).to(dev)
    for parts in a.parts:
        n = a.batch // parts
        exs = [ORBextractor(2000, 1.2, 8, 20, 7, device=0) for _ in range(parts)]
        mk = exs[0].max_keypoints(rows, cols)
        kps = torch.zeros((a.batch, mk, 6), dtype=torch.float32, device=dev)
        desc = torch.zeros((a.batch, mk, 32), dtype=torch.uint8, device=dev)
        nout = torch.zeros(a.batch, dtype=torch.int32, device=dev)
        streams = [torch.cuda.Stream(device=dev) for _ in range(parts)]

        def run():
            for p in range(parts):
                exs[p].extract_batch_device(d_img.data_ptr() + p * n * rows * cols, n, rows, cols,
                                            kps.data_ptr() + p * n * mk * 24, desc.data_ptr() + p * n * mk * 32,
                                            nout.data_ptr() + p * n * 4, mk, stream=streams[p].cuda_stream)
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            run()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.steps
        print(f"parts={parts}: {dt * 1e3:.4f} ms per {a.batch} frames, nkp={int(nout.sum())}", flush=True)
        del exs


if __name__ == "__main__":
    main()
