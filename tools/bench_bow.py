"""Throughput of the vocabulary-tree transform on one GPU: a complete k=10, L=6 tree (the size
of the ORB vocabulary), 2000 descriptors per frame, frames batched into one launch.
    python tools/bench_bow.py [--frames 64] [--reps 50]"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from orb_slam2_amd import _lib  # noqa: E402
from orb_slam2_amd.bow import ORBVocabulary  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--features", type=int, default=2000)
    ap.add_argument("--reps", type=int, default=50)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--L", type=int, default=6)
    ap.add_argument("--check", type=int, default=2000, help="features compared with the CPU oracle")
    a = ap.parse_args()
    t0 = time.time()
    voc = ORBVocabulary.synthetic_full(a.k, a.L, 1)
    print("vocabulary: %d nodes, %d words (%.1f s to generate)" % (len(voc.parent), voc.size(), time.time() - t0), flush=True)
    rng = np.random.default_rng(2)
    n = a.frames * a.features
    leaves = np.flatnonzero(voc.is_leaf)
    feats = voc.node_desc[leaves[rng.integers(0, len(leaves), n)]] ^ np.packbits(rng.random((n, 256)) < 0.05, axis=1, bitorder="little")
    d = torch.from_numpy(np.ascontiguousarray(feats)).cuda()
    out = torch.empty((2, n), dtype=torch.int32, device="cuda")
    L = _lib.lib()
    h = voc._device()
    s = torch.cuda.current_stream()

    def launch():
        _lib.check(L.orb_bow_transform_device(h, n, C.c_void_p(d.data_ptr()), 4, C.c_void_p(out[0].data_ptr()),
                                              C.c_void_p(out[1].data_ptr()), C.c_void_p(s.cuda_stream)))
    for _ in range(3):
        launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        launch()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.reps
    # bytes a descent touches: k children x 36 B (id + descriptor) per level + the feature itself
    gb = n * (a.L * a.k * 36 + 32 + 8) / 1e9
    print("%d frames x %d features: %.3f ms per launch, %.0f frames/s, %.1f M features/s, %.0f GB/s of gathered node data"
          % (a.frames, a.features, ms, a.frames / ms * 1e3, n / ms / 1e3, gb / ms * 1e3), flush=True)
    if a.check:
        from oracle import oracle
        m = min(a.check, n)
        t0 = time.time()
        ow, on, _ = oracle.bow_transform(voc.parent, voc.is_leaf, voc.node_desc, voc.node_weight, voc.m_L, feats[:m], 4)
        cpu = time.time() - t0
        w = out.cpu().numpy()
        ok = np.array_equal(w[0, :m], ow) and np.array_equal(w[1, :m], on)
        print("oracle: %d features in %.1f ms on one core (%.2f M features/s); parity %s"
              % (m, cpu * 1e3, m / cpu / 1e6, "ok" if ok else "MISMATCH"), flush=True)
        if not ok:
            sys.exit(1)


if __name__ == "__main__":
    main()
