"""Host-buffer batch throughput (images and results in pageable host memory) for one build of the library.
usage: python tools/host_batch_time.py [lib.so] [--batch N] [--reps R] [--threads 8,12,16]"""
import argparse
import os
import sys
import time

import numpy as np

ap = argparse.ArgumentParser()
ap.add_argument("lib", nargs="?", default=None)
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--reps", type=int, default=9)
ap.add_argument("--threads", default="", help="comma-separated ORBHIP_HOST_THREADS values to go through")
args = ap.parse_args()
sys.path.insert(0, ".")
from orb_slam2_amd import _lib  # noqa: E402

if args.lib:
    _lib.LIB_PATH = os.path.abspath(args.lib)
from orb_slam2_amd import ORBextractor  # noqa: E402
from orb_slam2_amd.synth import synth_sequence  # noqa: E402

from orb_slam2_amd._lib import KP_DTYPE, check, lib  # noqa: E402

ROWS, COLS, MAX_KP = 376, 1241, 2600
seq = synth_sequence(ROWS, COLS, args.batch, seed=3)
L = lib()
hk = np.zeros((args.batch, MAX_KP), KP_DTYPE)
hd = np.zeros((args.batch, MAX_KP, 32), np.uint8)
hn = np.zeros(args.batch, np.int32)

for th in (args.threads.split(",") if args.threads else [""]):
    if th:
        os.environ["ORBHIP_HOST_THREADS"] = th
    ex = ORBextractor(2000, 1.2, 8, 20, 7)  # the handle makes its host threads at the first large batch

    def once():
        # the C entry point with caller-owned arrays, as bench.py times it
        check(L.orb_extract_batch(ex._h, args.batch, seq.ctypes.data, ROWS, COLS, COLS, ROWS * COLS, MAX_KP,
                                  hk.ctypes.data, hd.ctypes.data, hn.ctypes.data))

    once()
    once()
    t = []
    for _ in range(args.reps):
        t0 = time.perf_counter()
        once()
        t.append(time.perf_counter() - t0)
    print("%-20s threads %-3s %d frames: median %.3f ms (%.1f k frames/s), min %.3f ms" % (
        os.path.basename(args.lib or "liborbhip.so"), th or "-", args.batch, 1e3 * float(np.median(t)),
        args.batch / float(np.median(t)) / 1e3, 1e3 * min(t)))
    del ex
