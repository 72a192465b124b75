"""One KITTI-sized frame host to host, repeated: for a kernel trace of the small-batch (graph) path.
usage: python tools/latency_trace.py [repeats]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from orb_slam2_amd import ORBextractor  # noqa: E402
from orb_slam2_amd.synth import synth_image  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
img = synth_image(376, 1241, 42)
ex = ORBextractor(2000, 1.2, 8, 20, 7)
for _ in range(5):
    ex(img)
t = []
for _ in range(n):
    t0 = time.perf_counter()
    ex(img)
    t.append(time.perf_counter() - t0)
print("median %.4f ms, min %.4f ms" % (1e3 * float(np.median(t)), 1e3 * min(t)))
