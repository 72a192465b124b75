#!/usr/bin/env python3
"""Latency of the host-buffer SearchByProjection(Cur, Last) entry point: one call = one frame
pair with 2000 features (C call alone, and with the Python mirror around it)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import kp_struct, projection_scene, scale_tables
from orb_slam2_amd import Frame, MapPoint, ORBmatcher

rng = np.random.default_rng(0)
n = 2000
sc = projection_scene(rng, n, n, 376, 1241)
sf, s2 = scale_tables()
L, Cu = sc["last"], sc["cur"]
last = Frame(kp_struct(np.zeros(n), np.zeros(n), L["octave"], L["angle"]), np.zeros((n, 32), np.uint8), sf, s2, sc["K"], sc["bounds"])
for i in range(n):
    if L["mp_obs"][i] >= 0:
        last.mvpMapPoints[i] = MapPoint(L["xw"][i], L["mp_desc"][i], L["mp_obs"][i])
m = ORBmatcher(0.9, True)
tc = [0.0]
raw = m._lib.orb_search_by_projection_frames
class Timed:
    def __call__(self, *a):
        t0 = time.perf_counter(); r = raw(*a); tc[0] += time.perf_counter() - t0; return r
m._lib = type("L", (), {"orb_search_by_projection_frames": Timed()})()
K = 20
for rep in range(2):
    tc[0] = 0.0
    t0 = time.perf_counter()
    for k in range(K):
        cur = Frame(kp_struct(Cu["x"], Cu["y"], Cu["octave"], Cu["angle"]), Cu["desc"], sf, s2, sc["K"], sc["bounds"], sc["Tcw"])
        nm = m.SearchByProjection(cur, last, 7.0, True)
    dt = (time.perf_counter() - t0) / K
print("SearchByProjection(Cur, Last): %d matches; C call %.3f ms, with the Python mirror %.3f ms" % (nm, tc[0] / K * 1e3, dt * 1e3))
