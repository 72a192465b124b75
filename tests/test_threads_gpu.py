"""The reference calls its matchers from three threads (Tracking src/Tracking.cc:1065,
LocalMapping src/LocalMapping.cc:327, LoopClosing src/LoopClosing.cc:307) while the tracking
thread also extracts.  Here three host threads call SearchByProjection(Cur, Last),
SearchForTriangulation and ORBextractor::operator() at the same time, many times over, and every
result must still equal the oracle's (bit-exact integers; angles within 1e-4 rad)."""
import threading

import numpy as np
import pytest

from helpers import flat_feature_vector, kp_struct, projection_scene, triangulation_scene

pytestmark = pytest.mark.gpu

ROUNDS = 12


def test_three_threads_match_and_extract_concurrently(oracle):
    from orb_slam2_amd import Frame, KeyFrame, MapPoint, ORBextractor, ORBmatcher
    from orb_slam2_amd.matcher import _mat3_mul_add
    from orb_slam2_amd.synth import synth_image

    # -- projection search: inputs and expectation
    rng = np.random.default_rng(3)
    sc = projection_scene(rng, 1800, 1900, 376, 1241, conflict_clusters=30)
    want_p = oracle.search_by_projection_frames(sc["Tcw"], sc["K"], sc["bounds"], sc["sf"], sc["last"], sc["cur"], 7.0, True)
    La, Cu = sc["last"], sc["cur"]
    s2 = (sc["sf"] * sc["sf"]).astype(np.float32)

    def run_projection():
        nl, nc = len(La["octave"]), len(Cu["x"])
        last = Frame(kp_struct(np.zeros(nl), np.zeros(nl), La["octave"], La["angle"]), np.zeros((nl, 32), np.uint8),
                     sc["sf"], s2, sc["K"], sc["bounds"])
        for i in range(nl):
            if La["mp_obs"][i] >= 0:
                last.mvpMapPoints[i] = MapPoint(La["xw"][i], La["mp_desc"][i], La["mp_obs"][i])
        cur = Frame(kp_struct(Cu["x"], Cu["y"], Cu["octave"], Cu["angle"]), Cu["desc"], sc["sf"], s2, sc["K"],
                    sc["bounds"], sc["Tcw"])
        for i2 in range(nc):
            if Cu["obs"][i2] >= 0:
                cur.mvpMapPoints[i2] = MapPoint(np.zeros(3), np.zeros(32, np.uint8), Cu["obs"][i2])
        n = ORBmatcher(0.9, True).SearchByProjection(cur, last, 7.0, True)
        assert n == want_p[0]
        for i2 in np.nonzero(want_p[1] >= 0)[0]:
            assert cur.mvpMapPoints[i2] is last.mvpMapPoints[want_p[1][i2]]
        for i2 in np.nonzero(want_p[1] == -2)[0]:
            assert cur.mvpMapPoints[i2] is None

    # -- triangulation search
    ts = triangulation_scene(np.random.default_rng(4), 1700, 1800)
    k1, k2 = ts["k1"], ts["k2"]
    kf1 = KeyFrame(kp_struct(k1["x"], k1["y"], k1["octave"], k1["angle"]), k1["desc"], ts["sf"], ts["s2"], ts["K"],
                   ts["bounds"], ts["T1"], feat_vec=ts["fv1"])
    kf2 = KeyFrame(kp_struct(k2["x"], k2["y"], k2["octave"], k2["angle"]), k2["desc"], ts["sf"], ts["s2"], ts["K"],
                   ts["bounds"], ts["T2"], feat_vec=ts["fv2"])
    for i in np.nonzero(k1["has_mp"])[0]:
        kf1.mvpMapPoints[i] = MapPoint(np.zeros(3), k1["desc"][i])
    for i in np.nonzero(k2["has_mp"])[0]:
        kf2.mvpMapPoints[i] = MapPoint(np.zeros(3), k2["desc"][i])
    K = ts["K"]
    C2 = _mat3_mul_add(kf2.GetRotation(), kf1.GetCameraCenter(), kf2.GetTranslation())
    invz = np.float32(1.0) / C2[2]
    ex = np.float32(np.float32(np.float32(K[0] * C2[0]) * invz) + K[2])
    ey = np.float32(np.float32(np.float32(K[1] * C2[1]) * invz) + K[3])
    want_n, want_m = oracle.search_for_triangulation(k1, k2, flat_feature_vector(kf1.mFeatVec),
                                                     flat_feature_vector(kf2.mFeatVec), ts["F12"], (ex, ey), ts["sf"],
                                                     ts["s2"], False)
    want_pairs = [(i, int(want_m[i])) for i in range(len(want_m)) if want_m[i] >= 0]
    assert want_n > 20

    def run_triangulation():
        pairs = []
        n = ORBmatcher(0.6, False).SearchForTriangulation(kf1, kf2, ts["F12"], pairs, False)
        assert n == want_n and pairs == want_pairs

    # -- extraction on a handle of its own
    img = synth_image(376, 1241, seed=77)
    ex_ = ORBextractor(2000, 1.2, 8, 20, 7, device=0)
    rk, rd = oracle.Extractor(2000, 1.2, 8, 20, 7)(img)

    def run_extract():
        k, d = ex_(img)
        assert len(k) == len(rk)
        for f in ("x", "y", "octave", "response", "size"):
            assert np.array_equal(k[f], rk[f]), f
        assert np.abs(k["angle"] - rk["angle"]).max() < 1e-4 * 180 / np.pi
        assert (d == rd).all(1).mean() >= 0.999

    for f in (run_projection, run_triangulation, run_extract):  # each alone first
        f()
    errors = []
    start = threading.Barrier(3)

    def worker(f):
        try:
            start.wait(timeout=60)
            for _ in range(ROUNDS):
                f()
        except BaseException as e:  # noqa: BLE001 - reported below
            errors.append((f.__name__, repr(e)))

    threads = [threading.Thread(target=worker, args=(f,)) for f in (run_projection, run_triangulation, run_extract)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(150)
        assert not t.is_alive(), "a worker is stuck"
    assert not errors, errors
