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


def test_extractor_changes_size_and_reads_levels_while_matchers_run(oracle):
    """The tracking thread's handle changes image size every other call (each change reallocates
    its buffers) and reads a pyramid level and a blurred level back after every call, while two
    other threads keep calling SearchByProjection and the medoid of ComputeDistinctiveDescriptors:
    the allocations, releases and read-backs of one thread wait for that thread's own work only,
    and everybody's results stay equal to the oracle's."""
    from orb_slam2_amd import ORBextractor, _lib
    from orb_slam2_amd._lib import FrameBounds, check, ptr
    from orb_slam2_amd.synth import synth_image
    L = _lib.lib()
    shapes = [(240, 320), (200, 500), (376, 1241)]
    imgs = [synth_image(r, c, seed=90 + i) for i, (r, c) in enumerate(shapes)]
    refs = []
    for im in imgs:
        ref = oracle.Extractor(700, 1.2, 6, 20, 7)
        rk, rd = ref(im)
        refs.append((rk, rd, ref.pyramid_level(2), ref.blurred_level(1)))
    ex = ORBextractor(700, 1.2, 6, 20, 7)

    def run_extract(it):
        i = (it // 2) % len(imgs)      # two calls per size, then the next size
        k, d = ex(imgs[i])
        rk, rd, lvl2, blur1 = refs[i]
        assert len(k) == len(rk)
        for f in ("x", "y", "octave", "response"):
            assert np.array_equal(k[f], rk[f]), f
        assert (d == rd).all(1).mean() >= 0.999
        assert np.array_equal(ex.pyramid_level(2, padded=True), lvl2)
        if blur1 is not None:
            assert np.array_equal(ex.blurred_level(1), blur1)

    rng = np.random.default_rng(12)
    sc = projection_scene(rng, 1500, 1600, 376, 1241, conflict_clusters=20)
    want_n, want_m = oracle.search_by_projection_frames(sc["Tcw"], sc["K"], sc["bounds"], sc["sf"], sc["last"], sc["cur"], 7.0, True)
    La, Cu = sc["last"], sc["cur"]
    nl, nc = len(La["octave"]), len(Cu["x"])
    tcw = np.ascontiguousarray(sc["Tcw"].reshape(16), np.float32)
    n_l, n_c = np.array([nl], np.int32), np.array([nc], np.int32)

    def run_projection(it):
        m, nm = np.zeros(nc, np.int32), np.zeros(1, np.int32)
        K = sc["K"]
        check(L.orb_search_by_projection_frames(
            0, 1, nl, nc, ptr(tcw), float(K[0]), float(K[1]), float(K[2]), float(K[3]), FrameBounds(*sc["bounds"]),
            ptr(sc["sf"]), 8, ptr(n_l), ptr(La["mp_obs"]), ptr(La["xw"]), ptr(La["mp_desc"]), ptr(La["octave"]), ptr(La["angle"]),
            ptr(n_c), ptr(Cu["x"]), ptr(Cu["y"]), ptr(Cu["octave"]), ptr(Cu["angle"]), ptr(Cu["desc"]), ptr(Cu["obs"]),
            7.0, 1, ptr(m), ptr(nm)))
        assert int(nm[0]) == want_n and np.array_equal(m, want_m)

    # map points with 2 .. 90 observing descriptors each (both medoid kernels), sizes that change per round
    counts = rng.integers(2, 90, 300)
    start = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    descs = rng.integers(0, 256, (int(start[-1]), 32), dtype=np.uint8)
    want_med = np.array([oracle.distinctive_descriptor(descs[start[p]:start[p + 1]])[0] for p in range(len(counts))], np.int32)

    def run_medoid(it):
        npts = 100 + 100 * (it % 3)
        bi, bm = np.zeros(npts, np.int32), np.zeros(npts, np.int32)
        check(L.orb_distinctive_descriptors(0, npts, ptr(start), ptr(descs), ptr(bi), ptr(bm)))
        assert np.array_equal(bi, want_med[:npts])

    for f in (run_extract, run_projection, run_medoid):
        f(0)
    errors = []
    go = threading.Barrier(3)

    def worker(f):
        try:
            go.wait(timeout=60)
            for it in range(ROUNDS):
                f(it)
        except BaseException as e:  # noqa: BLE001 - reported below
            errors.append((f.__name__, repr(e)))

    threads = [threading.Thread(target=worker, args=(f,)) for f in (run_extract, run_projection, run_medoid)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(150)
        assert not t.is_alive(), "a worker is stuck"
    assert not errors, errors
