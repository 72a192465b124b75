"""GPU parity of the matcher entry points against the CPU oracle (bit-exact: distances,
indices and accept decisions are integer work)."""
import numpy as np
import pytest

from helpers import flip_bits, kp_struct, projection_scene, random_frame_arrays, scale_tables

pytestmark = pytest.mark.gpu


def test_descriptor_distance_kat(oracle):
    from orb_slam2_amd import ORBmatcher
    z, o = np.zeros(32, np.uint8), np.full(32, 255, np.uint8)
    assert ORBmatcher.DescriptorDistance(z, o) == 256
    assert ORBmatcher.DescriptorDistance(o, o) == 0
    one = z.copy(); one[17] = 4
    assert ORBmatcher.DescriptorDistance(z, one) == 1
    rng = np.random.default_rng(5)
    a = rng.integers(0, 256, (5000, 32), dtype=np.uint8)
    b = rng.integers(0, 256, (5000, 32), dtype=np.uint8)
    got = ORBmatcher.DescriptorDistances(a, b)
    want = np.unpackbits(a ^ b, axis=1).sum(1)
    assert np.array_equal(got, want)
    assert got[7] == oracle.descriptor_distance(a[7], b[7])


@pytest.mark.parametrize("nq,ndb", [(1, 1), (37, 5), (1000, 3000), (2500, 700), (4096, 1024)])
def test_knn2_vs_oracle(oracle, nq, ndb):
    from orb_slam2_amd import knn2_bruteforce
    from orb_slam2_amd.synth import synth_descriptors
    db, q = synth_descriptors(ndb, nq, seed=nq + ndb)
    if ndb > 4:
        db[3] = db[1]  # exact duplicates: the first one must win
    b, s, i, a = knn2_bruteforce(q, db, 0.7)
    rb, rs, ri, ra = oracle.knn2_bruteforce(q, db, 0.7)
    assert np.array_equal(b, rb) and np.array_equal(s, rs) and np.array_equal(i, ri) and np.array_equal(a, ra)


def test_knn2_empty_db():
    from orb_slam2_amd import knn2_bruteforce
    q = np.zeros((5, 32), np.uint8)
    b, s, i, a = knn2_bruteforce(q, np.zeros((0, 32), np.uint8), 0.7)
    assert (b == 256).all() and (s == 256).all() and (i == -1).all() and (a == 0).all()


@pytest.mark.parametrize("seed,nl,nc,clusters,ori", [(0, 300, 400, 0, True), (1, 2000, 2000, 0, True),
                                                      (2, 2000, 1500, 40, True), (3, 1500, 2000, 60, False),
                                                      (4, 5, 3, 0, True)])
def test_search_by_projection_frames(oracle, seed, nl, nc, clusters, ori):
    from orb_slam2_amd import Frame, MapPoint, ORBmatcher
    rng = np.random.default_rng(seed)
    sc = projection_scene(rng, nl, nc, 376, 1241, conflict_clusters=clusters)
    want_n, want_match = oracle.search_by_projection_frames(sc["Tcw"], sc["K"], sc["bounds"], sc["sf"], sc["last"],
                                                            sc["cur"], 7.0, ori)
    sf, s2 = scale_tables()
    L, Cu = sc["last"], sc["cur"]
    last = Frame(kp_struct(np.zeros(nl), np.zeros(nl), L["octave"], L["angle"]), np.zeros((nl, 32), np.uint8), sf, s2,
                 sc["K"], sc["bounds"])
    for i in range(nl):
        if L["mp_obs"][i] >= 0:
            last.mvpMapPoints[i] = MapPoint(L["xw"][i], L["mp_desc"][i], L["mp_obs"][i])
    cur = Frame(kp_struct(Cu["x"], Cu["y"], Cu["octave"], Cu["angle"]), Cu["desc"], sf, s2, sc["K"], sc["bounds"], sc["Tcw"])
    pre = {}
    for i2 in range(nc):
        if Cu["obs"][i2] >= 0:
            cur.mvpMapPoints[i2] = pre[i2] = MapPoint(np.zeros(3), np.zeros(32, np.uint8), Cu["obs"][i2])
    got_n = ORBmatcher(0.9, ori).SearchByProjection(cur, last, 7.0, True)
    assert got_n == want_n
    for i2 in range(nc):
        if want_match[i2] >= 0:
            assert cur.mvpMapPoints[i2] is last.mvpMapPoints[want_match[i2]], i2
        elif want_match[i2] == -2:
            assert cur.mvpMapPoints[i2] is None
        else:
            assert cur.mvpMapPoints[i2] is pre.get(i2)
    assert want_n > 0 or nl < 10


def test_search_by_projection_frames_batch(oracle):
    from orb_slam2_amd import Frame, MapPoint, ORBmatcher
    sf, s2 = scale_tables()
    curs, lasts, wants = [], [], []
    for p in range(5):
        rng = np.random.default_rng(100 + p)
        nl, nc = 800 + 100 * p, 900 - 50 * p
        sc = projection_scene(rng, nl, nc, 376, 1241, conflict_clusters=10)
        wants.append(oracle.search_by_projection_frames(sc["Tcw"], sc["K"], sc["bounds"], sc["sf"], sc["last"], sc["cur"], 7.0, True))
        L, Cu = sc["last"], sc["cur"]
        last = Frame(kp_struct(np.zeros(nl), np.zeros(nl), L["octave"], L["angle"]), np.zeros((nl, 32), np.uint8), sf, s2, sc["K"], sc["bounds"])
        for i in range(nl):
            if L["mp_obs"][i] >= 0:
                last.mvpMapPoints[i] = MapPoint(L["xw"][i], L["mp_desc"][i], L["mp_obs"][i])
        cur = Frame(kp_struct(Cu["x"], Cu["y"], Cu["octave"], Cu["angle"]), Cu["desc"], sf, s2, sc["K"], sc["bounds"], sc["Tcw"])
        for i2 in range(nc):
            if Cu["obs"][i2] >= 0:
                cur.mvpMapPoints[i2] = MapPoint(np.zeros(3), np.zeros(32, np.uint8), Cu["obs"][i2])
        curs.append(cur); lasts.append(last)
    nm, match = ORBmatcher(0.9, True).search_by_projection_frames_batch(curs, lasts, 7.0)
    for p in range(5):
        assert nm[p] == wants[p][0]
        assert np.array_equal(match[p, :curs[p].N], wants[p][1])


@pytest.mark.parametrize("seed,nl,nc,spread", [(0, 1500, 1200, 30.0), (1, 2000, 700, 12.0), (2, 900, 2000, 60.0)])
def test_search_by_projection_frames_in_one_crowded_corner(oracle, seed, nl, nc, spread):
    """Every current keypoint in one small patch, on one level, with descriptors a few bits apart:
    every map point reaches most of them, so all queries hang together in ONE chain of conflicts
    (one wavefront replays it while the others have nothing), kept lists run out and windows are
    scanned again, and equal distances are decided by the visiting order."""
    from orb_slam2_amd import Frame, MapPoint, ORBmatcher
    rng = np.random.default_rng(seed)
    sc = projection_scene(rng, nl, nc, 376, 1241)
    Cu, L = sc["cur"], sc["last"]
    Cu["x"][:] = (600 + rng.uniform(-spread, spread, nc)).astype(np.float32)
    Cu["y"][:] = (180 + rng.uniform(-spread, spread, nc)).astype(np.float32)
    Cu["octave"][:] = 2
    Cu["desc"][:] = flip_bits(rng, np.repeat(Cu["desc"][:1], nc, 0), 10)
    Cu["obs"][:] = -1
    Cu["angle"][:] = (40 + rng.normal(0, 3, nc)).astype(np.float32)
    L["angle"][:] = (55 + rng.normal(0, 3, nl)).astype(np.float32)
    fx, fy, cx, cy = sc["K"]
    depth = rng.uniform(4, 30, nl).astype(np.float32)
    u = (600 + rng.uniform(-spread, spread, nl)).astype(np.float32)
    v = (180 + rng.uniform(-spread, spread, nl)).astype(np.float32)
    Xc = np.stack([(u - cx) / fx * depth, (v - cy) / fy * depth, depth], 1).astype(np.float32)
    L["xw"][:] = (Xc - sc["Tcw"][:3, 3][None]).astype(np.float32)
    L["octave"][:] = rng.integers(1, 4, nl)
    L["mp_desc"][:] = flip_bits(rng, np.repeat(Cu["desc"][:1], nl, 0), 10)
    L["mp_obs"][:] = rng.integers(0, 3, nl)
    want_n, want_match = oracle.search_by_projection_frames(sc["Tcw"], sc["K"], sc["bounds"], sc["sf"], L, Cu, 7.0, True)
    assert want_n > min(nl, nc) // 2, "scene too poor"
    sf, s2 = scale_tables()
    last = Frame(kp_struct(np.zeros(nl), np.zeros(nl), L["octave"], L["angle"]), np.zeros((nl, 32), np.uint8), sf, s2,
                 sc["K"], sc["bounds"])
    for i in range(nl):
        last.mvpMapPoints[i] = MapPoint(L["xw"][i], L["mp_desc"][i], L["mp_obs"][i])
    cur = Frame(kp_struct(Cu["x"], Cu["y"], Cu["octave"], Cu["angle"]), Cu["desc"], sf, s2, sc["K"], sc["bounds"], sc["Tcw"])
    nm, match = ORBmatcher(0.9, True).search_by_projection_frames_batch([cur], [last], 7.0)
    assert nm[0] == want_n and np.array_equal(match[0, :nc], want_match)


@pytest.mark.parametrize("seed,nmp,nf,th,ratio", [(0, 500, 800, 1.0, 0.8), (1, 3000, 2000, 5.0, 0.8), (2, 2000, 600, 3.0, 0.6)])
def test_search_by_projection_mappoints(oracle, seed, nmp, nf, th, ratio):
    from orb_slam2_amd import Frame, MapPoint, ORBmatcher
    rng = np.random.default_rng(seed)
    sf, s2 = scale_tables()
    fx_, fy_, foct, fang, fdesc = random_frame_arrays(rng, nf, 480, 752)
    # crowd part of the frame so that best and second often share a level and compete
    m = rng.integers(0, nf, nf // 3)
    fx_[m] = fx_[m[0]] + rng.uniform(-30, 30, len(m)).astype(np.float32)
    fy_[m] = fy_[m[0]] + rng.uniform(-30, 30, len(m)).astype(np.float32)
    src = rng.integers(0, nf, nmp)
    mps = dict(track=(rng.uniform(0, 1, nmp) < 0.9).astype(np.uint8),
               level=np.clip(foct[src] + rng.integers(0, 2, nmp), 0, 7).astype(np.int32),
               viewcos=rng.uniform(0.99, 1.0, nmp).astype(np.float32),
               projx=(fx_[src] + rng.uniform(-3, 3, nmp)).astype(np.float32),
               projy=(fy_[src] + rng.uniform(-3, 3, nmp)).astype(np.float32),
               desc=flip_bits(rng, fdesc[src], 70), obs=rng.integers(0, 3, nmp).astype(np.int32))
    fobs = np.where(rng.uniform(0, 1, nf) < 0.1, rng.integers(0, 3, nf), -1).astype(np.int32)
    frame = dict(x=fx_, y=fy_, octave=foct, desc=fdesc, obs=fobs)
    bounds = (0.0, 752.0, 0.0, 480.0)
    want_n, want_match = oracle.search_by_projection_mappoints(bounds, sf, frame, mps, th, ratio)
    F = Frame(kp_struct(fx_, fy_, foct, fang), fdesc, sf, s2, (500, 500, 376, 240), bounds)
    pre = {}
    for idx in range(nf):
        if fobs[idx] >= 0:
            F.mvpMapPoints[idx] = pre[idx] = MapPoint(np.zeros(3), np.zeros(32, np.uint8), fobs[idx])
    vp = []
    for i in range(nmp):
        mp = MapPoint(np.zeros(3), mps["desc"][i], mps["obs"][i])
        mp.mbNeedTrackInView = bool(mps["track"][i])
        mp.mnTrackedScaleLevel, mp.mTrackedViewCosine = int(mps["level"][i]), float(mps["viewcos"][i])
        mp.mTrackedProjX, mp.mTrackedProjY = float(mps["projx"][i]), float(mps["projy"][i])
        vp.append(mp)
    got_n = ORBmatcher(ratio, True).SearchByProjection(F, vp, th)
    assert got_n == want_n
    assert want_n > 0, "scene too poor"
    for idx in range(nf):
        if want_match[idx] >= 0:
            assert F.mvpMapPoints[idx] is vp[want_match[idx]]
        else:
            assert F.mvpMapPoints[idx] is pre.get(idx)


@pytest.mark.parametrize("seed,n1,n2,ori", [(0, 600, 700, False), (1, 2000, 2000, False), (2, 1500, 1800, True)])
def test_search_for_triangulation(oracle, seed, n1, n2, ori):
    from orb_slam2_amd import KeyFrame, MapPoint, ORBmatcher
    rng = np.random.default_rng(seed)
    sf, s2 = scale_tables()
    rows, cols = 480, 752
    x2, y2, o2, a2, d2 = random_frame_arrays(rng, n2, rows, cols)
    src = rng.integers(0, n2, n1)
    # pure x-translation between the keyframes: epipolar lines are image rows
    x1 = (x2[src] + rng.uniform(5, 60, n1)).astype(np.float32)
    y1 = (y2[src] + rng.normal(0, 0.7, n1)).astype(np.float32)
    o1 = o2[src].copy()
    a1 = ((a2[src] + rng.normal(0, 6, n1)) % 360).astype(np.float32)
    d1 = flip_bits(rng, d2[src], 45)
    K = (np.float32(500), np.float32(500), np.float32(376), np.float32(240))
    F12 = np.array([[0, 0, 0], [0, 0, -1], [0, 1, 0]], np.float32) * np.float32(0.002)
    nnodes = 60
    node_of2 = rng.integers(0, nnodes, n2)
    node_of1 = np.where(rng.uniform(0, 1, n1) < 0.8, node_of2[src], rng.integers(0, nnodes + 5, n1))
    fv1 = {int(k) * 3: [int(i) for i in np.nonzero(node_of1 == k)[0]] for k in np.unique(node_of1)}
    fv2 = {int(k) * 3: [int(i) for i in np.nonzero(node_of2 == k)[0]] for k in np.unique(node_of2)}
    has1 = rng.uniform(0, 1, n1) < 0.3
    has2 = rng.uniform(0, 1, n2) < 0.3
    T1 = np.eye(4, dtype=np.float32)
    T2 = np.eye(4, dtype=np.float32); T2[0, 3] = 0.5; T2[2, 3] = 0.3
    kf1 = KeyFrame(kp_struct(x1, y1, o1, a1), d1, sf, s2, K, (0, cols, 0, rows), T1, feat_vec=fv1)
    kf2 = KeyFrame(kp_struct(x2, y2, o2, a2), d2, sf, s2, K, (0, cols, 0, rows), T2, feat_vec=fv2)
    for i in np.nonzero(has1)[0]:
        kf1.mvpMapPoints[i] = MapPoint(np.zeros(3), d1[i])
    for i in np.nonzero(has2)[0]:
        kf2.mvpMapPoints[i] = MapPoint(np.zeros(3), d2[i])
    pairs = []
    got_n = ORBmatcher(0.6, ori).SearchForTriangulation(kf1, kf2, F12, pairs, False)

    def flat(fv):
        nodes = np.array(list(fv.keys()), np.int32); start = [0]; items = []
        for v in fv.values():
            items.extend(v); start.append(len(items))
        return dict(node=nodes, start=np.array(start, np.int32), items=np.array(items, np.int32))
    from orb_slam2_amd.matcher import _mat3_mul_add
    C2 = _mat3_mul_add(kf2.GetRotation(), kf1.GetCameraCenter(), kf2.GetTranslation())
    invz = np.float32(1.0) / C2[2]
    ex = np.float32(np.float32(np.float32(K[0] * C2[0]) * invz) + K[2])
    ey = np.float32(np.float32(np.float32(K[1] * C2[1]) * invz) + K[3])
    want_n, want_m = oracle.search_for_triangulation(
        dict(x=x1, y=y1, octave=o1, angle=a1, desc=d1, has_mp=has1.astype(np.uint8)),
        dict(x=x2, y=y2, octave=o2, angle=a2, desc=d2, has_mp=has2.astype(np.uint8)),
        flat(kf1.mFeatVec), flat(kf2.mFeatVec), F12, (ex, ey), sf, s2, ori)
    assert got_n == want_n
    assert want_n > 20, "scene too poor"
    assert pairs == [(i, int(want_m[i])) for i in range(n1) if want_m[i] >= 0]


def test_projection_many_queries_one_spot(oracle):
    """Forty map points with the same descriptor project onto one crowded spot and each
    occupies what it takes: every query has to skip what the earlier ones took, far beyond
    the few candidates a query keeps from its first scan."""
    from orb_slam2_amd import Frame, MapPoint, ORBmatcher
    rng = np.random.default_rng(77)
    sf, s2 = scale_tables()
    nc, nl = 300, 40
    cx_, cy_, coct, cang, cdesc = random_frame_arrays(rng, nc, 376, 1241)
    base = rng.integers(0, 256, 32, dtype=np.uint8)
    crowd = np.arange(60)
    cx_[crowd] = 600 + rng.uniform(-5, 5, 60).astype(np.float32)
    cy_[crowd] = 200 + rng.uniform(-5, 5, 60).astype(np.float32)
    coct[crowd] = 2
    cdesc[crowd] = flip_bits(rng, np.repeat(base[None], 60, 0), 30)
    K = (np.float32(700), np.float32(700), np.float32(620.5), np.float32(188))
    Xw = np.tile(np.array([[(600 - 620.5) / 700 * 10, (200 - 188) / 700 * 10, 10]], np.float32), (nl, 1))
    last = dict(mp_obs=np.full(nl, 2, np.int32), xw=Xw, mp_desc=np.repeat(base[None], nl, 0),
                octave=np.full(nl, 2, np.int32), angle=rng.uniform(0, 360, nl).astype(np.float32))
    cur = dict(x=cx_, y=cy_, octave=coct, angle=cang, desc=cdesc, obs=np.full(nc, -1, np.int32))
    bounds = (0.0, 1241.0, 0.0, 376.0)
    want_n, want_m = oracle.search_by_projection_frames(np.eye(4, dtype=np.float32), K, bounds, sf, last, cur, 7.0, False)
    assert want_n == nl and len(set(want_m[want_m >= 0].tolist())) == nl  # forty distinct slots
    lf = Frame(kp_struct(np.zeros(nl), np.zeros(nl), last["octave"], last["angle"]), np.zeros((nl, 32), np.uint8), sf, s2, K, bounds)
    for i in range(nl):
        lf.mvpMapPoints[i] = MapPoint(Xw[i], base, 2)
    cf = Frame(kp_struct(cx_, cy_, coct, cang), cdesc, sf, s2, K, bounds)
    nm, match = ORBmatcher(0.9, False).search_by_projection_frames_batch([cf], [lf], 7.0)
    assert nm[0] == want_n and np.array_equal(match[0, :nc], want_m)


def test_knn2_merge_of_shards(oracle):
    """Three database shards searched separately on one GPU and merged equal one scan."""
    import torch
    from orb_slam2_amd.dist import hip_local_search, hip_merge, shard_bounds
    from orb_slam2_amd.synth import synth_descriptors
    db, q = synth_descriptors(5000, 1500, seed=9)
    db[3000] = db[10]; db[4000] = db[10]
    dev = torch.device("cuda", 0)
    d_q = torch.from_numpy(q).to(dev)
    parts = []
    for r in range(3):
        lo, hi = shard_bounds(len(db), 3, r)
        parts.append(hip_local_search(d_q, torch.from_numpy(db[lo:hi]).to(dev), lo, 0.7, 0)().clone())
    g = torch.stack(parts, 1).contiguous()  # [3][shards][nq]
    out, acc = hip_merge(0.7, 0)(g)
    torch.cuda.synchronize()
    rb, rs, ri, ra = oracle.knn2_bruteforce(q, db, 0.7)
    assert np.array_equal(out[0].cpu().numpy(), rb) and np.array_equal(out[1].cpu().numpy(), rs)
    assert np.array_equal(out[2].cpu().numpy(), ri) and np.array_equal(acc.cpu().numpy(), ra)


def _windowed_case(rng, nt, nq, rows=480, cols=752):
    tx, ty, to, ta, td = random_frame_arrays(rng, nt, rows, cols)
    crowd = rng.integers(0, nt, nt // 4)
    tx[crowd] = tx[crowd[0]] + rng.uniform(-12, 12, len(crowd)).astype(np.float32)
    ty[crowd] = ty[crowd[0]] + rng.uniform(-12, 12, len(crowd)).astype(np.float32)
    src = rng.integers(0, nt, nq)
    lvl = np.clip(to[src] + rng.integers(-1, 2, nq), 0, 7).astype(np.int32)
    sf, _ = scale_tables()
    train = dict(x=tx, y=ty, octave=to, angle=ta, desc=td, taken=(rng.uniform(0, 1, nt) < 0.15).astype(np.uint8))
    q = dict(active=(rng.uniform(0, 1, nq) < 0.9).astype(np.uint8),
             u=(tx[src] + rng.uniform(-5, 5, nq)).astype(np.float32), v=(ty[src] + rng.uniform(-5, 5, nq)).astype(np.float32),
             r=(np.float32(10) * sf[lvl]).astype(np.float32), minl=lvl - 1, maxl=lvl + rng.integers(0, 2, nq).astype(np.int32),
             angle=((ta[src] + rng.normal(0, 10, nq)) % 360).astype(np.float32), desc=flip_bits(rng, td[src], 70))
    return train, q, (0.0, float(cols), 0.0, float(rows))


@pytest.mark.parametrize("seed,nt,nq,maxd,ori", [(0, 900, 700, 50, False), (1, 2000, 3000, 100, True), (2, 1500, 2500, 50, True)])
def test_match_windowed_vs_oracle(oracle, seed, nt, nq, maxd, ori):
    import ctypes as C
    from orb_slam2_amd._lib import FrameBounds, check, lib, ptr
    rng = np.random.default_rng(seed)
    train, q, bounds = _windowed_case(rng, nt, nq)
    want_n, want_m = oracle.match_windowed(bounds, train, q, maxd, ori)
    match = np.zeros(nt, np.int32); nm = C.c_int()
    f = lambda a, t: np.ascontiguousarray(a, t)
    args = [f(train["x"], "f4"), f(train["y"], "f4"), f(train["octave"], "i4"), f(train["angle"], "f4"), f(train["desc"], "u1"),
            f(train["taken"], "u1")]
    qa = [f(q["active"], "u1"), f(q["u"], "f4"), f(q["v"], "f4"), f(q["r"], "f4"), f(q["minl"], "i4"), f(q["maxl"], "i4"),
          f(q["angle"], "f4"), f(q["desc"], "u1")]
    check(lib().orb_match_windowed(0, FrameBounds(*bounds), nt, *[ptr(a) for a in args], nq, *[ptr(a) for a in qa],
                                   maxd, int(ori), ptr(match), C.byref(nm)))
    assert nm.value == want_n and np.array_equal(match, want_m)
    assert want_n > 50, "scene too poor"


def test_relocalisation_and_sim3_searches(oracle):
    """The two remaining SearchByProjection overloads through the host mirror: the window
    parameters come from the mirror's pose algebra, the scan from the GPU; compared with the
    oracle's scan on the same windows and with the gates of the reference restated here."""
    from orb_slam2_amd import Frame, KeyFrame, MapPoint, ORBmatcher
    rng = np.random.default_rng(11)
    sf, s2 = scale_tables()
    rows, cols, n = 480, 752, 1200
    K = (np.float32(500), np.float32(500), np.float32(376), np.float32(240))
    bounds = (0.0, float(cols), 0.0, float(rows))
    x, y, o, a, d = random_frame_arrays(rng, n, rows, cols)
    T = np.eye(4, dtype=np.float32); T[0, 3] = 0.1
    cur = Frame(kp_struct(x, y, o, a), d, sf, s2, K, bounds, T)
    kf = KeyFrame(kp_struct(x + 1, y, o, a), d, sf, s2, K, bounds, np.eye(4, dtype=np.float32))
    mps = []
    for i in range(n):
        if rng.uniform() < 0.7:
            Z = np.float32(rng.uniform(3, 20))
            Xc = np.array([(x[i] - K[2]) / K[0] * Z, (y[i] - K[3]) / K[1] * Z, Z], np.float32)
            mp = MapPoint(Xc - T[:3, 3], flip_bits(rng, d[i:i + 1], 40)[0], 1)
            mp.mfMaxDistance = float(Z * sf[o[i]] * rng.uniform(0.9, 1.1)); mp.mfMinDistance = mp.mfMaxDistance / float(sf[7])
            kf.mvpMapPoints[i] = mp
            mps.append(mp)
    already = set(mps[::17])
    for i2 in range(0, n, 9):
        cur.mvpMapPoints[i2] = MapPoint(np.zeros(3), d[i2])
    before = list(cur.mvpMapPoints)
    got = ORBmatcher(0.9, True).SearchByProjection(cur, kf, already, 10.0, 100)
    new = [i2 for i2 in range(n) if cur.mvpMapPoints[i2] is not before[i2]]
    assert got > 200 and len(new) >= got
    for i2 in new:
        mp = cur.mvpMapPoints[i2]
        if mp is not None:
            assert mp not in already and before[i2] is None
            assert oracle.descriptor_distance(mp.GetDescriptor(), d[i2]) <= 100
    # Sim3 variant: identity similarity with scale 2 (points twice as far in the keyframe's view)
    S = np.eye(4, dtype=np.float32); S[:3, :3] *= 2
    matched = [None] * n
    for i2 in range(0, n, 7):
        matched[i2] = mps[i2 % len(mps)]
    pts = [mp for mp in mps if rng.uniform() < 0.8]
    pre = list(matched)
    got2 = ORBmatcher(0.75, True).SearchByProjection(kf, S, pts, matched, 10)
    new2 = [i for i in range(n) if matched[i] is not pre[i]]
    assert got2 == len(new2) and got2 > 100
    for i in new2:
        assert pre[i] is None and oracle.descriptor_distance(matched[i].GetDescriptor(), d[i]) <= 50
        assert matched[i] not in [m for m in pre if m is not None]


def test_stereo_pair_euroc_shape(oracle):
    """752x480 stereo configuration: both eyes through the extractor (1200 features), then the
    row-band Hamming association; best index, best distance and match count against the oracle."""
    from orb_slam2_amd import ORBextractor, stereo_match
    from orb_slam2_amd.synth import synth_sequence
    rows, cols, disp = 480, 752, 12
    left, right = synth_sequence(rows, cols, 2, seed=21, step=disp)  # the scene shifts left in the right eye
    ex = ORBextractor(1200, 1.2, 8, 20, 7)
    lk, ld = ex(left)
    rk, rd = ex(right)
    sf = ex.GetScaleFactors()
    ref_ex = oracle.Extractor(1200, 1.2, 8, 20, 7)
    ok, od = ref_ex(right)
    assert len(ok) == len(rk) and np.array_equal(od, rd)  # the right eye too is bit-exact
    for (mind, maxd, th) in [(0.0, 60.0, 75), (0.0, 752.0, 100), (5.0, 20.0, 50)]:
        n, bi, bd = stereo_match(rows, (lk, ld), (rk, rd), sf, mind, maxd, th)
        wn, wbi, wbd = oracle.stereo_match(rows, dict(x=lk["x"], y=lk["y"], octave=lk["octave"], desc=ld),
                                           dict(x=rk["x"], y=rk["y"], octave=rk["octave"], desc=rd), sf, mind, maxd, th)
        assert n == wn and np.array_equal(bi, wbi) and np.array_equal(bd, wbd)
    n, bi, bd = stereo_match(rows, (lk, ld), (rk, rd), sf, 0.0, 60.0, 75)
    assert n > 300
    d = lk["x"][bi >= 0] - rk["x"][bi[bi >= 0]]
    assert np.median(d) == pytest.approx(disp, abs=1.5)


def test_stereo_random_crowded(oracle):
    """Random keypoints on few rows (long candidate runs, many ties on equal descriptors)."""
    from orb_slam2_amd import stereo_match
    from orb_slam2_amd._lib import KP_DTYPE
    rng = np.random.default_rng(3)
    sf, _ = scale_tables()
    rows = 120
    for nl, nr in [(500, 900), (1, 1), (700, 3)]:
        def kps(n):
            k = np.zeros(n, KP_DTYPE)
            k["x"] = rng.uniform(0, 700, n).astype(np.float32)
            k["y"] = rng.uniform(16, rows - 16, n).astype(np.float32)
            k["octave"] = rng.integers(0, 8, n)
            return k
        lk, rk = kps(nl), kps(nr)
        base = rng.integers(0, 256, (8, 32), dtype=np.uint8)
        ld = flip_bits(rng, base[rng.integers(0, 8, nl)], 30)
        rd = flip_bits(rng, base[rng.integers(0, 8, nr)], 30)
        rd[::3] = base[0]   # exact duplicates: ties go to the lowest right index
        n, bi, bd = stereo_match(rows, (lk, ld), (rk, rd), sf, 0.0, 300.0, 75)
        wn, wbi, wbd = oracle.stereo_match(rows, dict(x=lk["x"], y=lk["y"], octave=lk["octave"], desc=ld),
                                           dict(x=rk["x"], y=rk["y"], octave=rk["octave"], desc=rd), sf, 0.0, 300.0, 75)
        assert n == wn and np.array_equal(bi, wbi) and np.array_equal(bd, wbd)


def _bow_scene(rng, n1, n2, nnodes=50, dup=True):
    sf, s2 = scale_tables()
    rows, cols = 480, 752
    x2, y2, o2, a2, d2 = random_frame_arrays(rng, n2, rows, cols)
    src = rng.integers(0, n2, n1)
    x1, y1, o1 = x2[src].copy(), y2[src].copy(), o2[src].copy()
    a1 = ((a2[src] + rng.normal(0, 8, n1)) % 360).astype(np.float32)
    d1 = flip_bits(rng, d2[src], 60)
    if dup:  # several side-1 features want the same side-2 feature; exact ties in a bucket
        d2[1::7] = d2[0::7][:len(d2[1::7])]
    node_of2 = rng.integers(0, nnodes, n2)
    node_of1 = np.where(rng.uniform(0, 1, n1) < 0.85, node_of2[src], rng.integers(0, nnodes + 5, n1))
    fv1 = {int(k) * 2: [int(i) for i in np.nonzero(node_of1 == k)[0]] for k in np.unique(node_of1)}
    fv2 = {int(k) * 2: [int(i) for i in np.nonzero(node_of2 == k)[0]] for k in np.unique(node_of2)}
    K = (np.float32(500), np.float32(500), np.float32(376), np.float32(240))
    b = (0, cols, 0, rows)
    return (kp_struct(x1, y1, o1, a1), d1, fv1), (kp_struct(x2, y2, o2, a2), d2, fv2), (sf, s2, K, b)


def _flat(fv):
    nodes = np.array(list(fv.keys()), np.int32); start = [0]; items = []
    for v in fv.values():
        items.extend(v); start.append(len(items))
    return dict(node=nodes, start=np.array(start, np.int32), items=np.array(items if items else [0], np.int32))


@pytest.mark.parametrize("seed,n1,n2,ratio,ori,nnodes", [(0, 800, 900, 0.7, True, 50), (1, 2000, 2000, 0.9, True, 12),
                                                          (2, 1500, 1200, 0.75, False, 200), (3, 300, 2500, 0.9, True, 3)])
def test_search_by_bow_both_overloads(oracle, seed, n1, n2, ratio, ori, nnodes):
    from orb_slam2_amd import Frame, KeyFrame, MapPoint, ORBmatcher
    rng = np.random.default_rng(seed)
    (k1, d1, fv1), (k2, d2, fv2), (sf, s2, K, b) = _bow_scene(rng, n1, n2, nnodes)
    kf1 = KeyFrame(k1, d1, sf, s2, K, b, feat_vec=fv1)
    has1 = rng.uniform(0, 1, n1) < 0.7
    bad1 = rng.uniform(0, 1, n1) < 0.05
    for i in np.nonzero(has1)[0]:
        kf1.mvpMapPoints[i] = MapPoint(np.zeros(3), d1[i])
        kf1.mvpMapPoints[i]._bad = bool(bad1[i])
    ok1 = (has1 & ~bad1).astype(np.uint8)
    m = ORBmatcher(ratio, ori)
    # overload 1: keyframe against a frame
    fr = Frame(k2, d2, sf, s2, K, b, feat_vec=fv2)
    out = []
    got = m.SearchByBoW(kf1, fr, out)
    wn, w12 = oracle.search_by_bow(dict(angle=k1["angle"], desc=d1, ok=ok1), dict(angle=k2["angle"], desc=d2, ok=np.ones(n2, np.uint8)),
                                   _flat(kf1.mFeatVec), _flat(fr.mFeatVec), ratio, 50, ori)
    assert got == wn and len(out) == n2
    assert wn > 30, "scene too poor"
    want = [None] * n2
    for i1 in range(n1):
        if w12[i1] >= 0:
            want[w12[i1]] = kf1.mvpMapPoints[i1]
    assert all(a is w for a, w in zip(out, want))
    # overload 2: keyframe against keyframe
    kf2 = KeyFrame(k2, d2, sf, s2, K, b, feat_vec=fv2)
    has2 = rng.uniform(0, 1, n2) < 0.8
    for i in np.nonzero(has2)[0]:
        kf2.mvpMapPoints[i] = MapPoint(np.zeros(3), d2[i])
    out12 = []
    got = m.SearchByBoW(kf1, kf2, out12)
    wn, w12 = oracle.search_by_bow(dict(angle=k1["angle"], desc=d1, ok=ok1), dict(angle=k2["angle"], desc=d2, ok=has2.astype(np.uint8)),
                                   _flat(kf1.mFeatVec), _flat(kf2.mFeatVec), ratio, 49, ori)
    assert got == wn and len(out12) == n1
    assert wn > 20, "scene too poor"
    assert all((a is None and w < 0) or (w >= 0 and a is kf2.mvpMapPoints[w]) for a, w in zip(out12, w12))


def test_search_by_bow_edge_cases(oracle):
    from orb_slam2_amd import Frame, KeyFrame, MapPoint, ORBmatcher
    rng = np.random.default_rng(8)
    (k1, d1, fv1), (k2, d2, fv2), (sf, s2, K, b) = _bow_scene(rng, 50, 60, 5)
    kf1 = KeyFrame(k1, d1, sf, s2, K, b, feat_vec=fv1)
    out = ["x"]
    assert ORBmatcher(0.7, True).SearchByBoW(kf1, Frame(k2, d2, sf, s2, K, b, feat_vec=fv2), out) == 0  # no map points
    assert out == [None] * 60
    for i in range(50):
        kf1.mvpMapPoints[i] = MapPoint(np.zeros(3), d1[i])
    assert ORBmatcher(0.7, True).SearchByBoW(kf1, Frame(k2, d2, sf, s2, K, b, feat_vec={}), out) == 0     # no common node
    assert ORBmatcher(0.7, True).SearchByBoW(kf1, Frame(k2[:0], d2[:0], sf, s2, K, b), out) == 0 and out == []


def _fuse_scene(rng, n_kp=1500, n_mp=900, rows=480, cols=752):
    """A keyframe at the origin and map points that project near its keypoints (several per
    keypoint now and then, so that Replace / AddMapPoint of one point matters to a later one)."""
    from orb_slam2_amd import KeyFrame, MapPoint
    sf, s2 = scale_tables()
    K = (np.float32(500), np.float32(500), np.float32(376), np.float32(240))
    x, y, o, a, d = random_frame_arrays(rng, n_kp, rows, cols)
    kf = KeyFrame(kp_struct(x, y, o, a), d, sf, s2, K, (0.0, float(cols), 0.0, float(rows)), np.eye(4, dtype=np.float32))
    for i in range(0, n_kp, 3):
        mp = MapPoint(np.zeros(3), d[i], observations=int(rng.integers(1, 6)))
        mp.mObservations[kf] = i
        kf.mvpMapPoints[i] = mp
    pts = []
    src = rng.integers(0, n_kp, n_mp)
    src[1::5] = src[0::5][:len(src[1::5])]  # duplicates: two points aim at the same keypoint
    for j in src:
        Z = np.float32(rng.uniform(3, 20))
        du, dv = rng.normal(0, 1.2, 2)
        Xc = np.array([(x[j] + du - K[2]) / K[0] * Z, (y[j] + dv - K[3]) / K[1] * Z, Z], np.float32)
        mp = MapPoint(Xc, flip_bits(rng, d[j:j + 1], 70)[0], observations=int(rng.integers(1, 6)))
        mp.mfMaxDistance = float(np.linalg.norm(Xc) * sf[o[j]] * rng.uniform(0.95, 1.05))
        mp.mfMinDistance = mp.mfMaxDistance / float(sf[7])
        mp.mNormalVector = (Xc / np.linalg.norm(Xc)).astype(np.float32)
        pts.append(mp)
    pts[5] = None
    pts[7]._bad = True
    return kf, pts


def test_fuse_against_sequential_restatement(oracle):
    """ORBmatcher::Fuse(pKF, vpMapPoints, th): the GPU scan of all points at once plus the host
    loop must leave the map in the state a point-by-point evaluation with the oracle's scan does."""
    import copy
    from orb_slam2_amd import ORBmatcher
    rng = np.random.default_rng(31)
    kf, pts = _fuse_scene(rng)
    kf_b, pts_b = copy.deepcopy((kf, pts))
    m = ORBmatcher(0.8, True)
    got = m.Fuse(kf, pts, 3.0)
    # sequential restatement on the copy (reference src/ORBmatcher.cc:1078-1202)
    k = kf_b.mvKeysUn
    train = dict(x=k["x"], y=k["y"], octave=k["octave"], desc=kf_b.mDescriptors)
    bounds = (kf_b.mnMinX, kf_b.mnMaxX, kf_b.mnMinY, kf_b.mnMaxY)
    want = 0
    for pMP in pts_b:
        if pMP is None or pMP.isBad() or pMP.IsInKeyFrame(kf_b):
            continue
        q = m._blank_queries(1)
        if not m._project_for_fuse(kf_b, kf_b.GetRotation(), kf_b.GetTranslation(), kf_b.GetCameraCenter(), pMP, 3.0, q, 0):
            continue
        bi, bd = oracle.window_best(bounds, train, dict(active=q["active"], u=q["u"], v=q["v"], r=q["r"], level=q["minl"],
                                                        desc=q["desc"]), kf_b.mvInvLevelSigma2, 5.99)
        if bi[0] >= 0 and bd[0] <= 50:
            other = kf_b.GetMapPoint(int(bi[0]))
            if other is not None:
                if not other.isBad():
                    if other.Observations() > pMP.Observations():
                        pMP.Replace(other)
                    else:
                        other.Replace(pMP)
            else:
                pMP.AddObservation(kf_b, int(bi[0]))
                kf_b.AddMapPoint(pMP, int(bi[0]))
            want += 1
    assert got == want and want > 200
    ida = {id(p): j for j, p in enumerate(pts) if p is not None}
    idb = {id(p): j for j, p in enumerate(pts_b) if p is not None}
    slot_a = [ida.get(id(p), -2 if p is not None else -1) for p in kf.mvpMapPoints]
    slot_b = [idb.get(id(p), -2 if p is not None else -1) for p in kf_b.mvpMapPoints]
    assert slot_a == slot_b
    assert [p is not None and p.isBad() for p in pts] == [p is not None and p.isBad() for p in pts_b]


def test_fuse_sim3_and_search_by_sim3(oracle):
    from orb_slam2_amd import KeyFrame, MapPoint, ORBmatcher
    rng = np.random.default_rng(32)
    kf, pts = _fuse_scene(rng, 1200, 700)
    pts = [p for p in pts if p is not None]
    m = ORBmatcher(0.8, True)
    S = np.eye(4, dtype=np.float32); S[:3, :3] *= np.float32(1.5)  # scale is divided out again
    for p in pts:
        p._pos = (p._pos / np.float32(1.5)).astype(np.float32)
        p.mfMaxDistance /= 1.5; p.mfMinDistance /= 1.5
    repl = [None] * len(pts)
    before = list(kf.mvpMapPoints)
    got = m.Fuse(kf, S, pts, 4.0, repl)
    k = kf.mvKeysUn
    train = dict(x=k["x"], y=k["y"], octave=k["octave"], desc=kf.mDescriptors)
    bounds = (kf.mnMinX, kf.mnMaxX, kf.mnMinY, kf.mnMaxY)
    want, wrepl, wadd = 0, [None] * len(pts), {}
    cur = list(before)
    Rcw = np.eye(3, dtype=np.float32); tcw = np.zeros(3, np.float32)
    for i, p in enumerate(pts):
        if p.isBad():
            continue
        q = m._blank_queries(1)
        if not m._project_for_fuse(kf, Rcw, tcw, np.zeros(3, np.float32), p, 4.0, q, 0):
            continue
        bi, bd = oracle.window_best(bounds, train, dict(active=q["active"], u=q["u"], v=q["v"], r=q["r"], level=q["minl"], desc=q["desc"]))
        if bi[0] >= 0 and bd[0] <= 50:
            if cur[bi[0]] is not None:
                if not cur[bi[0]].isBad():
                    wrepl[i] = cur[bi[0]]
            else:
                cur[bi[0]] = p
            want += 1
    assert got == want and want > 150
    assert all(a is b for a, b in zip(repl, wrepl)) and all(a is b for a, b in zip(kf.mvpMapPoints, cur))

    # SearchBySim3: two keyframes seeing the same points, identity relative transform
    sf, s2 = scale_tables()
    K = (np.float32(500), np.float32(500), np.float32(376), np.float32(240))
    n = 900
    x, y, o, a, d = random_frame_arrays(rng, n, 480, 752)
    kf1 = KeyFrame(kp_struct(x, y, o, a), d, sf, s2, K, (0.0, 752.0, 0.0, 480.0), np.eye(4, dtype=np.float32))
    perm = rng.permutation(n)
    kf2 = KeyFrame(kp_struct(x[perm] + rng.normal(0, 0.5, n).astype(np.float32), y[perm], o[perm], a[perm]), flip_bits(rng, d[perm], 30),
                   sf, s2, K, (0.0, 752.0, 0.0, 480.0), np.eye(4, dtype=np.float32))
    for i in range(n):
        for kfx, idx in ((kf1, i), (kf2, int(np.nonzero(perm == i)[0][0]))):
            if rng.uniform() < 0.7:
                Z = np.float32(rng.uniform(3, 20))
                kk = kfx.mvKeysUn[idx]
                Xc = np.array([(kk["x"] - K[2]) / K[0] * Z, (kk["y"] - K[3]) / K[1] * Z, Z], np.float32)
                mp = MapPoint(Xc, kfx.mDescriptors[idx])
                mp.mfMaxDistance = float(np.linalg.norm(Xc) * sf[kk["octave"]] * rng.uniform(0.95, 1.05))
                mp.mfMinDistance = mp.mfMaxDistance / float(sf[7])
                mp.mObservations[kfx] = idx
                kfx.mvpMapPoints[idx] = mp
    v12 = [None] * n
    for i in range(0, n, 11):
        j = int(np.nonzero(perm == i)[0][0])
        if kf2.mvpMapPoints[j] is not None:
            v12[i] = kf2.mvpMapPoints[j]
    pre = list(v12)
    got = m.SearchBySim3(kf1, kf2, v12, 1.0, np.eye(3, dtype=np.float32), np.zeros(3, np.float32), 7.5)

    def one_way(src, dst, done):
        kd = dst.mvKeysUn
        tr = dict(x=kd["x"], y=kd["y"], octave=kd["octave"], desc=dst.mDescriptors)
        out = [-1] * src.N
        for i, p in enumerate(src.mvpMapPoints):
            if p is None or done[i] or p.isBad():
                continue
            q = m._blank_queries(1)
            pc = p.GetWorldPos()
            if not m._project_for_fuse(dst, None, None, None, p, 7.5, q, 0, camera_dist=pc):
                continue
            bi, bd = oracle.window_best((0.0, 752.0, 0.0, 480.0), tr, dict(active=q["active"], u=q["u"], v=q["v"], r=q["r"],
                                                                           level=q["minl"], desc=q["desc"]))
            if bi[0] >= 0 and bd[0] <= 100:
                out[i] = int(bi[0])
        return out
    done1 = [p is not None for p in pre]
    done2 = [False] * n
    for p in pre:
        if p is not None:
            done2[p.GetIndexInKeyFrame(kf2)] = True
    m1, m2 = one_way(kf1, kf2, done1), one_way(kf2, kf1, done2)
    want = sum(1 for i in range(n) if m1[i] >= 0 and m2[m1[i]] == i)
    assert got == want and want > 100
    for i in range(n):
        if m1[i] >= 0 and m2[m1[i]] == i:
            assert v12[i] is kf2.mvpMapPoints[m1[i]]
        else:
            assert v12[i] is pre[i]


def test_window_best_raw(oracle):
    """orb_window_best through the C ABI against the oracle, crowded windows and the chi2 gate."""
    from orb_slam2_amd._lib import FrameBounds, check, lib, ptr
    rng = np.random.default_rng(33)
    train, q, bounds = _windowed_case(rng, 2500, 3000)
    sf, s2 = scale_tables()
    inv = (np.float32(1) / s2).astype(np.float32)
    lvl = np.clip(q["maxl"], 0, 7).astype(np.int32)
    for chi2 in (0.0, 5.99):
        wbi, wbd = oracle.window_best(bounds, train, dict(active=q["active"], u=q["u"], v=q["v"], r=q["r"], level=lvl, desc=q["desc"]), inv, chi2)
        nq = len(lvl)
        bi, bd = np.zeros(nq, np.int32), np.zeros(nq, np.int32)
        f = lambda a, t: np.ascontiguousarray(a, t)
        t = [f(train["x"], "f4"), f(train["y"], "f4"), f(train["octave"], "i4"), f(train["desc"], "u1")]
        qa = [f(q["active"], "u1"), f(q["u"], "f4"), f(q["v"], "f4"), f(q["r"], "f4"), lvl, f(q["desc"], "u1")]
        check(lib().orb_window_best(0, FrameBounds(*bounds), len(t[0]), *[ptr(a) for a in t], ptr(inv), 8, chi2, nq,
                                    *[ptr(a) for a in qa], ptr(bi), ptr(bd)))
        assert np.array_equal(bi, wbi) and np.array_equal(bd, wbd) and (bi >= 0).sum() > 500


@pytest.mark.parametrize("seed,n,win,ratio,ori", [(0, 1500, 100, 0.9, True), (1, 2500, 40, 0.8, True), (2, 800, 100, 0.95, False),
                                                   (3, 2000, 250, 0.9, True)])
def test_search_for_initialization(oracle, seed, n, win, ratio, ori):
    """Level-0 window search with match stealing and the matched-distance table: near-duplicate
    descriptors in F1 make several features compete for the same F2 feature."""
    from orb_slam2_amd import Frame, ORBmatcher
    rng = np.random.default_rng(seed)
    sf, s2 = scale_tables()
    rows, cols = 480, 752
    K = (np.float32(500), np.float32(500), np.float32(376), np.float32(240))
    b = (0.0, float(cols), 0.0, float(rows))
    x2, y2, o2, a2, d2 = random_frame_arrays(rng, n, rows, cols)
    o2[rng.uniform(0, 1, n) < 0.6] = 0
    src = rng.integers(0, n, n)
    src[1::4] = src[0::4][:len(src[1::4])]            # competing features
    x1 = (x2[src] + rng.normal(0, 8, n)).astype(np.float32)
    y1 = (y2[src] + rng.normal(0, 8, n)).astype(np.float32)
    o1 = o2[src].copy()
    a1 = ((a2[src] + rng.normal(0, 8, n)) % 360).astype(np.float32)
    d1 = flip_bits(rng, d2[src], 50)
    F1 = Frame(kp_struct(x1, y1, o1, a1), d1, sf, s2, K, b)
    F2 = Frame(kp_struct(x2, y2, o2, a2), d2, sf, s2, K, b)
    prev = np.stack([x1, y1], 1).astype(np.float32)
    prev0 = prev.copy()
    out = []
    got = ORBmatcher(ratio, ori).SearchForInitialization(F1, F2, prev, out, win)
    wn, w12 = oracle.search_for_initialization(b, dict(octave=o1, angle=a1, desc=d1), prev0,
                                               dict(x=x2, y=y2, octave=o2, angle=a2, desc=d2), win, ratio, ori)
    assert got == wn and np.array_equal(np.array(out, np.int32), w12)
    assert wn > 100, "scene too poor"
    hit = w12 >= 0
    assert np.array_equal(prev[hit, 0], x2[w12[hit]]) and np.array_equal(prev[~hit], prev0[~hit])


@pytest.mark.parametrize("nq,ndb,seed", [(1, 1, 0), (37, 5, 1), (512, 2048, 2), (513, 2079, 3), (1000, 33, 4),
                                          (3000, 10000, 5), (64, 70001, 6),
                                          (300, (1 << 21) + 70000, 7)])  # the last: more rows than one key chunk holds
def test_knn2_matrix_core_kernel_equals_popcount_kernel(oracle, nq, ndb, seed):
    """The two kNN-2 kernels (vector popcount / int8 matrix cores) return the same integers,
    ties included, and both equal the oracle."""
    torch = pytest.importorskip("torch")
    import ctypes as C
    from orb_slam2_amd import _lib
    rng = np.random.default_rng(seed)
    db = rng.integers(0, 256, (ndb, 32), dtype=np.uint8)
    if ndb > 8:
        db[ndb // 2] = db[1]            # exact duplicates: the lower row must win
        db[ndb - 1] = db[0]
    q = db[rng.integers(0, ndb, nq)].copy()
    flips = rng.integers(0, 2, nq).astype(bool)
    q[flips] = flip_bits(rng, q[flips], 40) if flips.any() else q[flips]
    q[::7] = rng.integers(0, 256, (len(q[::7]), 32), dtype=np.uint8)
    if nq >= 8 and ndb > 8:
        q[1:5] = db[[0, 1, ndb // 2, ndb - 1]]  # the duplicated rows themselves: distance 0 twice
    d_q, d_db = torch.from_numpy(q).cuda(), torch.from_numpy(db).cuda()
    outs = []
    for algo in (1, 2):
        res = torch.full((3, nq), -7, dtype=torch.int32, device="cuda")
        acc = torch.zeros(nq, dtype=torch.uint8, device="cuda")
        _lib.check(_lib.lib().orb_knn2_device_algo(0, C.c_void_p(d_q.data_ptr()), nq, C.c_void_p(d_db.data_ptr()), ndb, 1000, 0.7,
                                                   C.c_void_p(res[0].data_ptr()), C.c_void_p(res[1].data_ptr()),
                                                   C.c_void_p(res[2].data_ptr()), C.c_void_p(acc.data_ptr()), None, algo))
        torch.cuda.synchronize()
        outs.append((res.cpu().numpy(), acc.cpu().numpy()))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    wb, ws, wi, wa = oracle.knn2_bruteforce(q, db, 0.7)
    assert np.array_equal(outs[1][0][0], wb) and np.array_equal(outs[1][0][1], ws)
    assert np.array_equal(outs[1][0][2], wi + 1000) and np.array_equal(outs[1][1], wa)


def test_knn2_full_size_properties(oracle):
    """The 1M x 1M kNN-2 of the benchmark, checked through properties that need no full CPU scan:
    planted queries find their source row at the planted distance; a sample of queries equals
    the oracle; the popcount kernel agrees on a sample; eight ascending shards merged equal the
    single scan bit for bit (the multi-GPU decomposition, run here on one GPU)."""
    torch = pytest.importorskip("torch")
    import ctypes as C
    from orb_slam2_amd import _lib
    from orb_slam2_amd.synth import synth_descriptors
    L = _lib.lib()
    n = 1000000
    db, q = synth_descriptors(n, n, seed=1)
    # the generator's planted half: source rows and the distance that is left after the flips
    rq = np.random.default_rng(2)
    rq.integers(0, 256, size=(n, 32), dtype=np.uint8)
    src = rq.integers(0, n, size=n // 2)
    planted_d = np.unpackbits(q[:n // 2] ^ db[src], axis=1).sum(1).astype(np.int32)
    assert planted_d.max() <= 40
    d_q, d_db = torch.from_numpy(q).cuda(), torch.from_numpy(db).cuda()

    def scan(qptr, nq, lo, hi, algo=0):
        res = torch.full((3, nq), -7, dtype=torch.int32, device="cuda")
        acc = torch.zeros(nq, dtype=torch.uint8, device="cuda")
        _lib.check(L.orb_knn2_device_algo(0, C.c_void_p(qptr), nq, C.c_void_p(d_db.data_ptr() + 32 * lo), hi - lo, lo, 0.7,
                                          C.c_void_p(res[0].data_ptr()), C.c_void_p(res[1].data_ptr()),
                                          C.c_void_p(res[2].data_ptr()), C.c_void_p(acc.data_ptr()), None, algo))
        return res, acc

    full, facc = scan(d_q.data_ptr(), n, 0, n)
    torch.cuda.synchronize()
    best, second, idx = (x.cpu().numpy() for x in full)
    acc = facc.cpu().numpy()
    # planted queries: nothing among a million random rows comes near 40 bits
    assert np.array_equal(best[:n // 2], planted_d) and np.array_equal(idx[:n // 2], src)
    assert (second >= best).all() and (idx >= 0).all() and (idx < n).all()
    assert np.array_equal(acc, (best.astype(np.float32) < np.float32(0.7) * second.astype(np.float32)).astype(np.uint8))
    # the distance reported is the distance to the row reported
    sample = np.random.default_rng(5).integers(0, n, 4096)
    assert np.array_equal(np.unpackbits(q[sample] ^ db[idx[sample]], axis=1).sum(1), best[sample])
    # oracle on 48 queries of each half
    pick = np.concatenate([np.arange(48), n // 2 + np.arange(48)])
    wb, ws, wi, wa = oracle.knn2_bruteforce(q[pick], db, 0.7)
    assert np.array_equal(best[pick], wb) and np.array_equal(second[pick], ws) and np.array_equal(idx[pick], wi)
    assert np.array_equal(acc[pick], wa)
    # the popcount kernel on 4096 consecutive queries across the middle
    lo_q = n // 2 - 2048
    pres, pacc = scan(d_q.data_ptr() + 32 * lo_q, 4096, 0, n, algo=1)
    torch.cuda.synchronize()
    assert torch.equal(pres, full[:, lo_q:lo_q + 4096]) and torch.equal(pacc, facc[lo_q:lo_q + 4096])
    # eight shards, merged
    shards = [scan(d_q.data_ptr(), n, n * r // 8, n * (r + 1) // 8)[0] for r in range(8)]
    g = torch.stack(shards, 1).contiguous()  # [3][8][n]
    out = torch.zeros((3, n), dtype=torch.int32, device="cuda")
    macc = torch.zeros(n, dtype=torch.uint8, device="cuda")
    _lib.check(L.orb_knn2_merge_device(0, 8, n, C.c_void_p(g[0].data_ptr()), C.c_void_p(g[1].data_ptr()),
                                       C.c_void_p(g[2].data_ptr()), 0.7, C.c_void_p(out[0].data_ptr()),
                                       C.c_void_p(out[1].data_ptr()), C.c_void_p(out[2].data_ptr()),
                                       C.c_void_p(macc.data_ptr()), None))
    torch.cuda.synchronize()
    assert torch.equal(out, full) and torch.equal(macc, facc)


@pytest.mark.parametrize("bounds", [(-3.7, 755.2, -2.4, 482.9), (3.6, 748.9, 2.7, 477.2)])
def test_keyframe_grid_uses_truncated_origin(oracle, bounds):
    """KeyFrame::GetFeaturesInArea (reference src/KeyFrame.cc:732-776) subtracts the KeyFrame's
    const int mnMinX / mnMinY (include/KeyFrame.h:411-414) on the grid the Frame built with float
    bounds; the *_keyframe entry points against the oracle's KeyFrame variant, with the bounds of
    a distorted camera, and the Frame forms on the same bounds beside them."""
    import ctypes as C
    from orb_slam2_amd._lib import FrameBounds, check, lib, ptr
    rng = np.random.default_rng(int(abs(bounds[0]) * 10))
    train, q, _ = _windowed_case(rng, 2200, 2600)
    # spread part of the keypoints and queries right up to the (undistorted) image bounds
    edge = rng.integers(0, 2200, 500)
    train["x"][edge] = rng.uniform(bounds[0], bounds[1], 500).astype(np.float32)
    train["y"][edge] = rng.uniform(bounds[2], bounds[3], 500).astype(np.float32)
    qe = rng.integers(0, 2600, 800)
    src = rng.integers(0, 2200, 800)
    q["u"][qe] = (train["x"][src] + rng.uniform(-5, 5, 800)).astype(np.float32)
    q["v"][qe] = (train["y"][src] + rng.uniform(-5, 5, 800)).astype(np.float32)
    q["desc"][qe] = flip_bits(rng, train["desc"][src], 60)
    f = lambda a, t: np.ascontiguousarray(a, t)
    nt, nq = 2200, 2600
    targs = [f(train["x"], "f4"), f(train["y"], "f4"), f(train["octave"], "i4"), f(train["angle"], "f4"),
             f(train["desc"], "u1"), f(train["taken"], "u1")]
    qa = [f(q["active"], "u1"), f(q["u"], "f4"), f(q["v"], "f4"), f(q["r"], "f4"), f(q["minl"], "i4"), f(q["maxl"], "i4"),
          f(q["angle"], "f4"), f(q["desc"], "u1")]
    L = lib()
    for kf, fn in ((False, L.orb_match_windowed), (True, L.orb_match_windowed_keyframe)):
        want_n, want_m = oracle.match_windowed(bounds, train, q, 50, True, keyframe=kf)
        match = np.zeros(nt, np.int32); nm = C.c_int()
        check(fn(0, FrameBounds(*bounds), nt, *[ptr(a) for a in targs], nq, *[ptr(a) for a in qa], 50, 1, ptr(match),
                 C.byref(nm)))
        assert nm.value == want_n and np.array_equal(match, want_m), kf
        assert want_n > 100
    sf, s2 = scale_tables()
    inv = (np.float32(1) / s2).astype(np.float32)
    lvl = np.clip(q["maxl"], 0, 7).astype(np.int32)
    qd = dict(active=q["active"], u=q["u"], v=q["v"], r=q["r"], level=lvl, desc=q["desc"])
    tb = [f(train["x"], "f4"), f(train["y"], "f4"), f(train["octave"], "i4"), f(train["desc"], "u1")]
    qb = [f(q["active"], "u1"), f(q["u"], "f4"), f(q["v"], "f4"), f(q["r"], "f4"), lvl, f(q["desc"], "u1")]
    for kf, fn in ((False, L.orb_window_best), (True, L.orb_window_best_keyframe)):
        wbi, wbd = oracle.window_best(bounds, train, qd, inv, 5.99, keyframe=kf)
        bi, bd = np.zeros(nq, np.int32), np.zeros(nq, np.int32)
        check(fn(0, FrameBounds(*bounds), nt, *[ptr(a) for a in tb], ptr(inv), 8, 5.99, nq, *[ptr(a) for a in qb], ptr(bi),
                 ptr(bd)))
        assert np.array_equal(bi, wbi) and np.array_equal(bd, wbd), kf


@pytest.mark.gpu
def test_knn2_matrix_core_kernel_across_chunks_and_extreme_rows():
    """The matrix-core kernel folds its running results every 2^21 database rows and starts its
    accumulators from the rows' set-bit counts: a database of 2^21 + 4999 rows (not a multiple of
    32) that holds all-zero and all-one rows, duplicates on both sides of the chunk boundary and
    queries that are all zero / all one, against the popcount kernel."""
    torch = pytest.importorskip("torch")
    import ctypes as C
    from orb_slam2_amd import _lib
    rng = np.random.default_rng(77)
    ndb, nq = (1 << 21) + 4999, 777
    db = rng.integers(0, 256, (ndb, 32), dtype=np.uint8)
    db[5] = 0; db[6] = 255; db[(1 << 21) + 3] = 0; db[(1 << 21) + 4] = 255
    db[(1 << 21) + 100] = db[100]           # the same row in both chunks: the lower index wins
    db[ndb - 1] = db[200]
    q = db[rng.integers(0, ndb, nq)].copy()
    q[0] = 0; q[1] = 255; q[2] = db[100]; q[3] = db[200]; q[4] = db[(1 << 21) + 7]
    q[5::9] = rng.integers(0, 256, (len(q[5::9]), 32), dtype=np.uint8)
    d_q, d_db = torch.from_numpy(q).cuda(), torch.from_numpy(db).cuda()
    outs = []
    for algo in (1, 2):
        res = torch.full((3, nq), -7, dtype=torch.int32, device="cuda")
        acc = torch.zeros(nq, dtype=torch.uint8, device="cuda")
        _lib.check(_lib.lib().orb_knn2_device_algo(0, C.c_void_p(d_q.data_ptr()), nq, C.c_void_p(d_db.data_ptr()), ndb, 0, 0.7,
                                                   C.c_void_p(res[0].data_ptr()), C.c_void_p(res[1].data_ptr()),
                                                   C.c_void_p(res[2].data_ptr()), C.c_void_p(acc.data_ptr()), None, algo))
        torch.cuda.synchronize()
        outs.append((res.cpu().numpy(), acc.cpu().numpy()))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    best, second, idx = outs[1][0]
    assert best[0] == 0 and idx[0] == 5 and second[0] == 0         # all-zero query: rows 5 and 2^21 + 3
    assert best[1] == 0 and idx[1] == 6 and second[1] == 0
    assert best[2] == 0 and idx[2] == 100 and second[2] == 0       # duplicate across the chunk boundary
    assert best[3] == 0 and idx[3] == 200 and second[3] == 0
    assert best[4] == 0 and idx[4] == (1 << 21) + 7
