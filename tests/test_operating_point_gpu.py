"""The reference's own shipped operating point and the matchers' size limits.

The configuration the reference ships is 1920x1440 images with ORBextractor.nFeatures 4000
(reference Examples/Monocular/shenzhen_simple.yaml:28-29,36) and Tracking builds a second extractor
with 2 * nFeatures = 8000 for initialisation (src/Tracking.cc:174-182), so SearchForInitialization
(src/ORBmatcher.cc:543) and SearchByProjection(Cur, Last) (:1633) see frames of about 4000 and 8000
keypoints.  Frames of more than about 7900 keypoints (6500 for SearchForInitialization) no longer
fit the matchers' on-chip grid and take the second form of the kernel; both forms are compared with
the CPU oracle here, at the boundary and well beyond it."""
import numpy as np
import pytest

from helpers import flip_bits, kp_struct, projection_scene, random_frame_arrays, scale_tables
from test_extract_gpu import check_against, check_stages

pytestmark = pytest.mark.gpu

ROWS, COLS = 1440, 1920


@pytest.mark.parametrize("nf", [4000, 8000])
def test_extract_shipped_configuration(oracle, nf):
    """1440 x 1920, 4000 and 8000 features: one frame stage by stage (pyramid bytes, FAST candidates,
    quad-tree order, blurred levels), then a batch of ten frames, every frame against the oracle and
    the first and last of them stage by stage."""
    from orb_slam2_amd import ORBextractor
    from orb_slam2_amd.synth import synth_image
    ex = ORBextractor(nf, 1.2, 8, 20, 7)
    ref = oracle.Extractor(nf, 1.2, 8, 20, 7)
    img = synth_image(ROWS, COLS, seed=60)
    kps, desc = ex(img)
    rk, rd = ref(img)
    assert len(rk) >= nf * 0.9
    check_against(kps, desc, rk, rd)
    check_stages(ex, ref, 8)
    batch = np.stack([synth_image(ROWS, COLS, seed=61 + k) for k in range(10)])
    bk, bd = ex.extract_batch(batch)
    for f in range(10):
        rk, rd = ref(batch[f])
        check_against(bk[f], bd[f], rk, rd)
        if f in (0, 9):
            check_stages(ex, ref, 8, frame=f)


def _projection_case(oracle, seed, nl, nc, clusters, ori=True):
    from orb_slam2_amd import Frame, MapPoint, ORBmatcher
    rng = np.random.default_rng(seed)
    sc = projection_scene(rng, nl, nc, ROWS, COLS, conflict_clusters=clusters)
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
    for i2 in range(nc):
        if Cu["obs"][i2] >= 0:
            cur.mvpMapPoints[i2] = MapPoint(np.zeros(3), np.zeros(32, np.uint8), Cu["obs"][i2])
    nm, match = ORBmatcher(0.9, ori).search_by_projection_frames_batch([cur], [last], 7.0)
    assert nm[0] == want_n and np.array_equal(match[0, :nc], want_match)
    assert want_n > min(nl, nc) // 10
    return sc, want_n, want_match


def _first_size_of_the_large_form(nl):
    """Smallest number of current-frame keypoints that no longer fits on chip, found through the
    scratch the device form asks for."""
    from orb_slam2_amd import _lib
    L = _lib.lib()
    base = L.orb_match_scratch_bytes_for(1, nl, 16)
    lo, hi = 16, 65535
    assert L.orb_match_scratch_bytes_for(1, nl, hi) > base
    while hi - lo > 1:
        mid = (lo + hi) // 2
        if L.orb_match_scratch_bytes_for(1, nl, mid) > base:
            hi = mid
        else:
            lo = mid
    return hi


@pytest.mark.parametrize("nl,nc,clusters", [(4000, 4000, 30), (8000, 8000, 60), (8000, 12000, 200), (3000, 30000, 0),
                                             (500, 65535, 0), (70000, 3000, 40)])  # (the last: more queries than 16 bits count)
def test_search_by_projection_at_and_beyond_the_shipped_size(oracle, nl, nc, clusters):
    _projection_case(oracle, 1000 + nc, nl, nc, clusters)


def test_search_by_projection_either_side_of_the_on_chip_limit(oracle):
    b = _first_size_of_the_large_form(2000)
    assert 7800 < b < 8600, b
    for nc in (b - 1, b):
        _projection_case(oracle, nc, 2000, nc, 50)


@pytest.mark.parametrize("seed,n,win", [(0, 4000, 100), (1, 8000, 100), (2, 8000, 300), (3, 20000, 60)])
def test_search_for_initialization_at_the_shipped_size(oracle, seed, n, win):
    """SearchForInitialization on frames of the initialisation extractor's size (2 x nFeatures):
    8000 keypoints are already beyond what the on-chip form holds together with the owner table."""
    from orb_slam2_amd import Frame, ORBmatcher
    rng = np.random.default_rng(seed)
    sf, s2 = scale_tables()
    K = (np.float32(1500), np.float32(1500), np.float32(COLS / 2), np.float32(ROWS / 2))
    b = (0.0, float(COLS), 0.0, float(ROWS))
    x2, y2, o2, a2, d2 = random_frame_arrays(rng, n, ROWS, COLS)
    o2[rng.uniform(0, 1, n) < 0.6] = 0
    src = rng.integers(0, n, n)
    src[1::4] = src[0::4][:len(src[1::4])]            # features that compete for one keypoint
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
    got = ORBmatcher(0.9, True).SearchForInitialization(F1, F2, prev, out, win)
    wn, w12 = oracle.search_for_initialization(b, dict(octave=o1, angle=a1, desc=d1), prev0,
                                               dict(x=x2, y=y2, octave=o2, angle=a2, desc=d2), win, 0.9, True)
    assert got == wn and np.array_equal(np.array(out, np.int32), w12)
    assert wn > n // 20, "scene too poor"


def test_map_point_projection_and_triangulation_with_large_frames(oracle):
    """The other modes of the kernel in its large form: SearchByProjection(F, MapPoints) with 12000
    frame keypoints and the window scan behind Fuse / SearchBySim3 with 15000."""
    from orb_slam2_amd import _lib
    from orb_slam2_amd._lib import FrameBounds, check, ptr
    L = _lib.lib()
    rng = np.random.default_rng(9)
    sf, _ = scale_tables()
    nf, nmp = 12000, 6000
    x, y, octv, ang, desc = random_frame_arrays(rng, nf, ROWS, COLS)
    src = rng.integers(0, nf, nmp)
    frame = dict(x=x, y=y, octave=octv, desc=desc, obs=np.where(rng.uniform(0, 1, nf) < 0.1, 1, -1).astype(np.int32))
    mps = dict(track=(rng.uniform(0, 1, nmp) < 0.9).astype(np.uint8), level=octv[src].astype(np.int32),
               viewcos=rng.uniform(0.99, 1.0, nmp).astype(np.float32),
               projx=(x[src] + rng.normal(0, 2, nmp)).astype(np.float32), projy=(y[src] + rng.normal(0, 2, nmp)).astype(np.float32),
               desc=flip_bits(rng, desc[src], 40), obs=rng.integers(0, 3, nmp).astype(np.int32))
    b = (0.0, float(COLS), 0.0, float(ROWS))
    wn, wm = oracle.search_by_projection_mappoints(b, sf, frame, mps, 3.0, 0.8)
    fm = np.zeros(nf, np.int32)
    nm = np.zeros(1, np.int32)
    check(L.orb_search_by_projection_mappoints(0, FrameBounds(*b), ptr(sf), 8, nf, ptr(x), ptr(y), ptr(octv), ptr(desc),
                                               ptr(frame["obs"]), nmp, ptr(mps["track"]), ptr(mps["level"]), ptr(mps["viewcos"]),
                                               ptr(mps["projx"]), ptr(mps["projy"]), ptr(mps["desc"]), ptr(mps["obs"]), 3.0, 0.8,
                                               ptr(fm), ptr(nm)))
    assert int(nm[0]) == wn and np.array_equal(fm, wm) and wn > 1000
    # independent best candidate per query (Fuse / SearchBySim3), 15000 train keypoints
    nt, nq = 15000, 5000
    tx, ty, toct, _, tdesc = random_frame_arrays(rng, nt, ROWS, COLS)
    qs = rng.integers(0, nt, nq)
    queries = dict(active=np.ones(nq, np.uint8), u=(tx[qs] + rng.normal(0, 3, nq)).astype(np.float32),
                   v=(ty[qs] + rng.normal(0, 3, nq)).astype(np.float32), r=np.full(nq, 12.0, np.float32),
                   level=toct[qs].astype(np.int32), desc=flip_bits(rng, tdesc[qs], 30))
    wi, wd = oracle.window_best(b, dict(x=tx, y=ty, octave=toct, desc=tdesc), queries)
    bi, bd = np.zeros(nq, np.int32), np.zeros(nq, np.int32)
    check(L.orb_window_best(0, FrameBounds(*b), nt, ptr(tx), ptr(ty), ptr(toct), ptr(tdesc), None, 8, 0.0, nq,
                            ptr(queries["active"]), ptr(queries["u"]), ptr(queries["v"]), ptr(queries["r"]),
                            ptr(queries["level"]), ptr(queries["desc"]), ptr(bi), ptr(bd)))
    assert np.array_equal(bi, wi) and np.array_equal(bd, wd) and (wi >= 0).sum() > 1000


def test_device_form_with_large_frames_needs_the_larger_scratch(oracle):
    """orb_search_by_projection_frames_device on frames beyond the on-chip limit: with the scratch
    orb_match_scratch_bytes_for asks for it equals the host-buffer form (and the oracle); with the
    plain size it is refused, and so is the allocation-free form."""
    import torch
    from orb_slam2_amd import _lib
    from orb_slam2_amd._lib import FrameBounds, KP_DTYPE
    L = _lib.lib()
    dev = torch.device("cuda", 0)
    npairs, nl, nc = 2, 3000, 11000
    rng = np.random.default_rng(3)
    wants, host = [], {}
    lk = np.zeros((npairs, nl), KP_DTYPE); ck = np.zeros((npairs, nc), KP_DTYPE)
    lobs = np.zeros((npairs, nl), np.int32); lxw = np.zeros((npairs, nl, 3), np.float32)
    ldesc = np.zeros((npairs, nl, 32), np.uint8); cdesc = np.zeros((npairs, nc, 32), np.uint8)
    cobs = np.zeros((npairs, nc), np.int32); tcw = np.zeros((npairs, 16), np.float32)
    for p in range(npairs):
        sc = projection_scene(rng, nl, nc, ROWS, COLS, conflict_clusters=40)
        wants.append(oracle.search_by_projection_frames(sc["Tcw"], sc["K"], sc["bounds"], sc["sf"], sc["last"], sc["cur"], 7.0, True))
        la, cu = sc["last"], sc["cur"]
        lk[p] = kp_struct(np.zeros(nl), np.zeros(nl), la["octave"], la["angle"])
        ck[p] = kp_struct(cu["x"], cu["y"], cu["octave"], cu["angle"])
        lobs[p], lxw[p], ldesc[p], cdesc[p], cobs[p], tcw[p] = la["mp_obs"], la["xw"], la["mp_desc"], cu["desc"], cu["obs"], sc["Tcw"].reshape(16)
    d = {k: torch.from_numpy(v).to(dev) for k, v in dict(tcw=tcw, lk=lk.view(np.uint8), lobs=lobs, lxw=lxw, ldesc=ldesc,
                                                         ck=ck.view(np.uint8), cdesc=cdesc, cobs=cobs).items()}
    d_nl = torch.full((npairs,), nl, dtype=torch.int32, device=dev)
    d_nc = torch.full((npairs,), nc, dtype=torch.int32, device=dev)
    K, sf = sc["K"], sc["sf"]
    head = [0, npairs, nl, nc, d["tcw"].data_ptr(), float(K[0]), float(K[1]), float(K[2]), float(K[3]), FrameBounds(*sc["bounds"]),
            sf.ctypes.data, 8, d_nl.data_ptr(), d["lk"].data_ptr(), d["lobs"].data_ptr(), d["lxw"].data_ptr(), d["ldesc"].data_ptr(),
            d_nc.data_ptr(), d["ck"].data_ptr(), d["cdesc"].data_ptr(), d["cobs"].data_ptr(), 7.0, 1]
    m = torch.full((npairs, nc), -9999, dtype=torch.int32, device=dev)
    nm = torch.zeros(npairs, dtype=torch.int32, device=dev)
    small, full = L.orb_match_scratch_bytes(npairs, nl), L.orb_match_scratch_bytes_for(npairs, nl, nc)
    assert full > small
    scratch = torch.zeros(full, dtype=torch.uint8, device=dev)
    assert L.orb_search_by_projection_frames_device(*head, scratch.data_ptr(), small, m.data_ptr(), nm.data_ptr(), None) == _lib.ERR_CAPACITY
    assert b"orb_match_scratch_bytes_for" in L.orb_last_error()
    assert L.orb_search_by_projection_frames_noalloc(*head, m.data_ptr(), nm.data_ptr(), None) == _lib.ERR_CAPACITY
    _lib.check(L.orb_search_by_projection_frames_device(*head, scratch.data_ptr(), full, m.data_ptr(), nm.data_ptr(), None))
    torch.cuda.synchronize()
    for p in range(npairs):
        assert int(nm[p]) == wants[p][0] and np.array_equal(m[p].cpu().numpy(), wants[p][1]), p


def test_the_other_searches_beyond_the_on_chip_grid(oracle):
    """SearchForTriangulation, the windowed SearchByProjection overloads and SearchByBoW on frames of
    9 000 to 12 000 keypoints take the kernel form with part of its tables in global memory, like the
    frame-to-frame search above: the parity tests of tests/test_match_gpu.py at those sizes."""
    import test_match_gpu as T
    T.test_search_for_triangulation(oracle, 11, 9000, 10000, True)
    T.test_match_windowed_vs_oracle(oracle, 12, 12000, 6000, 100, True)
    T.test_search_by_bow_both_overloads(oracle, 13, 9000, 9500, 0.8, True, 300)
    T.test_search_by_projection_mappoints(oracle, 14, 5000, 11000, 3.0, 0.8)
