"""MapPoint::ComputeDistinctiveDescriptors on the GPU against the oracle (bit-exact index and
median), through the C ABI and through the MapPoint mirror."""
import numpy as np
import pytest

from oracle import oracle
from orb_slam2_amd import _lib
from helpers import flip_bits, kp_struct, scale_tables

pytestmark = pytest.mark.gpu


def _call(start, desc):
    start = np.asarray(start, np.int32)
    n = len(start) - 1
    best, med = np.empty(max(n, 1), np.int32), np.empty(max(n, 1), np.int32)
    _lib.check(_lib.lib().orb_distinctive_descriptors(0, n, _lib.ptr(start), _lib.ptr(desc), _lib.ptr(best), _lib.ptr(med)))
    return best[:n], med[:n]


def _clusters(rng, sizes, max_flips=40):
    """Per point: noisy copies of one descriptor, with a few exact duplicates (equal medians)."""
    rows, start = [], [0]
    for n in sizes:
        if n:
            d = flip_bits(rng, np.repeat(rng.integers(0, 256, (1, 32), dtype=np.uint8), n, 0), max_flips)
            if n > 3:
                d[n // 2] = d[0]
                d[n - 1] = d[1]
            rows.append(d)
        start.append(start[-1] + n)
    desc = np.concatenate(rows) if rows else np.zeros((0, 32), np.uint8)
    return np.array(start, np.int32), np.ascontiguousarray(desc)


def test_many_points_all_sizes():
    rng = np.random.default_rng(0)
    sizes = list(range(0, 70)) + [1, 2, 2, 3, 64, 64, 65, 128, 255, 256, 257, 300, 1000] + list(rng.integers(1, 40, 3000))
    rng.shuffle(sizes)
    start, desc = _clusters(rng, sizes)
    best, med = _call(start, desc)
    ties = 0
    for p, n in enumerate(sizes):
        wi, wm = oracle.distinctive_descriptor(desc[start[p]:start[p + 1]])
        assert (best[p], med[p]) == (wi, wm), (p, n)
        ties += wi > 0
    assert ties > 1000, "scene too poor: the first descriptor always wins"


def test_identical_and_far_descriptors():
    d = np.zeros((5, 32), np.uint8)
    best, med = _call([0, 5], d)
    assert (best[0], med[0]) == (0, 0)             # all equal: the first one
    d[0] = 255                                     # one outlier in front: median of row 0 is 256
    best, med = _call([0, 5], d)
    assert (best[0], med[0]) == (1, 0)
    two = np.zeros((2, 32), np.uint8); two[1] = 255
    best, med = _call([0, 2], two)                 # N = 2: index (int)(0.5) = 0 -> the distance to itself
    assert (best[0], med[0]) == (0, 0)
    assert _call([0, 0, 0], np.zeros((0, 32), np.uint8))[0].tolist() == [-1, -1]


def test_mappoint_mirror():
    from orb_slam2_amd import KeyFrame, MapPoint, compute_distinctive_descriptors
    rng = np.random.default_rng(3)
    sf, s2 = scale_tables()
    K, b = (500.0, 500.0, 320.0, 240.0), (0, 640, 0, 480)
    n_kf, n_feat, n_mp = 12, 400, 300
    base = rng.integers(0, 256, (n_mp, 32), dtype=np.uint8)
    kfs = []
    for _ in range(n_kf):
        x = rng.uniform(0, 640, n_feat).astype(np.float32)
        which = rng.permutation(n_mp)[:n_feat] if n_mp >= n_feat else rng.integers(0, n_mp, n_feat)
        d = flip_bits(rng, base[which % n_mp], 50)
        kf = KeyFrame(kp_struct(x, x * 0.5, np.zeros(n_feat, np.int32), np.zeros(n_feat, np.float32)), d, sf, s2, K, b)
        kf._which = which % n_mp
        kfs.append(kf)
    kfs[3].mbBad = True
    mps = [MapPoint(np.zeros(3), base[i], observations=0) for i in range(n_mp)]
    for kf in kfs:
        for idx, m in enumerate(kf._which):
            if rng.uniform() < 0.7:
                mps[m].AddObservation(kf, idx)
    mps[5]._bad = True
    before5 = mps[5].GetDescriptor().copy()
    compute_distinctive_descriptors(mps)
    checked = 0
    for m, mp in enumerate(mps):
        rows = [kf.mDescriptors[i] for kf, i in mp.mObservations.items() if not kf.isBad()]
        if m == 5 or not rows:
            continue
        wi, _ = oracle.distinctive_descriptor(np.stack(rows))
        assert np.array_equal(mp.GetDescriptor(), rows[wi])
        checked += 1
    assert checked > 100, "scene too poor"
    assert np.array_equal(mps[5].GetDescriptor(), before5)   # bad points are left alone
    one = mps[0]
    one.ComputeDistinctiveDescriptors()
