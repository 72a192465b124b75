"""Vocabulary-tree transform on the GPU (through the C ABI) against the oracle: word ids and
node ids bit-exact, BowVector values bit-exact doubles."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle
from orb_slam2_amd import _lib
from orb_slam2_amd.bow import ORBVocabulary
from test_bow_cpu import _features

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def voc():
    return ORBVocabulary.synthetic(k=10, L=4, seed=21)


def _oracle_ids(voc, feats, levelsup):
    return oracle.bow_transform(voc.parent, voc.is_leaf, voc.node_desc, voc.node_weight, voc.m_L, feats, levelsup)


@pytest.mark.parametrize("levelsup", [0, 1, 2, 4, 7])
def test_ids_match_oracle(voc, levelsup):
    feats = _features(voc, 5000, 100 + levelsup)
    w, nid = voc.transform_ids(feats, levelsup)
    ow, on, _ = _oracle_ids(voc, feats, levelsup)
    assert np.array_equal(w, ow)
    assert np.array_equal(nid, on)
    assert len(np.unique(w)) > 500, "scene too poor: descents do not spread over the tree"


def test_transform_matches_oracle_maps(voc):
    feats = _features(voc, 2000, 7)
    bow, fv = voc.transform(feats, 2)
    ow, on, wt = _oracle_ids(voc, feats, 2)
    ids, vals, keep = oracle.bow_vector_l1(ow, wt)
    assert list(bow.keys()) == ids.tolist()
    assert np.array_equal(np.array(list(bow.values())), vals)
    ref = {}
    for i in np.flatnonzero(keep):
        ref.setdefault(int(on[i]), []).append(int(i))
    assert fv == dict(sorted(ref.items()))


def test_sizes_and_odd_trees():
    # uneven branching: a binary tree with leaves at every depth, and a single-level tree
    v1 = ORBVocabulary.synthetic(k=2, L=9, seed=4, p_early_leaf=0.2)
    v2 = ORBVocabulary.synthetic(k=13, L=1, seed=5)
    for v in (v1, v2):
        for n in (1, 63, 64, 65, 257, 3001):
            feats = _features(v, n, n)
            w, nid = v.transform_ids(feats, 1)
            ow, on, _ = _oracle_ids(v, feats, 1)
            assert np.array_equal(w, ow) and np.array_equal(nid, on)
    assert v1.transform(np.zeros((0, 32), np.uint8)) == ({}, {})


def test_exact_ties_between_siblings():
    voc = ORBVocabulary.synthetic(k=10, L=3, seed=8, p_tie=0.6)
    feats = voc.node_desc[1:2001].copy()  # exact node descriptors: distance-0 ties with duplicates
    w, nid = voc.transform_ids(feats, 1)
    ow, on, _ = _oracle_ids(voc, feats, 1)
    assert np.array_equal(w, ow) and np.array_equal(nid, on)


def test_device_entry_point(voc):
    torch = pytest.importorskip("torch")
    feats = _features(voc, 4096, 33)
    d = torch.from_numpy(feats).cuda()
    out = torch.empty((2, len(feats)), dtype=torch.int32, device="cuda")
    s = torch.cuda.current_stream()
    L = _lib.lib()
    _lib.check(L.orb_bow_transform_device(voc._device(), len(feats), C.c_void_p(d.data_ptr()), 2,
                                          C.c_void_p(out[0].data_ptr()), C.c_void_p(out[1].data_ptr()),
                                          C.c_void_p(s.cuda_stream)))
    s.synchronize()
    ow, on, _ = _oracle_ids(voc, feats, 2)
    assert np.array_equal(out[0].cpu().numpy(), ow) and np.array_equal(out[1].cpu().numpy(), on)


def test_rejects_bad_trees():
    L = _lib.lib()
    h = C.c_void_p()
    start = np.array([0, 2, 2], np.int32)
    count = np.array([2, 0, 0], np.int32)
    desc = np.zeros((3, 32), np.uint8)
    word = np.array([-1, 0, 1], np.int32)
    bad_children = np.array([1, 0], np.int32)  # a child that is not above its parent
    rc = L.orb_vocabulary_create(0, 3, 1, _lib.ptr(start), _lib.ptr(count), _lib.ptr(bad_children), _lib.ptr(desc),
                                 _lib.ptr(word), C.byref(h))
    assert rc == _lib.ERR_ARG and not h.value
    children = np.array([1, 2], np.int32)
    no_word = np.array([-1, 0, -1], np.int32)
    rc = L.orb_vocabulary_create(0, 3, 1, _lib.ptr(start), _lib.ptr(count), _lib.ptr(children), _lib.ptr(desc),
                                 _lib.ptr(no_word), C.byref(h))
    assert rc == _lib.ERR_ARG
    assert L.orb_bow_transform(None, 4, _lib.ptr(desc), 0, _lib.ptr(word), _lib.ptr(word)) == _lib.ERR_ARG


def test_extract_computebow_searchbybow_chain():
    """Two frames of a sequence through the extractor, Frame::ComputeBoW and
    SearchByBoW(KeyFrame, Frame); the oracle repeats the last two steps on the same descriptors."""
    from orb_slam2_amd import Frame, KeyFrame, MapPoint, ORBextractor, ORBmatcher
    from orb_slam2_amd.synth import synth_sequence
    from helpers import scale_tables
    imgs = synth_sequence(376, 1241, 2, seed=3)
    ex = ORBextractor(2000, 1.2, 8, 20, 7)
    (k1, d1), (k2, d2) = ex(imgs[0]), ex(imgs[1])
    assert len(k1) > 1000 and len(k2) > 1000, "scene too poor"
    voc = ORBVocabulary.synthetic_full(10, 5, seed=2)
    # make the top of the tree look like the data, so that the ten level-1 buckets all fill up
    voc.node_desc[1:11] = d1[:: len(d1) // 10][:10]
    sf, s2 = scale_tables()
    K, b = (500.0, 500.0, 620.0, 188.0), (0, 1241, 0, 376)
    kf, fr = KeyFrame(k1, d1, sf, s2, K, b), Frame(k2, d2, sf, s2, K, b)
    kf.ComputeBoW(voc)
    fr.ComputeBoW(voc)
    for f, d in ((kf, d1), (fr, d2)):
        ow, on, wt = _oracle_ids(voc, d, 4)
        ids, vals, keep = oracle.bow_vector_l1(ow, wt)
        assert list(f.mBowVec.keys()) == ids.tolist() and np.array_equal(np.array(list(f.mBowVec.values())), vals)
        ref = {}
        for i in np.flatnonzero(keep):
            ref.setdefault(int(on[i]), []).append(int(i))
        assert f.mFeatVec == dict(sorted(ref.items()))
    assert len(kf.mFeatVec) >= 8, "scene too poor: features fall into too few nodes"
    assert 0.0 < voc.score(kf.mBowVec, fr.mBowVec) <= 1.0
    for i in range(len(k1)):
        kf.mvpMapPoints[i] = MapPoint(np.zeros(3), d1[i])
    out = []
    got = ORBmatcher(0.7, True).SearchByBoW(kf, fr, out)

    def flat(fv):
        start, items = [0], []
        for v in fv.values():
            items.extend(v)
            start.append(len(items))
        return dict(node=np.array(list(fv.keys()), np.int32), start=np.array(start, np.int32),
                    items=np.array(items, np.int32))
    wn, w12 = oracle.search_by_bow(dict(angle=k1["angle"], desc=d1, ok=np.ones(len(k1), np.uint8)),
                                   dict(angle=k2["angle"], desc=d2, ok=np.ones(len(k2), np.uint8)),
                                   flat(kf.mFeatVec), flat(fr.mFeatVec), 0.7, 50, True)
    assert got == wn
    assert wn > 100, "scene too poor"
    want = [None] * len(k2)
    for i1 in np.flatnonzero(w12 >= 0):
        want[w12[i1]] = kf.mvpMapPoints[i1]
    assert all(a is w for a, w in zip(out, want))
