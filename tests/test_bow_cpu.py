"""Vocabulary-tree transform: the oracle against an independent pure-Python descent, the text
format round trip, and the host assembly of BowVector / FeatureVector (no GPU)."""
import numpy as np

from oracle import oracle
from orb_slam2_amd.bow import ORBVocabulary


def _py_transform(voc, feats, levelsup):
    """Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1232-1273, one feature at a time."""
    kids = {}
    for i in range(1, len(voc.parent)):
        kids.setdefault(int(voc.parent[i]), []).append(i)
    words, nodes = [], []
    for f in feats:
        node, level, nid = 0, 0, 0
        while True:
            level += 1
            best, pick = None, None
            for c in kids[node]:
                d = int(np.unpackbits(voc.node_desc[c] ^ f).sum())
                if best is None or d < best:
                    best, pick = d, c
            node = pick
            if level == voc.m_L - levelsup:
                nid = node
            if node not in kids:
                break
        words.append(int(voc.word_id[node]))
        nodes.append(nid)
    return np.array(words, np.int32), np.array(nodes, np.int32)


def _features(voc, n, seed):
    """Noisy copies of node descriptors, so that descents spread over the tree and meet ties."""
    rng = np.random.default_rng(seed)
    src = voc.node_desc[rng.integers(1, len(voc.parent), n)]
    noise = np.packbits(rng.random((n, 256)) < 0.08, axis=1, bitorder="little")
    return np.ascontiguousarray(src ^ noise)


def test_oracle_descent_matches_python():
    voc = ORBVocabulary.synthetic(k=4, L=3, seed=3)
    feats = _features(voc, 300, 1)
    for levelsup in (0, 1, 2, 3, 5):
        w, nid, wt = oracle.bow_transform(voc.parent, voc.is_leaf, voc.node_desc, voc.node_weight, voc.m_L, feats,
                                          levelsup)
        pw, pn = _py_transform(voc, feats, levelsup)
        assert np.array_equal(w, pw) and np.array_equal(nid, pn)
        leaf_nodes = np.flatnonzero(voc.is_leaf)
        assert np.array_equal(wt, voc.node_weight[leaf_nodes[w]])


def test_ties_take_first_child():
    # two identical children: the first one wins
    parent = [-1, 0, 0, 0]
    desc = np.zeros((4, 32), np.uint8)
    desc[3] = 255
    voc = ORBVocabulary().set_nodes(3, 1, parent, [0, 1, 1, 1], desc, [0, 1.0, 2.0, 3.0])
    w, nid, _ = oracle.bow_transform(voc.parent, voc.is_leaf, voc.node_desc, voc.node_weight, 1,
                                     np.zeros((1, 32), np.uint8), 0)
    assert w[0] == 0 and nid[0] == 1


def test_text_round_trip(tmp_path):
    voc = ORBVocabulary.synthetic(k=5, L=3, seed=9)
    path = str(tmp_path / "voc.txt")
    voc.saveToTextFile(path)
    back = ORBVocabulary()
    assert back.loadFromTextFile(path)
    assert (back.m_k, back.m_L, back.m_scoring, back.m_weighting) == (5, 3, 0, 0)
    for name in ("parent", "is_leaf", "node_desc", "node_weight", "word_id"):
        assert np.array_equal(getattr(back, name), getattr(voc, name)), name
    assert back.size() == voc.size() and not back.empty()
    bad = tmp_path / "bad.txt"
    bad.write_text("30 3 0 0\n")
    assert not ORBVocabulary().loadFromTextFile(str(bad))


def test_binary_round_trip(tmp_path):
    voc = ORBVocabulary.synthetic(k=5, L=3, seed=10)
    path = str(tmp_path / "voc.bin")
    voc.saveToBinaryFile(path)
    raw = open(path, "rb").read()
    assert len(raw) == 24 + 41 * (len(voc.parent) - 1)
    back = ORBVocabulary()
    assert back.loadFromBinaryFile(path)
    assert (back.m_k, back.m_L) == (5, 3)
    for name in ("parent", "is_leaf", "node_desc", "word_id"):
        assert np.array_equal(getattr(back, name), getattr(voc, name)), name
    assert np.array_equal(back.node_weight, voc.node_weight.astype(np.float32).astype(np.float64))
    assert not ORBVocabulary().loadFromBinaryFile(str(tmp_path / "missing.bin"))


def test_children_csr_keeps_file_order():
    voc = ORBVocabulary.synthetic(k=4, L=3, seed=5)
    start, count, children = voc.children_csr()
    assert count.sum() == len(voc.parent) - 1
    for i in range(len(voc.parent)):
        c = children[start[i]:start[i] + count[i]]
        assert np.all(voc.parent[c] == i) and np.all(np.diff(c) > 0)
        assert (count[i] == 0) == bool(voc.is_leaf[i]) or i == 0


def test_assembly_matches_sequential_maps():
    voc = ORBVocabulary.synthetic(k=6, L=3, seed=11, p_stop=0.1)
    feats = _features(voc, 2000, 2)
    w, nid, wt = oracle.bow_transform(voc.parent, voc.is_leaf, voc.node_desc, voc.node_weight, voc.m_L, feats, 2)
    ids, vals, keep = oracle.bow_vector_l1(w, wt)
    assert not keep.all() and keep.any()
    bow, fv = voc.assemble(w, nid)
    assert list(bow.keys()) == ids.tolist()
    assert np.array_equal(np.array(list(bow.values())), vals)  # bit-exact doubles
    assert abs(sum(bow.values()) - 1.0) < 1e-12
    # FeatureVector: ascending nodes, ascending feature indices, only features that were kept
    ref = {}
    for i in np.flatnonzero(keep):
        ref.setdefault(int(nid[i]), []).append(int(i))
    assert fv == dict(sorted(ref.items())) and list(fv.keys()) == sorted(fv.keys())
    # L1 score of a vector with itself is 1, with a disjoint one 0
    assert abs(voc.score(bow, bow) - 1.0) < 1e-12
    assert voc.score(bow, {max(bow) + 1: 1.0}) == 0.0
