"""Host mirror of the reference's ORBVocabulary (include/ORBVocabulary.h: a
DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB>) for the part the front end calls:
loading the text vocabulary and transform(features, BowVector, FeatureVector, levelsup)
(Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1141-1208, called from src/Frame.cc:483-497 and
src/KeyFrame.cc:120-131 with levelsup = 4).  The tree descent of every feature runs on the GPU
(liborbhip.so: orb_bow_transform); the two small ordered maps are put together here from the
per-feature (word, node) pairs in the reference's order of floating-point operations."""
import ctypes as C

import numpy as np

from . import _lib

# DBoW2 enums (Thirdparty/DBoW2/DBoW2/BowVector.h)
TF_IDF, TF, IDF, BINARY = 0, 1, 2, 3
L1_NORM, L2_NORM, CHI_SQUARE, KL, BHATTACHARYYA, DOT_PRODUCT = 0, 1, 2, 3, 4, 5


class ORBVocabulary:
    def __init__(self, device_id=0):
        self.device_id = int(device_id)
        self.m_k = 10
        self.m_L = 5
        self.m_scoring = L1_NORM
        self.m_weighting = TF_IDF
        self._clear()

    def _clear(self):
        self.parent = np.zeros(1, np.int32)  # node 0 = root
        self.is_leaf = np.zeros(1, np.uint8)
        self.node_desc = np.zeros((1, 32), np.uint8)
        self.node_weight = np.zeros(1, np.float64)
        self.word_id = np.full(1, -1, np.int32)
        self._handle = None

    def __del__(self):
        self._release()

    def _release(self):
        if getattr(self, "_handle", None):
            _lib.lib().orb_vocabulary_destroy(self._handle)
            self._handle = None

    # ---- construction -------------------------------------------------------------------
    def set_nodes(self, k, L, parent, is_leaf, node_desc, node_weight, scoring=L1_NORM, weighting=TF_IDF):
        """Records of nodes 1..n-1 in file order (entry 0 of each array is the root's and is
        ignored except for its shape)."""
        self._release()
        self.m_k, self.m_L, self.m_scoring, self.m_weighting = int(k), int(L), int(scoring), int(weighting)
        self.parent = np.ascontiguousarray(parent, np.int32)
        self.is_leaf = np.ascontiguousarray(is_leaf, np.uint8)
        self.node_desc = np.ascontiguousarray(node_desc, np.uint8).reshape(-1, 32)
        self.node_weight = np.ascontiguousarray(node_weight, np.float64)
        n = len(self.parent)
        if not (len(self.is_leaf) == len(self.node_desc) == len(self.node_weight) == n):
            raise ValueError("node arrays differ in length")
        if n > 1 and not np.all(self.parent[1:] < np.arange(1, n)) or np.any(self.parent[1:] < 0):
            raise ValueError("a node's parent must come before it")
        self.parent[0] = -1
        self.is_leaf[0] = 0
        # word ids number the leaves in file order (TemplatedVocabulary.h:1422-1429)
        self.word_id = np.where(self.is_leaf > 0, np.cumsum(self.is_leaf > 0) - 1, -1).astype(np.int32)
        return self

    def loadFromTextFile(self, filename):
        """TemplatedVocabulary.h:1352-1438: header 'k L scoring weighting', then one line per
        node: parent id, leaf flag, 32 descriptor bytes, weight.  Empty lines are skipped (the
        reference turns a trailing empty line into a node of indeterminate content)."""
        self._release()
        with open(filename, "r") as f:
            head = f.readline().split()
            try:
                k, L, n1, n2 = (int(v) for v in head[:4])
            except ValueError:
                return False
            if k < 0 or k > 20 or L < 1 or L > 10 or n1 < 0 or n1 > 5 or n2 < 0 or n2 > 3:
                return False
            rows = np.loadtxt(f, dtype=np.float64, ndmin=2)
        if rows.size and rows.shape[1] != 35:
            return False
        n = len(rows) + 1
        parent = np.zeros(n, np.int32)
        leaf = np.zeros(n, np.uint8)
        desc = np.zeros((n, 32), np.uint8)
        weight = np.zeros(n, np.float64)
        if len(rows):
            parent[1:] = rows[:, 0].astype(np.int32)
            leaf[1:] = rows[:, 1] > 0
            desc[1:] = rows[:, 2:34].astype(np.int64).astype(np.uint8)  # (unsigned char)n
            weight[1:] = rows[:, 34]
        self.set_nodes(k, L, parent, leaf, desc, weight, n1, n2)
        return True

    def saveToTextFile(self, filename):
        """TemplatedVocabulary.h:1521-1541 (the columns the loader reads; weights are written
        with all their digits, the reference writes six)."""
        with open(filename, "w") as f:
            f.write("%d %d %d %d\n" % (self.m_k, self.m_L, self.m_scoring, self.m_weighting))
            for i in range(1, len(self.parent)):
                f.write("%d %d %s %s\n" % (self.parent[i], 1 if self.is_leaf[i] else 0,
                                           " ".join(str(int(b)) for b in self.node_desc[i]),
                                           repr(float(self.node_weight[i]))))

    _BIN_NODE = np.dtype([("parent", "<i4"), ("desc", "u1", (32,)), ("weight", "<f4"), ("leaf", "u1")])

    def loadFromBinaryFile(self, filename):
        """TemplatedVocabulary.h:1444-1489: header nb_nodes (root included), size_node, k, L,
        scoring, weighting as 32-bit values, then one 41-byte record per node 1..nb_nodes-1:
        parent id, 32 descriptor bytes, float weight, leaf flag."""
        self._release()
        try:
            with open(filename, "rb") as f:
                head = np.fromfile(f, "<i4", 6)
                if len(head) < 6 or head[1] != self._BIN_NODE.itemsize:
                    return False
                rec = np.fromfile(f, self._BIN_NODE, max(int(head[0]) - 1, 0))
        except OSError:
            return False
        n = len(rec) + 1
        parent, leaf = np.zeros(n, np.int32), np.zeros(n, np.uint8)
        desc, weight = np.zeros((n, 32), np.uint8), np.zeros(n, np.float64)
        parent[1:], leaf[1:], desc[1:], weight[1:] = rec["parent"], rec["leaf"] != 0, rec["desc"], rec["weight"]
        self.set_nodes(head[2], head[3], parent, leaf, desc, weight, head[4], head[5])
        return True

    def saveToBinaryFile(self, filename):
        """TemplatedVocabulary.h:1495-1515 (weights narrow to float, as there)."""
        n = len(self.parent)
        rec = np.zeros(n - 1, self._BIN_NODE)
        rec["parent"], rec["desc"] = self.parent[1:], self.node_desc[1:]
        rec["weight"], rec["leaf"] = self.node_weight[1:].astype(np.float32), self.is_leaf[1:]
        with open(filename, "wb") as f:
            np.array([n, self._BIN_NODE.itemsize, self.m_k, self.m_L, self.m_scoring, self.m_weighting], "<i4").tofile(f)
            rec.tofile(f)

    @classmethod
    def synthetic(cls, k=10, L=4, seed=0, device_id=0, p_tie=0.1, p_early_leaf=0.05, p_stop=0.02):
        """Random k-ary tree of depth L in the id order hierarchical k-means creates nodes
        (a node's children get consecutive ids, then each child's subtree follows).  Children
        are bit-flipped copies of their parent; some siblings repeat a descriptor (ties), some
        branches end early, some words have weight 0 (stopped)."""
        rng = np.random.default_rng(seed)
        parent, leaf, desc, weight = [-1], [0], [np.zeros(32, np.uint8)], [0.0]

        def grow(node, level):
            kids = []
            base = desc[node] if node else rng.integers(0, 256, 32, dtype=np.uint8)
            for c in range(k):
                if c and rng.random() < p_tie:
                    d = desc[kids[int(rng.integers(0, c))]].copy()
                else:
                    flips = rng.random(256) < 0.5 / (level + 1)
                    d = base ^ np.packbits(flips, bitorder="little")
                is_leaf = level == L or rng.random() < p_early_leaf
                parent.append(node)
                leaf.append(1 if is_leaf else 0)
                desc.append(d)
                weight.append(0.0 if (is_leaf and rng.random() < p_stop) else float(rng.random() * 8 + 0.01))
                kids.append(len(parent) - 1)
            for c in kids:
                if not leaf[c]:
                    grow(c, level + 1)

        grow(0, 1)
        v = cls(device_id)
        return v.set_nodes(k, L, parent, leaf, np.stack(desc), weight)

    @classmethod
    def synthetic_full(cls, k=10, L=6, seed=0, device_id=0):
        """Complete k-ary tree of depth L built level by level (the size of the ORB vocabulary
        at k = 10, L = 6: 1111111 nodes, 10^6 words); for throughput measurements."""
        rng = np.random.default_rng(seed)
        parent, desc = [np.array([-1], np.int32)], [np.zeros((1, 32), np.uint8)]
        first, count = 0, 1
        for level in range(1, L + 1):
            ids = np.repeat(np.arange(first, first + count, dtype=np.int32), k)
            base = desc[-1][ids - first] if level > 1 else rng.integers(0, 256, (k, 32), dtype=np.uint8)
            flips = rng.integers(0, 256, (len(ids), 32), dtype=np.uint8) & rng.integers(0, 256, (len(ids), 32), dtype=np.uint8)
            for _ in range(level - 1):  # fewer flipped bits further down
                flips &= rng.integers(0, 256, flips.shape, dtype=np.uint8)
            parent.append(ids)
            desc.append(base ^ flips)
            first, count = first + count, len(ids)
        parent = np.concatenate(parent)
        n = len(parent)
        leaf = np.zeros(n, np.uint8)
        leaf[first:] = 1
        return cls(device_id).set_nodes(k, L, parent, leaf, np.concatenate(desc), rng.random(n) * 8 + 0.01)

    # ---- queries ------------------------------------------------------------------------
    def empty(self):
        return int(np.count_nonzero(self.is_leaf)) == 0

    def size(self):
        return int(np.count_nonzero(self.is_leaf))

    def getBranchingFactor(self):
        return self.m_k

    def getDepthLevels(self):
        return self.m_L

    def children_csr(self):
        """(child_start, child_count, children) with each node's children in file order."""
        n = len(self.parent)
        count = np.bincount(self.parent[1:], minlength=n).astype(np.int32)
        start = np.zeros(n, np.int32)
        start[1:] = np.cumsum(count)[:-1]
        children = (np.argsort(self.parent[1:], kind="stable") + 1).astype(np.int32)
        return start, count, children

    def _device(self):
        if self._handle is None:
            L = _lib.lib()
            start, count, children = self.children_csr()
            h = C.c_void_p()
            _lib.check(L.orb_vocabulary_create(self.device_id, len(self.parent), self.m_L, _lib.ptr(start),
                                               _lib.ptr(count), _lib.ptr(children), _lib.ptr(self.node_desc),
                                               _lib.ptr(self.word_id), C.byref(h)))
            self._handle = h
        return self._handle

    def transform_ids(self, features, levelsup=4):
        """Per feature: (word id, node id at depth L - levelsup), computed on the GPU."""
        feats = np.ascontiguousarray(features, np.uint8).reshape(-1, 32)
        n = len(feats)
        word, node = np.empty(max(n, 1), np.int32), np.empty(max(n, 1), np.int32)
        if n and not self.empty():
            _lib.check(_lib.lib().orb_bow_transform(self._device(), n, _lib.ptr(feats), int(levelsup),
                                                    _lib.ptr(word), _lib.ptr(node)))
        return word[:n], node[:n]

    def assemble(self, word, node):
        """BowVector and FeatureVector from per-feature ids (TemplatedVocabulary.h:1159-1207,
        BowVector.cpp:34-84, FeatureVector.cpp:31-46).  Both come back as dicts in ascending key
        order; per word the weights are added in feature order and the norm is accumulated in
        ascending word order, as the std::map loops of the reference do."""
        word = np.asarray(word, np.int32)
        node = np.asarray(node, np.int32)
        leaf_nodes = np.flatnonzero(self.is_leaf)  # word id -> node id
        w = self.node_weight[leaf_nodes[word]] if len(word) else np.zeros(0)
        if self.m_weighting in (TF, BINARY):
            w = np.where(w > 0, 1.0, 0.0)  # the stored weights are 1 for these (:1167, :1195)
        keep = w > 0
        idx = np.flatnonzero(keep)
        uniq, first, inv = np.unique(word[idx], return_index=True, return_inverse=True)
        vals = np.zeros(len(uniq), np.float64)
        if self.m_weighting in (TF, TF_IDF):
            np.add.at(vals, inv, w[idx])  # unbuffered: in feature order, like addWeight
        else:
            vals = w[idx][first]  # addIfNotExist keeps the first
        must, l2 = self.m_scoring != DOT_PRODUCT, self.m_scoring == L2_NORM
        if len(vals) and not must and self.m_weighting in (TF, TF_IDF):
            vals = vals / float(len(vals))
        if must and len(vals):
            seq = np.cumsum(vals * vals if l2 else np.abs(vals))  # sequential, ascending word id
            norm = np.sqrt(seq[-1]) if l2 else seq[-1]
            if norm > 0.0:
                vals = vals / norm
        bow = {int(k_): float(v_) for k_, v_ in zip(uniq, vals)}
        order = idx[np.argsort(node[idx], kind="stable")]
        fv = {}
        for nid, i in zip(node[order].tolist(), order.tolist()):
            fv.setdefault(nid, []).append(i)
        return bow, fv

    def transform(self, features, levelsup=4):
        """-> (BowVector, FeatureVector) of a frame's descriptors."""
        if self.empty():
            return {}, {}
        word, node = self.transform_ids(features, levelsup)
        return self.assemble(word, node)

    def score(self, v1, v2):
        """L1Scoring::score (Thirdparty/DBoW2/DBoW2/ScoringObject.cpp:23-68)."""
        if self.m_scoring != L1_NORM:
            raise NotImplementedError("only the L1 scoring of the ORB vocabulary is mirrored")
        s = 0.0
        for k_ in sorted(set(v1) & set(v2)):
            vi, wi = v1[k_], v2[k_]
            s += abs(vi - wi) - abs(vi) - abs(wi)
        return -s / 2.0
