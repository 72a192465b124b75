#!/usr/bin/env python3
"""Randomised parity rounds for the entry points around the extractor and the matchers:
vocabulary transform, distinctive descriptors, keypoint undistortion, colour input.
    python tools/fuzz_misc.py [rounds]       (needs a GPU; exits 1 on the first mismatch)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle
from orb_slam2_amd import _lib, ORBextractor
from orb_slam2_amd.bow import ORBVocabulary
from orb_slam2_amd.synth import synth_image


def fail(what, rnd, info):
    print("MISMATCH in %s, round %d: %s" % (what, rnd, info), flush=True)
    sys.exit(1)


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    t0 = time.time()
    L = _lib.lib()
    for rnd in range(rounds):
        rng = np.random.default_rng(1000 + rnd)
        # --- vocabulary transform
        k, depth = int(rng.integers(2, 14)), int(rng.integers(1, 5))
        voc = ORBVocabulary.synthetic(k=k, L=depth, seed=rnd, p_tie=float(rng.uniform(0, 0.5)),
                                      p_early_leaf=float(rng.uniform(0, 0.3)), p_stop=float(rng.uniform(0, 0.2)))
        n = int(rng.integers(1, 4000))
        src = voc.node_desc[rng.integers(1, len(voc.parent), n)]
        feats = np.ascontiguousarray(src ^ np.packbits(rng.random((n, 256)) < rng.uniform(0, 0.2), axis=1, bitorder="little"))
        lu = int(rng.integers(0, depth + 2))
        w, nid = voc.transform_ids(feats, lu)
        ow, on, wt = oracle.bow_transform(voc.parent, voc.is_leaf, voc.node_desc, voc.node_weight, voc.m_L, feats, lu)
        if not (np.array_equal(w, ow) and np.array_equal(nid, on)):
            fail("bow_transform", rnd, (k, depth, n, lu))
        bow, fv = voc.assemble(w, nid)
        ids, vals, keep = oracle.bow_vector_l1(ow, wt)
        if list(bow.keys()) != ids.tolist() or not np.array_equal(np.array(list(bow.values())), vals):
            fail("bow vector", rnd, (k, depth, n))
        # --- distinctive descriptors
        sizes = rng.integers(0, 40, int(rng.integers(1, 300))).tolist() + [int(rng.integers(60, 200))]
        start = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
        base = rng.integers(0, 256, (len(sizes), 32), dtype=np.uint8)
        desc = np.repeat(base, sizes, axis=0)
        desc ^= np.packbits(rng.random((len(desc), 256)) < rng.uniform(0, 0.3), axis=1, bitorder="little")
        desc = np.ascontiguousarray(desc)
        best, med = np.empty(len(sizes), np.int32), np.empty(len(sizes), np.int32)
        _lib.check(L.orb_distinctive_descriptors(0, len(sizes), _lib.ptr(start), _lib.ptr(desc), _lib.ptr(best), _lib.ptr(med)))
        for p in range(len(sizes)):
            if (best[p], med[p]) != oracle.distinctive_descriptor(desc[start[p]:start[p + 1]]):
                fail("distinctive descriptor", rnd, (p, sizes[p]))
        # --- undistortion
        K = (float(rng.uniform(300, 900)), float(rng.uniform(300, 900)), float(rng.uniform(200, 700)), float(rng.uniform(100, 400)))
        d = np.array([rng.uniform(-0.5, 0.5), rng.uniform(-1, 1), rng.uniform(-0.01, 0.01), rng.uniform(-0.01, 0.01),
                      rng.uniform(-1, 1) if rnd % 2 else 0.0], np.float32)
        m = int(rng.integers(1, 5000))
        xy = rng.uniform(-50, 1400, (m, 2)).astype(np.float32)
        out = np.empty_like(xy)
        _lib.check(L.orb_undistort_points(0, m, _lib.ptr(xy), _lib.Camera(*K, *d.tolist()), _lib.ptr(out)))
        want = oracle.undistort_points(xy, K, d)
        if not np.array_equal(out.view(np.uint32), want.view(np.uint32)):
            fail("undistort", rnd, (K, d.tolist()))
        # --- colour input
        rows, cols = int(rng.integers(120, 500)), int(rng.integers(160, 900))
        ch, rgb = int(rng.choice([3, 4])), bool(rng.integers(0, 2))
        g = synth_image(rows, cols, rnd).astype(np.int32)
        img = np.stack([np.clip(g + rng.integers(-60, 61, g.shape), 0, 255) for _ in range(ch)], 2).astype(np.uint8)
        gray = oracle.cvt_gray(img, rgb)
        nlv = 8 if min(rows, cols) >= 240 else 3
        ex = ORBextractor(int(rng.integers(100, 1500)), 1.2, nlv, 20, 7)
        ex.mbRGB = rgb
        try:
            kps, dsc = ex(np.ascontiguousarray(img))
        except _lib.OrbHipError as e:  # shapes on which the reference itself is undefined
            if e.code != _lib.ERR_UNSUPPORTED:
                raise
            kps = None
        if kps is not None:
            if not np.array_equal(ex.pyramid_level(0), gray):
                fail("colour level 0", rnd, (rows, cols, ch, rgb))
            k2, d2 = ex(gray)
            if not (np.array_equal(kps, k2) and np.array_equal(dsc, d2)):
                fail("colour vs grey extraction", rnd, (rows, cols, ch, rgb))
        ex.close()
        print("round %d ok (%d s)" % (rnd, time.time() - t0), flush=True)
    print("done: %d rounds, 0 mismatches" % rounds, flush=True)


if __name__ == "__main__":
    main()
