#!/usr/bin/env python3
"""Randomised parity sweep of the brute-force Hamming kNN-2 (run on a GPU box): both kernels
(popcount on the vector units, int8 matrix cores) at random sizes - ragged against the kernels'
tiles of 128 queries, 32 and 512 database rows - with planted duplicates and near matches, against
each other and against the oracle; every few rounds the database in ascending shards, merged.
    python tools/fuzz_knn.py [n_rounds] [seed]"""
import ctypes as C
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from oracle import oracle as O
from orb_slam2_amd import _lib
from helpers import flip_bits

O.lib()
L = _lib.lib()
n_rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
t0 = time.time()


def scan(d_q, nq, d_db, lo, hi, base, ratio, algo):
    res = torch.full((3, nq), -7, dtype=torch.int32, device="cuda")
    acc = torch.zeros(nq, dtype=torch.uint8, device="cuda")
    _lib.check(L.orb_knn2_device_algo(0, C.c_void_p(d_q.data_ptr()), nq, C.c_void_p(d_db.data_ptr() + 32 * lo), hi - lo,
                                      base + lo, ratio, C.c_void_p(res[0].data_ptr()), C.c_void_p(res[1].data_ptr()),
                                      C.c_void_p(res[2].data_ptr()), C.c_void_p(acc.data_ptr()), None, algo))
    torch.cuda.synchronize()
    return res, acc


for r in range(n_rounds):
    nq = int(rng.choice([1, 2, 63, 64, 65, 127, 128, 129])) if rng.integers(0, 2) else int(rng.integers(1, 4000))
    ndb = int(rng.choice([1, 2, 31, 32, 33, 511, 512, 513])) if rng.integers(0, 3) == 0 else int(rng.integers(1, 30000))
    ratio = float(rng.choice([0.6, 0.7, 0.9, 1.0]))
    base = int(rng.choice([0, 1000, 1 << 20]))
    db = rng.integers(0, 256, (ndb, 32), dtype=np.uint8)
    if ndb > 8:
        db[ndb // 2] = db[1]; db[ndb - 1] = db[0]        # exact duplicates: the lower row wins
    q = db[rng.integers(0, ndb, nq)].copy()
    near = rng.integers(0, 2, nq).astype(bool)
    if near.any():
        q[near] = flip_bits(rng, q[near], int(rng.integers(1, 80)))
    q[::7] = rng.integers(0, 256, (len(q[::7]), 32), dtype=np.uint8)
    d_q, d_db = torch.from_numpy(q).cuda(), torch.from_numpy(db).cuda()
    wb, ws, wi, wa = O.knn2_bruteforce(q, db, ratio)
    what = []
    for algo in (1, 2, 0):
        res, acc = scan(d_q, nq, d_db, 0, ndb, base, ratio, algo)
        res, acc = res.cpu().numpy(), acc.cpu().numpy()
        ok = np.array_equal(res[0], wb) and np.array_equal(res[1], ws) and np.array_equal(acc, wa)
        # (a query with no neighbour at all keeps index -1 whatever the base)
        ok = ok and np.array_equal(res[2], np.where(wi >= 0, wi + base, wi))
        if not ok:
            what.append("kernel %d" % algo)
    if r % 3 == 0 and ndb >= 4:
        ns = int(rng.integers(2, min(9, ndb + 1)))
        cuts = np.unique(np.concatenate([[0, ndb], rng.integers(1, ndb, ns - 1)]))
        parts = [scan(d_q, nq, d_db, int(a), int(b), base, ratio, int(rng.choice([1, 2]))) for a, b in zip(cuts[:-1], cuts[1:])]
        sb = torch.stack([p[0][0] for p in parts]).contiguous(); ss = torch.stack([p[0][1] for p in parts]).contiguous()
        si = torch.stack([p[0][2] for p in parts]).contiguous()
        out = torch.zeros((3, nq), dtype=torch.int32, device="cuda"); oa = torch.zeros(nq, dtype=torch.uint8, device="cuda")
        _lib.check(L.orb_knn2_merge_device(0, len(parts), nq, C.c_void_p(sb.data_ptr()), C.c_void_p(ss.data_ptr()),
                                           C.c_void_p(si.data_ptr()), ratio, C.c_void_p(out[0].data_ptr()),
                                           C.c_void_p(out[1].data_ptr()), C.c_void_p(out[2].data_ptr()),
                                           C.c_void_p(oa.data_ptr()), None))
        torch.cuda.synchronize()
        o, a = out.cpu().numpy(), oa.cpu().numpy()
        if not (np.array_equal(o[0], wb) and np.array_equal(o[1], ws) and np.array_equal(a, wa)
                and np.array_equal(o[2], np.where(wi >= 0, wi + base, wi))):
            what.append("%d shards merged" % len(parts))
        # the same shards as 8-byte records (the form ranks exchange), merged
        packed = torch.zeros((len(parts), nq, 2), dtype=torch.int32, device="cuda")
        for k, p in enumerate(parts):
            _lib.check(L.orb_knn2_pack_device(0, nq, C.c_void_p(p[0][0].data_ptr()), C.c_void_p(p[0][1].data_ptr()),
                                              C.c_void_p(p[0][2].data_ptr()), C.c_void_p(packed[k].data_ptr()), None))
        out.zero_(); oa.zero_()
        _lib.check(L.orb_knn2_merge_packed_device(0, len(parts), nq, C.c_void_p(packed.data_ptr()), ratio,
                                                  C.c_void_p(out[0].data_ptr()), C.c_void_p(out[1].data_ptr()),
                                                  C.c_void_p(out[2].data_ptr()), C.c_void_p(oa.data_ptr()), None))
        torch.cuda.synchronize()
        o, a = out.cpu().numpy(), oa.cpu().numpy()
        if not (np.array_equal(o[0], wb) and np.array_equal(o[1], ws) and np.array_equal(a, wa)
                and np.array_equal(o[2], np.where(wi >= 0, wi + base, wi))):
            what.append("%d shards as packed records merged" % len(parts))
    if what:
        bad += 1
        print("MISMATCH round %d: %d queries x %d rows, ratio %.1f, base %d: %s" % (r, nq, ndb, ratio, base, "; ".join(what)), flush=True)
    elif r % 20 == 0:
        print("round %d ok (%d x %d) %.0fs" % (r, nq, ndb, time.time() - t0), flush=True)
print("done: %d rounds, %d mismatches" % (n_rounds, bad))
sys.exit(1 if bad else 0)
