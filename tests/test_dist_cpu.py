"""world_size-2 (gloo, CPU) test of the sharded kNN-2 path: shard bounds, the all-gather
layout and the merge rule.  The per-shard scan and the merge are injected: here the CPU
oracle and a numpy restatement stand in for the HIP kernels (the kernels themselves are
checked by the GPU tests)."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def numpy_merge(ratio):
    def run(g):
        g = g.numpy()
        _, world, nq = g.shape
        bd = np.full(nq, 256, np.int32); sd = np.full(nq, 256, np.int32); bi = np.full(nq, -1, np.int32)
        for s in range(world):
            cb, cs, ci = g[0, s], g[1, s], g[2, s]
            valid = ci >= 0
            better = valid & (cb < bd)
            sd = np.where(better, np.minimum(bd, cs), np.where(valid, np.minimum(sd, cb), sd))
            bi = np.where(better, ci, bi)
            bd = np.where(better, cb, bd)
        acc = (bi >= 0) & (bd.astype(np.float32) < np.float32(ratio) * sd.astype(np.float32))
        return torch.from_numpy(np.stack([bd, sd, bi])), torch.from_numpy(acc.astype(np.uint8))
    return run


def _worker(rank, world, port, ndb, nq, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    from orb_slam2_amd.dist import knn2_sharded, shard_bounds
    from orb_slam2_amd.synth import synth_descriptors
    db, q = synth_descriptors(ndb, nq, seed=21)
    db[ndb // 2 + 3] = db[7]  # a duplicate row in the second shard must lose to the first
    lo, hi = shard_bounds(ndb, world, rank)

    def local():
        b, s, i, _ = O.knn2_bruteforce(q, db[lo:hi], 0.7)
        i = np.where(i >= 0, i + lo, -1)
        return torch.from_numpy(np.stack([b, s, i]).astype(np.int32))
    res, acc = knn2_sharded(local, numpy_merge(0.7), nq, torch.device("cpu"), world)
    if rank == 0:
        rb, rs, ri, ra = O.knn2_bruteforce(q, db, 0.7)
        ok = (np.array_equal(res[0].numpy(), rb) and np.array_equal(res[1].numpy(), rs) and
              np.array_equal(res[2].numpy(), ri) and np.array_equal(acc.numpy(), ra))
        out.put(bool(ok))
    dist.destroy_process_group()


def test_shard_bounds_cover_everything():
    from orb_slam2_amd.dist import shard_bounds
    for n in (0, 1, 7, 1000003):
        for w in (1, 2, 3, 8):
            b = [shard_bounds(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[r][1] == b[r + 1][0] for r in range(w - 1))
            assert max(h - l for l, h in b) - min(h - l for l, h in b) <= 1


def test_sharded_knn2_two_ranks_gloo():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 601, 150, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert out.get(timeout=5) is True
