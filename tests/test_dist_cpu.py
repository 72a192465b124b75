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


def _worker_codecs(rank, world, port, ndb, nq, out):
    """Both exchange formats through the same host logic: the 8-byte records must merge to exactly
    what the three int32 values merge to, and both to the single scan."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    from orb_slam2_amd.dist import knn2_sharded, shard_bounds
    from orb_slam2_amd.synth import synth_descriptors
    db, q = synth_descriptors(ndb, nq, seed=5)
    db[ndb - 2] = db[1]
    q[3] = db[1]                       # distance 0, present in both shards: the first shard's row wins
    q[4] = 255 - db[ndb - 1]           # distance 256 to one row: the largest value a record carries
    lo, hi = shard_bounds(ndb, world, rank)

    def local():
        b, s, i, _ = O.knn2_bruteforce(q, db[lo:hi], 0.7)
        i = np.where(i >= 0, i + lo, -1)
        return torch.from_numpy(np.stack([b, s, i]).astype(np.int32))
    t8 = {}
    r8, a8 = knn2_sharded(local, numpy_merge(0.7), nq, torch.device("cpu"), world, codec="packed8", timings=t8)
    r12, a12 = knn2_sharded(local, numpy_merge(0.7), nq, torch.device("cpu"), world, codec="int32x3")
    if rank == 0:
        rb, rs, ri, ra = O.knn2_bruteforce(q, db, 0.7)
        ok = torch.equal(r8, r12) and torch.equal(a8, a12)
        ok = ok and np.array_equal(r8[0].numpy(), rb) and np.array_equal(r8[1].numpy(), rs)
        ok = ok and np.array_equal(r8[2].numpy(), ri) and np.array_equal(a8.numpy(), ra)
        ok = ok and set(t8) == {"scan_s", "exchange_s"}
        out.put(bool(ok))
    dist.destroy_process_group()


def test_packed_records_round_trip():
    from orb_slam2_amd.dist import pack_records, unpack_records
    rng = np.random.default_rng(0)
    n = 1000
    best = rng.integers(0, 257, n).astype(np.int32)
    second = np.maximum(best, rng.integers(0, 257, n)).astype(np.int32)
    idx = rng.integers(-1, 2**31 - 1, n).astype(np.int32)
    best[:3], second[:3], idx[:3] = [0, 256, 256], [0, 256, 256], [0, -1, 2**31 - 1]
    res = torch.from_numpy(np.stack([best, second, idx]))
    rec = pack_records(res)
    assert rec.shape == (n, 2) and rec.dtype == torch.int32 and rec.element_size() * rec.shape[1] == 8
    back = unpack_records(rec.unsqueeze(0))
    assert torch.equal(back[:, 0, :], res)


def test_sharded_knn2_packed_equals_unpacked_two_ranks_gloo():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_worker_codecs, args=(r, 2, port, 500, 120, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert out.get(timeout=5) is True


def test_bench_spawns_its_ranks_and_relays_rank_zero():
    """bench.py --gpus N without a launcher: N child processes with the rank environment, rank 0's
    stdout handed on, a failing rank makes the whole run fail and the others are ended."""
    sys.path.insert(0, ROOT)
    import bench
    prog = ("import os, sys; r = int(os.environ['RANK']); "
            "assert os.environ['LOCAL_RANK'] == str(r) and os.environ['MASTER_ADDR'] == '127.0.0.1'; "
            "int(os.environ['MASTER_PORT']); "
            "print('line from rank %d of %s' % (r, os.environ['WORLD_SIZE']))")
    rc, out = bench.spawn_ranks([sys.executable, "-c", prog], 3)
    assert rc == 0 and out.decode().strip() == "line from rank 0 of 3"
    # rank 1 fails at once, rank 0 would wait for ever: the run fails and nobody is left behind
    prog = ("import os, sys, time; r = int(os.environ['RANK']);\n"
            "if r == 1: sys.exit(7)\n"
            "print('started', flush=True); time.sleep(600)")
    import time
    t0 = time.monotonic()
    rc, out = bench.spawn_ranks([sys.executable, "-c", prog], 2, grace_s=1.0)
    assert rc != 0 and time.monotonic() - t0 < 60
    assert out.decode().strip() == "started"


def test_bench_refuses_a_rank_count_that_differs_from_gpus():
    import subprocess
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, timeout=300)
    assert p.returncode == 2 and b"must agree" in p.stderr and p.stdout == b""
