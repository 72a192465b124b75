#!/usr/bin/env python3
"""Benchmark of the ORB hot path on MI355X.

One step = one batch of KITTI-shaped frames (1241x376, 2000 features, 8 levels) through
the extractor, then frame-to-frame SearchByProjection (th = 7) on the consecutive
pairs of that batch.  Inputs are synthetic and resident in HBM before the timed region.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]

For N > 1 there is one process (rank) per GPU: under torch.distributed.run the ranks exist
already (RANK / WORLD_SIZE in the environment); a plain `python bench.py --gpus N` starts its N
ranks itself, as child processes, before anything in the parent has touched a GPU, relays rank 0's
line and exits non-zero if any rank did.  Every rank extracts its own batch (weak scaling, no
collective on that path).  The Hamming kNN-2 leg shards the database over the ranks and merges
per-query (best, second, index) records after an RCCL all-gather.

Rank 0 prints one JSON line.
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ROWS, COLS, NFEAT, NLEVELS = 376, 1241, 2000, 8
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak
VALU_PEAK_LANEOPS = 256 * 64 * 2.4e9  # CUs x lanes x clock: peak 32-bit integer lane-ops/s
MFMA_I8_PEAK_OPS = 2 * 1024 * 1024 * 2.4e9  # 2 x (SIMDs x int8 multiply-adds per cycle x clock): dense int8 peak
VALU_ISSUE_PEAK = 1024 * 2.4e9 / 4  # wave-level vector instructions per second: 1024 SIMDs, one per 4 cycles each


def level_sizes(rows, cols, nlevels=8, f=1.2):
    sf = np.ones(nlevels, np.float32)
    for i in range(1, nlevels):
        sf[i] = np.float32(np.float64(sf[i - 1]) * np.float64(np.float32(f)))
    inv = (np.float32(1.0) / sf).astype(np.float32)
    return [(int(np.rint(np.float32(cols) * inv[l])), int(np.rint(np.float32(rows) * inv[l]))) for l in range(nlevels)]


def algorithmic_bytes(rows, cols, blur_fused=False):
    """Compulsory bytes per frame of the three bandwidth-bound stages.  With the blur made inside the
    descriptor kernel (no blurred level is written) there is no blur stage: its bytes, read S + write S
    of the reference's GaussianBlur pass, are priced against that kernel's time instead."""
    sz = level_sizes(rows, cols)
    S = sum(w * h for w, h in sz)
    P = sum((w + 38) * (h + 38) for w, h in sz)
    S0, S7 = sz[0][0] * sz[0][1], sz[-1][0] * sz[-1][1]
    return {"pyramid": (S0 + (S - S7)) + P, "fast": P, "orient_desc" if blur_fused else "blur": 2 * S}


def spawn_ranks(cmd, n, env=None, grace_s=20.0, poll_s=0.2):
    """Runs `cmd` (an argv list) as n child processes with RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT set, waits for all of them and returns (exit code, rank 0's stdout as
    bytes).  Rank 0's stdout is captured; the other ranks' stdout goes to this process's stderr.
    The exit code is 0 only if every rank's was; once a rank has failed the others get grace_s
    seconds to end by themselves (they may be waiting for it in a collective) and are then ended,
    each by its own process id.  The caller must not have touched a GPU: the children are plain
    child processes, nothing is replaced."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    base = dict(os.environ if env is None else env)
    base.update(WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    procs = []
    for r in range(n):
        e = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen(cmd, env=e, stdout=subprocess.PIPE if r == 0 else sys.stderr, stdin=subprocess.DEVNULL))
    out0 = []
    import threading
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    failed_at = None
    while True:
        codes = [p.poll() for p in procs]
        if all(c is not None for c in codes):
            break
        if failed_at is None and any(c not in (None, 0) for c in codes):
            failed_at = time.monotonic()
        if failed_at is not None and time.monotonic() - failed_at > grace_s:
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            t_end = time.monotonic() + 5.0
            while time.monotonic() < t_end and any(p.poll() is None for p in procs):
                time.sleep(poll_s)
            for p in procs:
                if p.poll() is None:
                    p.kill()
        time.sleep(poll_s)
    reader.join(10.0)
    codes = [p.wait() for p in procs]
    rc = next((c for c in codes if c != 0), 0)
    return (rc if rc > 0 else (1 if rc != 0 else 0)), (out0[0] if out0 else b"")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="frames per GPU per step")
    ap.add_argument("--knn", type=int, default=1000000, help="queries = database rows of the kNN-2 leg (0 = skip)")
    ap.add_argument("--overlap", type=int, default=1,
                    help="1: the matcher of step k runs on a second stream beside the extraction of step k+1 "
                         "(two sets of keypoint buffers); 0: one stream")
    ap.add_argument("--handles", type=int, default=1,
                    help="extractor handles that take the steps in turn, each on a stream of its own, so that step "
                         "k + 1 starts beside the end of step k (with --overlap 1; every handle holds its own pyramid "
                         "and work buffers).  2 is 1.3 %% faster over 80 steps and no faster over 20")
    ap.add_argument("--spinup-ms", type=float, default=100.0,
                    help="untimed steps are run for this long before the W warm-up steps: after the host-side set-up "
                         "the GPU takes tens of milliseconds of load to reach its steady state (20 steps after 5 "
                         "warm-up steps alone read 2 %% slower than after 60)")
    ap.add_argument("--split", type=int, default=1,
                    help="1: orb_extract_batch_device runs the batch as two staggered halves (its default); 0: on the "
                         "caller's stream alone")
    ap.add_argument("--cpu-frames", type=int, default=128, help="frames of the CPU baseline sample (0 = skip)")
    ap.add_argument("--knn-reps", type=int, default=3, help="timed repetitions of the kNN-2 leg (the median is reported)")
    ap.add_argument("--knn-popcount", type=int, default=1, help="1: also time the popcount kernel once on the same input")
    ap.add_argument("--knn-comm", choices=["auto", "torch", "rccl"], default="auto",
                    help="exchange of the sharded kNN-2 for N > 1: the library's own orb_knn2_sharded (ncclAllGather "
                         "inside: what a C++ host gets; the default), or torch.distributed all_gather + "
                         "orb_knn2_merge_packed_device (the code the gloo test runs; the default when the ranks share "
                         "one GPU, where RCCL cannot be used)")
    ap.add_argument("--host-reps", type=int, default=5, help="repetitions of the host-buffer batch call (0 = skip)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not started by a launcher: start the ranks from here (this process has imported nothing
        # that touches a GPU) and hand rank 0's line on
        rc, out = spawn_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], args.gpus)
        sys.stdout.flush()
        os.write(1, out)
        sys.exit(rc)

    # stdout carries the one JSON line and nothing else: libraries that print banners to file
    # descriptor 1 (RCCL does when a communicator is created) are sent to stderr instead
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d: the launcher's rank count and --gpus must agree\n" % (args.gpus, world))
        sys.exit(2)
    # Rehearsal on a box with fewer GPUs than ranks: ORBHIP_BENCH_SHARE_GPU=1 puts every rank on GPU 0
    # and the exchange on gloo (RCCL refuses two ranks on one GPU); never set for a measurement.
    share = os.environ.get("ORBHIP_BENCH_SHARE_GPU", "0") == "1"
    if share:
        local = 0
    knn_comm = args.knn_comm if args.knn_comm != "auto" else ("rccl" if world > 1 and not share else "torch")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    from orb_slam2_amd import _lib
    from orb_slam2_amd import dist as odist
    from orb_slam2_amd._lib import KP_DTYPE, check
    from orb_slam2_amd.chain import DeviceChain
    from orb_slam2_amd.synth import synth_descriptors, synth_sequence
    L = _lib.lib()

    B = args.batch
    # ---------------- inputs (not timed) ----------------
    seq = synth_sequence(ROWS, COLS, B, seed=1000 + rank, step=2)
    # two sets of extractor outputs: with --overlap the matcher reads one while the next step fills the other
    nsets = 2 if args.overlap else 1
    n_handles = max(1, min(args.handles, nsets)) if args.overlap else 1
    chain = DeviceChain(ROWS, COLS, B, nfeatures=NFEAT, nlevels=NLEVELS, device=local, nsets=nsets, th=7.0,
                        check_orientation=True, n_handles=n_handles)
    for e in chain.exs:
        check(L.orb_extractor_set_batch_split(e._h, args.split))
    chain.load(seq)  # uploads the frames, extracts once, builds every frame's map points and the pair poses
    ex, max_kp, npairs = chain.ex, chain.max_kp, chain.npairs
    d_nout, d_nm = chain.nout[0], chain.d_nm
    s_ex = torch.cuda.current_stream()
    s_m = torch.cuda.Stream(device=dev) if args.overlap else s_ex
    stream = s_ex.cuda_stream
    # extraction streams, one per handle
    s_exs = [s_ex] + [torch.cuda.Stream(device=dev) for _ in range(n_handles - 1)]

    def match(s=0):
        chain.match(s, s_m.cuda_stream)

    ex_done = [torch.cuda.Event() for _ in range(nsets)]
    m_done = [torch.cuda.Event() for _ in range(nsets)]

    def step(k, ev_a=None, ev_b=None):
        """Extraction of step k by handle k % n_handles on that handle's stream, its matching on s_m; set
        k % nsets is refilled only after the matcher of step k - nsets has read it.  With one stream the
        waits are no-ops."""
        s, hd = k % nsets, k % n_handles
        se = s_exs[hd]
        if args.overlap and k >= nsets:
            se.wait_event(m_done[s])
        chain.extract(s, se.cuda_stream, handle=hd)
        if args.overlap:
            ex_done[s].record(se)
            s_m.wait_event(ex_done[s])
        if ev_a is not None:
            ev_a.record(s_m)
        match(s)
        if ev_b is not None:
            ev_b.record(s_m)
        if args.overlap:
            m_done[s].record(s_m)

    # ---------------- timed region ----------------
    k0 = 0
    if args.spinup_ms > 0:
        t_spin = time.perf_counter()
        while (time.perf_counter() - t_spin) * 1e3 < args.spinup_ms:
            for _ in range(4):
                step(k0)
                k0 += 1
            torch.cuda.synchronize()
        k0 += (-k0) % (nsets * n_handles)  # (the steps below start at set 0, handle 0 again)
    for k in range(args.warmup):
        step(k0 + k)
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k)
    barrier()
    t1 = time.perf_counter()
    red = torch.device("cpu") if share else dev  # where the timing reductions live (gloo: host)
    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device=red)
    if world > 1:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    elapsed = float(elapsed.item())
    # ---------------- stage times: a pass of its own, after the timed region ----------------
    # Every kernel alone on the GPU: the whole batch on one stream, the matcher after the extraction
    # instead of beside the next one, HIP events between the stages on that stream (inside the library
    # for the extractor's stages).  These are the kernel times `roofline` and `stage_gbs` are computed
    # from; in the timed region above the stages of two half batches and the matcher overlap.
    n_prof = max(1, min(args.steps, 10))
    L.orb_extractor_set_profiling(ex._h, 1)
    ev_m0 = [torch.cuda.Event(enable_timing=True) for _ in range(n_prof)]
    ev_m1 = [torch.cuda.Event(enable_timing=True) for _ in range(n_prof)]
    s_last = (args.steps - 1) % nsets if args.steps > 0 else 0
    for k in range(n_prof):
        chain.extract(s_last, s_ex.cuda_stream)
        ev_m0[k].record(s_ex)
        chain.match(s_last, s_ex.cuda_stream)
        ev_m1[k].record(s_ex)
    torch.cuda.synchronize()
    stage_ms = np.zeros(5, np.float32)
    check(L.orb_extractor_stage_times(ex._h, stage_ms.ctypes.data))
    L.orb_extractor_set_profiling(ex._h, 0)
    match_ms = float(np.mean([a.elapsed_time(b) for a, b in zip(ev_m0, ev_m1)]))
    nkp = int(d_nout.sum().item())
    nmatch = int(d_nm.sum().item())
    frames_per_s = world * B * args.steps / elapsed

    # ---------------- host-buffer forms of the same extraction (rank 0, one GPU) ----------------
    # value_pcie_inclusive: host images in, host keypoints out through orb_extract_batch (the same B
    # frames); latency_ms_single_frame: one orb_extract call, the way Frame::ExtractORB calls the
    # reference (src/Frame.cc:220-228), host to host, median.
    host = None
    if rank == 0 and world == 1 and args.host_reps > 0:
        hk = np.zeros((B, max_kp), KP_DTYPE)
        hd = np.zeros((B, max_kp, 32), np.uint8)
        hn = np.zeros(B, np.int32)

        def host_batch(n):
            check(L.orb_extract_batch(ex._h, n, seq.ctypes.data, ROWS, COLS, COLS, ROWS * COLS, max_kp,
                                      hk.ctypes.data, hd.ctypes.data, hn.ctypes.data))
        host_batch(B)
        tb = []
        for _ in range(args.host_reps):
            t = time.perf_counter(); host_batch(B); tb.append(time.perf_counter() - t)
        host_batch(1)
        tl = []
        for _ in range(100):
            t = time.perf_counter(); host_batch(1); tl.append(time.perf_counter() - t)
        host = {"value_pcie_inclusive": round(B / float(np.median(tb)), 1),
                "latency_ms_single_frame": round(1e3 * float(np.median(tl)), 4),
                "latency_ms_single_frame_p90": round(1e3 * float(np.percentile(tl, 90)), 4)}
        assert int(hn.sum()) == nkp, "host-buffer and device-buffer extraction disagree on the keypoint count"
        if npairs >= 1:
            # the call a tracking thread makes next (src/Tracking.cc:1065): SearchByProjection(Cur, Last, 7, mono)
            # on host arrays, one frame pair (frames 0 and 1 of the batch), host to host
            s_last = (args.steps - 1) % nsets if args.steps > 0 else 0
            g_nout, g_kps, g_desc, g_match, g_nm = chain.results(s_last)
            n0, n1 = int(g_nout[0]), int(g_nout[1])
            kl, kc = g_kps[0, :n0], g_kps[1, :n1]
            as_c = lambda a, t: np.ascontiguousarray(a, dtype=t)
            l_oct, l_ang = as_c(kl["octave"], np.int32), as_c(kl["angle"], np.float32)
            c_x, c_y = as_c(kc["x"], np.float32), as_c(kc["y"], np.float32)
            c_oct, c_ang = as_c(kc["octave"], np.int32), as_c(kc["angle"], np.float32)
            l_desc, c_desc = as_c(g_desc[0, :n0], np.uint8), as_c(g_desc[1, :n1], np.uint8)
            l_xw = as_c(chain.d_xw[0, :n0].cpu().numpy(), np.float32)
            l_obs = as_c(chain.d_last_obs[0, :n0].cpu().numpy(), np.int32)
            tcw = as_c(chain.d_tcw[0].cpu().numpy(), np.float32)
            c_obs = np.full(n1, -1, np.int32)
            nl, nc = np.array([n0], np.int32), np.array([n1], np.int32)
            mt, nm1 = np.zeros(n1, np.int32), np.zeros(1, np.int32)
            sc = chain.scene

            def host_match():
                check(L.orb_search_by_projection_frames(
                    local, 1, n0, n1, tcw.ctypes.data, float(sc.fx), float(sc.fy), float(sc.cx), float(sc.cy),
                    chain.bounds, chain.sf.ctypes.data, NLEVELS, nl.ctypes.data, l_obs.ctypes.data, l_xw.ctypes.data,
                    l_desc.ctypes.data, l_oct.ctypes.data, l_ang.ctypes.data, nc.ctypes.data, c_x.ctypes.data,
                    c_y.ctypes.data, c_oct.ctypes.data, c_ang.ctypes.data, c_desc.ctypes.data, c_obs.ctypes.data,
                    7.0, 1, mt.ctypes.data, nm1.ctypes.data))
            for _ in range(3):
                host_match()
            assert int(nm1[0]) == int(g_nm[0]) and np.array_equal(mt, g_match[0, :n1]), \
                "host-buffer and device-buffer SearchByProjection disagree"
            tm, tt = [], []
            for _ in range(200):
                t = time.perf_counter(); host_match(); tm.append(time.perf_counter() - t)
            for _ in range(200):
                t = time.perf_counter(); host_batch(1); host_match(); tt.append(time.perf_counter() - t)
            host["latency_ms_single_pair_match"] = round(1e3 * float(np.median(tm)), 4)
            host["latency_ms_frame_total"] = round(1e3 * float(np.median(tt)), 4)
            host["latency_ms_frame_total_p90"] = round(1e3 * float(np.percentile(tt, 90)), 4)

    # ---------------- kNN-2 leg (C5 shape, database sharded over the ranks) ----------------
    knn = None
    if args.knn > 0:
        nq = ndb = args.knn
        db, q = synth_descriptors(ndb, nq, seed=1)
        lo, hi = odist.shard_bounds(ndb, world, rank)
        d_db = torch.from_numpy(db[lo:hi]).to(dev)
        d_q = torch.from_numpy(q).to(dev)
        # the code path tests/test_dist_cpu.py runs with gloo: shard scan, one all-gather, merge
        local_search = odist.hip_local_search(d_q, d_db, lo, 0.7, local, stream)
        merge = odist.hip_merge(0.7, local, stream)
        merge_packed = odist.hip_merge_packed(0.7, local, stream)
        res = torch.zeros((3, nq), dtype=torch.int32, device=dev)
        small = min(nq, 65536)
        check(L.orb_knn2_device(local, d_q.data_ptr(), small, d_db.data_ptr(), min(hi - lo, 65536), lo, 0.7,
                                res[0].data_ptr(), res[1].data_ptr(), res[2].data_ptr(), None, stream))  # warm-up
        rk = None
        if knn_comm == "rccl" or world == 1:
            rk = odist.RcclKnn(local, rank, world)
            rk.search(d_q[:small], d_db[:min(hi - lo, 65536)], lo, 0.7, stream)  # communicator and kernels warm
        times, tparts = [], []
        for _ in range(max(args.knn_reps, 1)):
            barrier()
            tk0 = time.perf_counter()
            if knn_comm == "rccl":
                out, acc = rk.search(d_q, d_db, lo, 0.7, stream)
            else:
                tm_ = {}
                out, acc = odist.knn2_sharded(local_search, merge, nq, dev, world, merge_packed=merge_packed, timings=tm_)
                tparts.append((1e3 * tm_["scan_s"], 1e3 * tm_["exchange_s"]))
            barrier()
            if knn_comm == "rccl":
                tparts.append(rk.times())
            tk = torch.tensor([time.perf_counter() - tk0], dtype=torch.float64, device=red)
            if world > 1:
                dist.all_reduce(tk, op=dist.ReduceOp.MAX)
            times.append(float(tk.item()))
        tk = float(np.median(times))
        pairs = float(nq) * float(ndb)
        knn = {"n_query": nq, "n_database": ndb, "ratio": 0.7, "seconds": round(tk, 4), "repetitions": len(times),
               "seconds_all": [round(t, 4) for t in times],
               "matches_per_s": round(nq / tk, 1), "pairs_per_s": round(pairs / tk, 1),
               "accepted": int(acc.sum().item())}
        # the exchange step on its own: all-gather + merge (rccl: HIP events on the stream inside the
        # library; torch: host clock around the collective and the merge, device synchronised), and
        # how many ranks the collective really had
        parts = torch.tensor([float(np.median([p[0] for p in tparts])), float(np.median([p[1] for p in tparts]))],
                             dtype=torch.float64, device=red)
        if world > 1:
            dist.all_reduce(parts, op=dist.ReduceOp.MAX)
        knn["exchange"] = {"via": "orb_knn2_sharded (ncclAllGather in the library)" if knn_comm == "rccl" else
                           "torch.distributed.all_gather_into_tensor (%s) + orb_knn2_merge_packed_device" % (dist.get_backend() if world > 1 else "none"),
                           "bytes_per_query_per_rank": 8}
        knn["scan_ms"] = round(float(parts[0].item()), 3)
        knn["exchange_ms"] = round(float(parts[1].item()), 3)
        knn["ranks_in_collective"] = rk.info()[2] if knn_comm == "rccl" else (dist.get_world_size() if world > 1 else 1)
        if rk is not None and knn_comm != "rccl":
            # one GPU: the library's own sharded entry point on a communicator of one rank, same input
            o2, a2 = rk.search(d_q, d_db, lo, 0.7, stream)  # allocates its exchange buffers
            barrier()
            tk0 = time.perf_counter()
            o2, a2 = rk.search(d_q, d_db, lo, 0.7, stream)
            barrier()
            knn["c_abi_sharded"] = {"seconds": round(time.perf_counter() - tk0, 4),
                                    "identical": bool(torch.equal(o2, out) and torch.equal(a2, acc)),
                                    "scan_ms": round(rk.times()[0], 3), "exchange_ms": round(rk.times()[1], 3),
                                    "ranks_in_collective": rk.info()[2]}
        if rk is not None:
            rk.close()
        mfma = hi - lo >= 2048 and nq >= 256
        if mfma:
            # the matrix-core kernel: one descriptor pair = 256 int8 multiply-adds
            knn["kernel"] = "knn2_mfma_kernel"
            knn["roofline"] = {"bound": "mfma", "achieved": round(512 * pairs / tk / 1e12, 1),
                               "peak": round(world * MFMA_I8_PEAK_OPS / 1e12, 1), "unit": "TOP/s",
                               "frac": round(512 * pairs / tk / (world * MFMA_I8_PEAK_OPS), 4)}
        if not mfma or args.knn_popcount:
            # the popcount kernel (what small inputs get): 16 vector instructions per pair, priced against
            # the vector units' integer rate; run once on the same shard, results must be the same integers
            r2 = torch.zeros((3, nq), dtype=torch.int32, device=dev)
            barrier()
            tp0 = time.perf_counter()
            check(L.orb_knn2_device_algo(local, d_q.data_ptr(), nq, d_db.data_ptr(), hi - lo, lo, 0.7,
                                         r2[0].data_ptr(), r2[1].data_ptr(), r2[2].data_ptr(), None, stream, 1))
            barrier()
            tp = torch.tensor([time.perf_counter() - tp0], dtype=torch.float64, device=red)
            if world > 1:
                dist.all_reduce(tp, op=dist.ReduceOp.MAX)
            tp = float(tp.item())
            same = True
            if mfma:
                check(L.orb_knn2_device_algo(local, d_q.data_ptr(), nq, d_db.data_ptr(), hi - lo, lo, 0.7,
                                             res[0].data_ptr(), res[1].data_ptr(), res[2].data_ptr(), None, stream, 2))
                same = bool(torch.equal(res, r2))
            pop = {"kernel": "knn2_kernel", "seconds": round(tp, 4), "matches_per_s": round(nq / tp, 1),
                   "identical_to_mfma_kernel": same,
                   "roofline": {"bound": "valu", "achieved": round(16 * pairs / tp / 1e12, 2),
                                "peak": round(world * VALU_PEAK_LANEOPS / 1e12, 2), "unit": "T lane-ops/s",
                                "frac": round(16 * pairs / tp / (world * VALU_PEAK_LANEOPS), 4)}}
            if mfma:
                knn["popcount"] = pop
            else:
                knn["kernel"] = "knn2_kernel"
                knn["roofline"] = pop["roofline"]

    # ---------------- CPU baseline (oracle, one core, rank 0, bounded sample) + parity of the timed chain ------------
    cpu = None
    parity = None
    if rank == 0 and world == 1 and args.cpu_frames > 1:
        from oracle import oracle as O
        from oracle.chain_ref import chain_reference
        n = min(args.cpu_frames, B)
        O.lib()
        tc0 = time.perf_counter()
        frames, pairs_ref = chain_reference(chain.scene, seq[:n], NFEAT, NLEVELS)
        tc = time.perf_counter() - tc0
        cpu = {"value": round(n / tc, 2), "unit": "frames/s", "cores": 1, "kind": "port",
               "sample": "%d frames of the same sequence: oracle extraction + SearchByProjection on consecutive pairs" % n}
        # the same sample frame-parallel on the host's cores (16 on the one-GPU box; os.cpu_count() there is
        # the whole machine's): a fairer CPU figure than one core, still the plain restatement
        from oracle.chain_ref import chain_reference_parallel
        nthreads = int(os.environ.get("ORBHIP_CPU_THREADS", "16"))
        tp0 = time.perf_counter()
        frames_p, pairs_p = chain_reference_parallel(chain.scene, seq[:n], NFEAT, NLEVELS, threads=nthreads)
        tpar = time.perf_counter() - tp0
        same = all(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) for a, b in zip(frames, frames_p)) and \
            all(a[0] == b[0] and np.array_equal(a[1], b[1]) for a, b in zip(pairs_ref, pairs_p))
        cpu["parallel"] = {"value": round(n / tpar, 2), "unit": "frames/s", "cores": nthreads, "kind": "port",
                           "identical_to_one_core": bool(same)}
        # what the timed chain left on the GPU against what the oracle just computed for the same frames
        g_nout, g_kps, g_desc, g_match, g_nm = chain.results((args.steps - 1) % nsets if args.steps > 0 else 0)
        bad_frames = bad_pairs = 0
        for k in range(n):
            rk, rd = frames[k]
            m = int(g_nout[k])
            ok = m == len(rk) and all(np.array_equal(g_kps[k, :m][f], rk[f]) for f in ("x", "y", "octave", "response"))
            ok = ok and np.abs(g_kps[k, :m]["angle"] - rk["angle"]).max() < 1e-4 * 180 / np.pi
            ok = ok and (g_desc[k, :m] == rd).all(1).mean() >= 0.999
            bad_frames += not ok
        for p in range(n - 1):
            m = int(g_nout[p + 1])
            ok = int(g_nm[p]) == pairs_ref[p][0] and m == len(pairs_ref[p][1]) and np.array_equal(g_match[p, :m], pairs_ref[p][1])
            bad_pairs += not ok
        parity = {"parity_checked_frames": n, "parity_frame_mismatches": int(bad_frames),
                  "parity_checked_pairs": n - 1, "parity_mismatches": int(bad_pairs)}
        if knn is not None:
            # the oracle's sequential kNN-2 on a slice of the same descriptors, reported per second of one core
            sq, sdb = q[:2000], db[:100000]
            tq0 = time.perf_counter()
            rb, rs, ri, ra = O.knn2_bruteforce(sq, sdb, 0.7)
            tq = time.perf_counter() - tq0
            knn["cpu_baseline"] = {"pairs_per_s": round(len(sq) * len(sdb) / tq, 1), "cores": 1, "kind": "port",
                                   "sample": "2000 queries x 100000 database rows of the same data"}
            # and the same slice on the GPU, compared
            gb = torch.zeros((3, len(sq)), dtype=torch.int32, device=dev)
            ga = torch.zeros(len(sq), dtype=torch.uint8, device=dev)
            check(L.orb_knn2_device(local, d_q.data_ptr(), len(sq), d_db.data_ptr(), len(sdb), 0, 0.7, gb[0].data_ptr(),
                                    gb[1].data_ptr(), gb[2].data_ptr(), ga.data_ptr(), stream))
            torch.cuda.synchronize()
            gbn = gb.cpu().numpy()
            knn["parity_checked_queries"] = len(sq)
            knn["parity_mismatches"] = int((~((gbn[0] == rb) & (gbn[1] == rs) & (gbn[2] == ri) & (ga.cpu().numpy() == ra))).sum())

    if rank == 0:
        # HBM bytes per launch from the PMC passes (tools/prof_pmc.sh + tools/pmc_to_traffic.py),
        # valid for the batch size they were taken at
        traffic, launches, valu = {}, {}, {}
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "r3_traffic.json")))
            if tj["frames_per_launch"] == B:
                traffic = {k.replace("_kernel", ""): v.get("hbm_bytes_per_step", v["hbm_bytes_per_launch"]) for k, v in tj["kernels"].items()}
                launches = {k.replace("_kernel", ""): v.get("launches_per_step", 1) for k, v in tj["kernels"].items()}
                # the pyramid stage is several kernels (levels 0 + 1 in one, then one launch per level)
                pyr = [k for k in traffic if k.startswith("pyr_")]
                traffic["pyramid"] = sum(traffic[k] for k in pyr)
                launches["pyramid"] = sum(launches[k] for k in pyr)
                # wave-level vector instructions per step (SQ_INSTS_VALU of the same PMC passes)
                per = {k.replace("_kernel", "").split("<")[0]: v.get("valu_insts_per_launch", 0) * v.get("launches_per_step", 1)
                       for k, v in tj["kernels"].items()}
                valu = {"pyramid": sum(v for k, v in per.items() if k.startswith("pyr_")), "fast": per.get("fast", 0),
                        "quadtree": per.get("octree", 0), "orient_desc": per.get("orient_desc", 0),
                        "search_by_projection": per.get("match", 0)}
        except (OSError, ValueError, KeyError):
            pass
        names = ["pyramid", "fast", "quadtree", "blur", "orient_desc"]
        stages = {nm: round(float(ms), 4) for nm, ms in zip(names, stage_ms)}
        # the blur stage is empty when the descriptor kernel blurs its patches itself (the default)
        blur_fused = stages["blur"] < 0.05 * stages["orient_desc"]
        ab = algorithmic_bytes(ROWS, COLS, blur_fused)
        if blur_fused:
            stages["orient_desc"] = round(stages["orient_desc"] + stages.pop("blur"), 4)
        stages["search_by_projection"] = round(match_ms, 4)
        gbs = {k: ab[k] * B / (stages[k] * 1e-3) / 1e9 for k in ab}
        # the roofline object is for the stage that streams every pixel and is furthest below its
        # roofline: FAST or the pyramid (the descriptor kernel gathers patches: see stage_gbs)
        dom = min(("pyramid", "fast"), key=lambda k: gbs[k])
        out = {
            "metric": "ORB frames/s (1241x376, 2000kp, 8 lvl) + Hamming kNN-2 matches/s at 1-8 GPU",
            "value": round(frames_per_s, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "KITTI-shaped 1241x376 sequence, 2000 features, 8 levels, scale 1.2, FAST 20/7; "
                                   "extract + frame-to-frame SearchByProjection(th=7)",
                       "frames_per_gpu_per_step": B,
                       "streams": (n_handles * (2 if args.split and B >= 64 else 1) + 1) if args.overlap else 1,
                       "extractor_handles": n_handles, "spinup_ms": args.spinup_ms,
                       "batch_split": bool(args.split and B >= 64), "pairs_per_gpu_per_step": npairs,
                       "keypoints_per_frame": round(nkp / B, 1), "matches_per_pair": round(nmatch / npairs, 1)},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(gbs[dom], 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(gbs[dom] / HBM_PEAK_GBS, 4), "traffic": traffic.get(dom),
                         "algorithmic_bytes_per_launch": ab[dom] * B, "algorithmic_bytes_per_frame": ab[dom],
                         # kernels that take several launches per step (FAST: one per view height class) are
                         # priced as one: bytes and time of all of them, back to back on the stream
                         "launches_per_step": launches.get(dom, 1)},
            "stages_ms_per_step": stages,
            "stages_measured": "separate pass after the timed region: whole batch on one stream, every kernel alone "
                               "on the GPU (in the timed region two half batches and the matcher overlap, so the "
                               "stages add up to more than ms_per_step)",
            "stage_gbs": {k: round(v, 1) for k, v in gbs.items()},
            # what the stages are really bound by: the share of the chip's vector-instruction issue slots
            # (1024 SIMDs, one wave instruction per 4 cycles at 2.4 GHz) each kernel fills while it runs alone;
            # instruction counts from the committed PMC passes, times from this run
            "stage_valu_issue_frac": {k: round(valu[k] / (stages[k] * 1e-3) / VALU_ISSUE_PEAK, 3)
                                      for k in valu if valu[k] and stages.get(k)},
            "cpu_baseline": cpu,
            "knn2": knn,
        }
        if host:
            out.update(host)
        if parity:
            out.update(parity)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
