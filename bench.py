#!/usr/bin/env python3
"""Benchmark of the ORB hot path on MI355X.

One step = one batch of KITTI-shaped frames (1241x376, 2000 features, 8 levels) through
the extractor, then frame-to-frame SearchByProjection (th = 7) on the consecutive
pairs of that batch.  Inputs are synthetic and resident in HBM before the timed region.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]

For N > 1 launch with torch.distributed.run (one rank per GPU); every rank extracts its
own batch (weak scaling, no collective on that path).  The Hamming kNN-2 leg shards the
database over the ranks and merges per-query (best, second, index) after an RCCL
all-gather.

Rank 0 prints one JSON line.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ROWS, COLS, NFEAT, NLEVELS = 376, 1241, 2000, 8
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak
VALU_PEAK_LANEOPS = 256 * 64 * 2.4e9  # CUs x lanes x clock: peak 32-bit integer lane-ops/s
MFMA_I8_PEAK_OPS = 2 * 1024 * 1024 * 2.4e9  # 2 x (SIMDs x int8 multiply-adds per cycle x clock): dense int8 peak


def level_sizes(rows, cols, nlevels=8, f=1.2):
    sf = np.ones(nlevels, np.float32)
    for i in range(1, nlevels):
        sf[i] = np.float32(np.float64(sf[i - 1]) * np.float64(np.float32(f)))
    inv = (np.float32(1.0) / sf).astype(np.float32)
    return [(int(np.rint(np.float32(cols) * inv[l])), int(np.rint(np.float32(rows) * inv[l]))) for l in range(nlevels)]


def algorithmic_bytes(rows, cols):
    """Compulsory bytes per frame of the three bandwidth-bound stages."""
    sz = level_sizes(rows, cols)
    S = sum(w * h for w, h in sz)
    P = sum((w + 38) * (h + 38) for w, h in sz)
    S0, S7 = sz[0][0] * sz[0][1], sz[-1][0] * sz[-1][1]
    return {"pyramid": (S0 + (S - S7)) + P, "fast": P, "blur": 2 * S}


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
    ap.add_argument("--cpu-frames", type=int, default=128, help="frames of the CPU baseline sample (0 = skip)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node N for --gpus N"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    from orb_slam2_amd import _lib
    from orb_slam2_amd._lib import FrameBounds, check
    from orb_slam2_amd.extractor import ORBextractor
    from orb_slam2_amd.synth import synth_descriptors, synth_sequence
    L = _lib.lib()

    B = args.batch
    # ---------------- inputs (not timed) ----------------
    seq = synth_sequence(ROWS, COLS, B, seed=1000 + rank, step=2)
    d_img = torch.from_numpy(seq).to(dev)
    ex = ORBextractor(NFEAT, 1.2, NLEVELS, 20, 7, device=local)
    max_kp = ex.max_keypoints(ROWS, COLS)
    # two sets of extractor outputs: with --overlap the matcher reads one while the next step fills the other
    nsets = 2 if args.overlap else 1
    kps_sets = [torch.zeros((B, max_kp, 6), dtype=torch.float32, device=dev) for _ in range(nsets)]
    desc_sets = [torch.zeros((B, max_kp, 32), dtype=torch.uint8, device=dev) for _ in range(nsets)]
    nout_sets = [torch.zeros(B, dtype=torch.int32, device=dev) for _ in range(nsets)]
    d_kps, d_desc, d_nout = kps_sets[0], desc_sets[0], nout_sets[0]
    s_ex = torch.cuda.current_stream()
    s_m = torch.cuda.Stream(device=dev) if args.overlap else s_ex
    stream = s_ex.cuda_stream

    def extract(s=0):
        ex.extract_batch_device(d_img.data_ptr(), B, ROWS, COLS, kps_sets[s].data_ptr(), desc_sets[s].data_ptr(),
                                nout_sets[s].data_ptr(), max_kp, stream=s_ex.cuda_stream)

    extract()
    torch.cuda.synchronize()
    # map points of frame p: its keypoints back-projected at depth Z; the camera moves
    # +t along x per frame so that the image content shifts 2 px per frame
    fx = fy = 700.0
    cx, cy, Z, tstep = COLS / 2.0, ROWS / 2.0, 10.0, 2.0 * 10.0 / 700.0
    npairs = B - 1
    kx, ky = d_kps[:npairs, :, 0], d_kps[:npairs, :, 1]
    cam = (torch.arange(npairs, device=dev, dtype=torch.float32) * tstep)[:, None]
    d_xw = torch.stack([(kx - cx) / fx * Z + cam, (ky - cy) / fy * Z, torch.full_like(kx, Z)], dim=2).contiguous()
    idx = torch.arange(max_kp, device=dev)[None, :]
    d_last_obs = torch.where(idx < d_nout[:npairs, None], 1, -1).to(torch.int32).contiguous()
    d_cur_obs = torch.full((npairs, max_kp), -1, dtype=torch.int32, device=dev)
    tcw = np.tile(np.eye(4, dtype=np.float32).reshape(1, 16), (npairs, 1))
    tcw[:, 3] = -(np.arange(1, npairs + 1, dtype=np.float32) * np.float32(tstep))
    d_tcw = torch.from_numpy(tcw).to(dev)
    sf = ex.GetScaleFactors()
    scratch_bytes = L.orb_match_scratch_bytes(npairs, max_kp)
    d_scratch = torch.zeros(scratch_bytes, dtype=torch.uint8, device=dev)
    d_match = torch.zeros((npairs, max_kp), dtype=torch.int32, device=dev)
    d_nm = torch.zeros(npairs, dtype=torch.int32, device=dev)
    bounds = FrameBounds(0.0, float(COLS), 0.0, float(ROWS))
    kp_bytes = max_kp * 24

    def match(s=0):
        n, k, d = nout_sets[s].data_ptr(), kps_sets[s].data_ptr(), desc_sets[s].data_ptr()
        check(L.orb_search_by_projection_frames_device(
            local, npairs, max_kp, max_kp, d_tcw.data_ptr(), fx, fy, cx, cy, bounds, sf.ctypes.data, NLEVELS,
            n, k, d_last_obs.data_ptr(), d_xw.data_ptr(), d,
            n + 4, k + kp_bytes, d + max_kp * 32, d_cur_obs.data_ptr(),
            7.0, 1, d_scratch.data_ptr(), scratch_bytes, d_match.data_ptr(), d_nm.data_ptr(), s_m.cuda_stream))

    ex_done = [torch.cuda.Event() for _ in range(nsets)]
    m_done = [torch.cuda.Event() for _ in range(nsets)]

    def step(k, ev_a=None, ev_b=None):
        """Extraction of step k on s_ex, its matching on s_m; set k % nsets is refilled only after the
        matcher of step k - nsets has read it.  With one stream the waits are no-ops."""
        s = k % nsets
        if args.overlap and k >= nsets:
            s_ex.wait_event(m_done[s])
        extract(s)
        if args.overlap:
            ex_done[s].record(s_ex)
            s_m.wait_event(ex_done[s])
        if ev_a is not None:
            ev_a.record(s_m)
        match(s)
        if ev_b is not None:
            ev_b.record(s_m)
        if args.overlap:
            m_done[s].record(s_m)

    # ---------------- timed region ----------------
    for k in range(args.warmup):
        step(k)
    barrier()
    L.orb_extractor_set_profiling(ex._h, 1)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev_m0 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ev_m1 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    t0 = time.perf_counter()
    ev0.record()
    for k in range(args.steps):
        step(k, ev_m0[k], ev_m1[k])
    ev1.record()
    barrier()
    t1 = time.perf_counter()
    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    elapsed = float(elapsed.item())
    stage_ms = np.zeros(5, np.float32)
    check(L.orb_extractor_stage_times(ex._h, stage_ms.ctypes.data))
    L.orb_extractor_set_profiling(ex._h, 0)
    match_ms = float(np.mean([a.elapsed_time(b) for a, b in zip(ev_m0, ev_m1)]))
    nkp = int(d_nout.sum().item())
    nmatch = int(d_nm.sum().item())
    frames_per_s = world * B * args.steps / elapsed

    # ---------------- kNN-2 leg (C5 shape, database sharded over the ranks) ----------------
    knn = None
    if args.knn > 0:
        nq = ndb = args.knn
        db, q = synth_descriptors(ndb, nq, seed=1)
        lo, hi = ndb * rank // world, ndb * (rank + 1) // world
        d_db = torch.from_numpy(db[lo:hi]).to(dev)
        d_q = torch.from_numpy(q).to(dev)
        res = torch.zeros((3, nq), dtype=torch.int32, device=dev)
        gathered = torch.zeros((world, 3, nq), dtype=torch.int32, device=dev)
        out = torch.zeros((3, nq), dtype=torch.int32, device=dev)
        acc = torch.zeros(nq, dtype=torch.uint8, device=dev)

        def knn_step():
            check(L.orb_knn2_device(local, d_q.data_ptr(), nq, d_db.data_ptr(), hi - lo, lo, 0.7,
                                    res[0].data_ptr(), res[1].data_ptr(), res[2].data_ptr(), None, stream))
            if world > 1:
                dist.all_gather_into_tensor(gathered, res)
                g = gathered.permute(1, 0, 2).contiguous()  # [3][world][nq]
            else:
                g = res.view(3, 1, nq)
            check(L.orb_knn2_merge_device(local, world, nq, g[0].data_ptr(), g[1].data_ptr(), g[2].data_ptr(), 0.7,
                                          out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), acc.data_ptr(), stream))

        small = min(nq, 65536)
        check(L.orb_knn2_device(local, d_q.data_ptr(), small, d_db.data_ptr(), min(hi - lo, 65536), lo, 0.7,
                                res[0].data_ptr(), res[1].data_ptr(), res[2].data_ptr(), None, stream))  # warm-up
        barrier()
        tk0 = time.perf_counter()
        knn_step()
        barrier()
        tk = torch.tensor([time.perf_counter() - tk0], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(tk, op=dist.ReduceOp.MAX)
        tk = float(tk.item())
        pairs = float(nq) * float(ndb)
        knn = {"n_query": nq, "n_database": ndb, "ratio": 0.7, "seconds": round(tk, 4),
               "matches_per_s": round(nq / tk, 1), "pairs_per_s": round(pairs / tk, 1),
               "accepted": int(acc.sum().item())}
        if hi - lo >= 2048 and nq >= 256:
            # the matrix-core kernel: one descriptor pair = 256 int8 multiply-adds
            knn["kernel"] = "knn2_mfma_kernel"
            knn["roofline"] = {"bound": "mfma", "achieved": round(512 * pairs / tk / 1e12, 1),
                               "peak": round(world * MFMA_I8_PEAK_OPS / 1e12, 1), "unit": "TOP/s",
                               "frac": round(512 * pairs / tk / (world * MFMA_I8_PEAK_OPS), 4)}
        else:
            # the popcount kernel: 16 vector instructions per pair
            knn["kernel"] = "knn2_kernel"
            knn["roofline"] = {"bound": "valu", "achieved": round(16 * pairs / tk / 1e12, 2),
                               "peak": round(world * VALU_PEAK_LANEOPS / 1e12, 2), "unit": "T lane-ops/s",
                               "frac": round(16 * pairs / tk / (world * VALU_PEAK_LANEOPS), 4)}

    # ---------------- CPU baseline (oracle, one core, rank 0, bounded sample) ----------------
    cpu = None
    if rank == 0 and world == 1 and args.cpu_frames > 1:
        from oracle import oracle as O
        n = min(args.cpu_frames, B)
        ref = O.Extractor(NFEAT, 1.2, NLEVELS, 20, 7)
        O.lib()
        tc0 = time.perf_counter()
        prev = None
        for k in range(n):
            kps, desc = ref(seq[k])
            if prev is not None:
                pk, pd = prev
                xw = np.stack([(pk["x"] - cx) / fx * Z + (k - 1) * tstep, (pk["y"] - cy) / fy * Z,
                               np.full(len(pk), Z)], 1).astype(np.float32)
                T = np.eye(4, dtype=np.float32)
                T[0, 3] = -np.float32(k * tstep)
                O.search_by_projection_frames(
                    T, (fx, fy, cx, cy), (0.0, float(COLS), 0.0, float(ROWS)), sf,
                    dict(mp_obs=np.ones(len(pk), np.int32), xw=xw, mp_desc=pd, octave=pk["octave"], angle=pk["angle"]),
                    dict(x=kps["x"], y=kps["y"], octave=kps["octave"], angle=kps["angle"], desc=desc,
                         obs=np.full(len(kps), -1, np.int32)), 7.0, True)
            prev = (kps, desc)
        tc = time.perf_counter() - tc0
        cpu = {"value": round(n / tc, 2), "unit": "frames/s", "cores": 1, "kind": "port",
               "sample": "%d frames of the same sequence: oracle extraction + SearchByProjection on consecutive pairs" % n}
        if knn is not None:
            # the oracle's sequential kNN-2 on a slice of the same descriptors, reported per second of one core
            sq, sdb = q[:2000], db[:100000]
            tq0 = time.perf_counter()
            O.knn2_bruteforce(sq, sdb, 0.7)
            tq = time.perf_counter() - tq0
            knn["cpu_baseline"] = {"pairs_per_s": round(len(sq) * len(sdb) / tq, 1), "cores": 1, "kind": "port",
                                   "sample": "2000 queries x 100000 database rows of the same data"}

    if rank == 0:
        # HBM bytes per launch from the PMC passes (tools/prof_pmc.sh + tools/pmc_to_traffic.py),
        # valid for the batch size they were taken at
        traffic = {}
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "r1_traffic.json")))
            if tj["frames_per_launch"] == B:
                traffic = {k.replace("_kernel", ""): v["hbm_bytes_per_launch"] for k, v in tj["kernels"].items()}
        except (OSError, ValueError, KeyError):
            pass
        ab = algorithmic_bytes(ROWS, COLS)
        names = ["pyramid", "fast", "quadtree", "blur", "orient_desc"]
        stages = {nm: round(float(ms), 4) for nm, ms in zip(names, stage_ms)}
        stages["search_by_projection"] = round(match_ms, 4)
        gbs = {k: ab[k] * B / (stages[k] * 1e-3) / 1e9 for k in ab}
        dom = max(ab, key=lambda k: stages[k])
        out = {
            "metric": "ORB frames/s (1241x376, 2000kp, 8 lvl) + Hamming kNN-2 matches/s at 1-8 GPU",
            "value": round(frames_per_s, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "KITTI-shaped 1241x376 sequence, 2000 features, 8 levels, scale 1.2, FAST 20/7; "
                                   "extract + frame-to-frame SearchByProjection(th=7)",
                       "frames_per_gpu_per_step": B, "streams": 2 if args.overlap else 1, "pairs_per_gpu_per_step": npairs,
                       "keypoints_per_frame": round(nkp / B, 1), "matches_per_pair": round(nmatch / npairs, 1)},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(gbs[dom], 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(gbs[dom] / HBM_PEAK_GBS, 4), "traffic": traffic.get(dom),
                         "algorithmic_bytes_per_launch": ab[dom] * B, "algorithmic_bytes_per_frame": ab[dom]},
            "stages_ms_per_step": stages,
            "stage_gbs": {k: round(v, 1) for k, v in gbs.items()},
            "cpu_baseline": cpu,
            "knn2": knn,
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
