#!/usr/bin/env python3
"""Randomised parity sweep of the extractor against the oracle (run on a GPU box):
random image sizes, feature counts, level counts, scale factors, thresholds and contents.
    python tools/fuzz_extract.py [n_cases] [seed]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as O
from orb_slam2_amd import ORBextractor
from orb_slam2_amd._lib import OrbHipError
from orb_slam2_amd.synth import synth_image

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
t0 = time.time()
for case in range(n_cases):
    rows = int(rng.integers(120, 800))
    cols = int(rows * rng.uniform(0.8, 3.2))
    nlev = int(rng.integers(1, 9))
    if rng.integers(0, 8) == 0:  # wide and short: up to four tile columns per level, the widest level the kernels take
        rows, cols, nlev = int(rng.integers(100, 180)), int(rng.integers(2000, 4120)), int(rng.integers(1, 4))
    sf = float(rng.choice([1.2, 1.2, 1.1, 1.3, 1.44]))
    if sf == 1.1 and rng.integers(0, 3) == 0:
        nlev = int(rng.integers(9, 13))  # (the build's limit is 12 levels)
    nf = int(rng.integers(50, 4500))
    ini, mn = int(rng.integers(8, 40)), int(rng.integers(2, 8))
    kind = rng.integers(0, 4)
    img = synth_image(rows, cols, int(rng.integers(0, 1 << 30)), n_shapes=None if kind else 30)
    if kind == 2:
        img = (img // 8 * 8).astype(np.uint8)            # plateaus: many equal scores
    if kind == 3:
        img = np.clip(img.astype(np.int16) + rng.integers(-25, 26, img.shape), 0, 255).astype(np.uint8)  # noisy: min-threshold cells
    # every sixth case with the arithmetic of an older OpenCV (rounded blur taps that sum to 257, saturated;
    # 3.4 also the 14-bit grey weights), in the oracle (a global setting) and in the handle
    compat = [(4, 5), (4, 2), (3, 4)][int(rng.integers(1, 3))] if case % 6 == 0 else (4, 5)
    O.set_opencv_compat(*compat)
    tag = "%dx%d nf=%d lv=%d sf=%.2f th=%d/%d kind=%d cv%d.%d" % (rows, cols, nf, nlev, sf, ini, mn, kind, compat[0], compat[1])
    try:
        ex = ORBextractor(nf, sf, nlev, ini, mn)
        ex.set_opencv_compat(*compat)
        kps, desc = ex(img)
    except OrbHipError as e:
        print("case %d %s: refused (%s)" % (case, tag, str(e)[:60]))
        continue
    rk, rd = O.Extractor(nf, sf, nlev, ini, mn)(img)
    ok = len(kps) == len(rk)
    if ok:
        ok = all(np.array_equal(kps[f], rk[f]) for f in ("x", "y", "octave", "response", "size"))
        ok = ok and (len(kps) == 0 or (np.abs(kps["angle"] - rk["angle"]).max() <= 1e-4 * 180 / np.pi and (desc == rd).all(1).mean() >= 0.999))
    if not ok:
        bad += 1
        print("MISMATCH case %d %s: %d vs %d keypoints" % (case, tag, len(kps), len(rk)))
    elif case % 10 == 0:
        print("case %d %s ok (%d kp, desc identical %.4f) %.0fs" % (case, tag, len(kps), (desc == rd).all(1).mean() if len(kps) else 1.0, time.time() - t0), flush=True)
    # the same image as the first and last frame of a batch (batches of more than 8 frames take other
    # launches: levels 0 and 1 in one pass over the image, taller resize tiles, FAST by view height),
    # the frames between them shifted copies; keypoints, descriptors and every padded level of both
    nb = int(rng.integers(9, 14))
    batch = np.stack([img] + [np.roll(img, int(rng.integers(1, 40)), axis=1) for _ in range(nb - 2)] + [img])
    ref = O.Extractor(nf, sf, nlev, ini, mn)
    rk, rd = ref(img)
    bk, bd = ex.extract_batch(batch)
    what = []  # what differs, for the report
    for f in (0, nb - 1):
        if len(bk[f]) != len(rk):
            what.append("frame %d: %d vs %d keypoints" % (f, len(bk[f]), len(rk)))
        else:
            what += ["frame %d: %s" % (f, n_) for n_ in ("x", "y", "octave", "response", "size") if not np.array_equal(bk[f][n_], rk[n_])]
            if len(rk) and (bd[f] == rd).all(1).mean() < 0.999:
                what.append("frame %d: descriptors" % f)
        for l in range(nlev):
            got, want = ex.pyramid_level(l, f, padded=True), ref.pyramid_level(l)
            if not np.array_equal(got, want):
                ys, xs = np.nonzero(got != want) if got.shape == want.shape else ([], [])
                what.append("frame %d: level %d differs in %d bytes, rows %s..%s cols %s..%s of %s" % (
                    f, l, len(ys), min(ys, default=-1), max(ys, default=-1), min(xs, default=-1), max(xs, default=-1), got.shape))
    if what:
        bad += 1
        print("MISMATCH (batch of %d) case %d %s: %s" % (nb, case, tag, "; ".join(what[:6])), flush=True)
    # every fifth case: the same handle on an image of another size (its buffers and captured launches are
    # rebuilt), then on a colour version of the first image (converted while level 0 is written), then on the first
    # image again, which must give the first result
    if case % 5 == 0:
        what = []
        r3, c3 = max(100, rows - int(rng.integers(1, 60))), max(130, cols - int(rng.integers(1, 90)))
        crop = np.ascontiguousarray(img[:r3, :c3])
        try:
            k3, d3 = ex(crop)
            rk3, rd3 = O.Extractor(nf, sf, nlev, ini, mn)(crop)
            if len(k3) != len(rk3) or not all(np.array_equal(k3[n_], rk3[n_]) for n_ in ("x", "y", "octave", "response")) \
                    or (len(rk3) and (d3 == rd3).all(1).mean() < 0.999):
                what.append("second size %dx%d" % (r3, c3))
        except OrbHipError:
            pass
        ch, rgb = int(rng.choice([3, 4])), bool(rng.integers(0, 2))
        col = np.zeros((rows, cols, ch), np.uint8)
        wts = rng.uniform(0.6, 1.0, 3)
        for k in range(3):
            col[..., k] = np.clip(img.astype(np.float32) * wts[k] + rng.integers(0, 20), 0, 255).astype(np.uint8)
        gray = O.cvt_gray(col, rgb)
        ex.mbRGB = rgb  # (the mirror passes it on with the channel count of every call)
        kc, dc = ex(col)
        rkc, rdc = O.Extractor(nf, sf, nlev, ini, mn)(gray)
        if len(kc) != len(rkc) or not all(np.array_equal(kc[n_], rkc[n_]) for n_ in ("x", "y", "octave", "response")) \
                or (len(rkc) and (dc == rdc).all(1).mean() < 0.999):
            what.append("colour input (%d channels, rgb %d)" % (ch, rgb))
        k1, d1 = ex(img)
        if not (np.array_equal(k1, kps) and np.array_equal(d1, desc)):
            what.append("first image again")
        if what:
            bad += 1
            print("MISMATCH (handle reused) case %d %s: %s" % (case, tag, "; ".join(what)), flush=True)
    # every eighth case: a device-resident batch of 64 .. 150 frames of a small crop (two staggered halves on
    # two streams; over 128 frames another dispatch order), a few frames of it against the oracle
    if case % 8 == 0:
        import torch
        r2, c2 = min(rows, int(rng.integers(120, 260))), min(cols, int(rng.integers(160, 420)))
        crop = np.ascontiguousarray(img[:r2, :c2])
        nb = int(rng.integers(64, 151))
        shifts = [0] + [int(rng.integers(1, 30)) for _ in range(nb - 2)] + [0]
        big = np.stack([np.roll(crop, sft, axis=1) for sft in shifts])
        try:
            ex2 = ORBextractor(nf, sf, nlev, ini, mn)
            ex2.set_opencv_compat(*compat)
            cap = ex2.max_keypoints(r2, c2)  # (refused if the crop's last level gets too small)
        except OrbHipError as e:
            continue
        dev = torch.device("cuda", 0)
        d_img = torch.from_numpy(big).to(dev)
        d_k = torch.zeros((nb, cap, 6), dtype=torch.float32, device=dev)
        d_d = torch.zeros((nb, cap, 32), dtype=torch.uint8, device=dev)
        d_n = torch.zeros(nb, dtype=torch.int32, device=dev)
        ex2.extract_batch_device(d_img.data_ptr(), nb, r2, c2, d_k.data_ptr(), d_d.data_ptr(), d_n.data_ptr(), cap,
                                 stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        from orb_slam2_amd._lib import KP_DTYPE
        hk = d_k.cpu().numpy().view(np.uint8).reshape(nb, cap, 24).copy().view(KP_DTYPE)[..., 0]
        hd, hn = d_d.cpu().numpy(), d_n.cpu().numpy()
        ref2 = O.Extractor(nf, sf, nlev, ini, mn)
        what = []
        for f in sorted({0, nb // 2 - 1, nb // 2, nb - 1, int(rng.integers(0, nb))}):
            rk2, rd2 = ref2(big[f], max_kp=cap + 64)
            n = int(hn[f])
            if n != len(rk2):
                what.append("frame %d: %d vs %d keypoints" % (f, n, len(rk2)))
                continue
            what += ["frame %d: %s" % (f, n_) for n_ in ("x", "y", "octave", "response", "size") if not np.array_equal(hk[f][:n][n_], rk2[n_])]
            if n and (hd[f][:n] == rd2).all(1).mean() < 0.999:
                what.append("frame %d: descriptors" % f)
        if not what:
            print("case %d: device batch of %d frames %dx%d ok" % (case, nb, r2, c2), flush=True)
        if what:
            bad += 1
            print("MISMATCH (device batch of %d, %dx%d) case %d %s: %s" % (nb, r2, c2, case, tag, "; ".join(what[:6])), flush=True)
O.set_opencv_compat(4, 5)
print("done: %d cases, %d mismatches" % (n_cases, bad))
sys.exit(1 if bad else 0)
