"""GPU parity of the extractor against the CPU oracle and the golden vectors.

Bars: pyramid pixels, keypoint x / y / octave and FAST responses
bit-exact; angles within 1e-4 rad; at least 99.9 % of descriptors bit-identical.
"""
import os
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ANGLE_TOL_DEG = 1e-4 * 180.0 / np.pi
DESC_MIN_IDENTICAL = 0.999


def check_against(kps, desc, rk, rd):
    assert len(kps) == len(rk)
    for f in ("x", "y", "octave", "response", "size"):
        assert np.array_equal(kps[f], rk[f]), f
    if len(kps):
        da = np.abs(kps["angle"].astype(np.float64) - rk["angle"].astype(np.float64))
        assert np.minimum(da, 360 - da).max() <= ANGLE_TOL_DEG
        assert (desc == rd).all(1).mean() >= DESC_MIN_IDENTICAL


def check_stages(ex, ref, nlevels, frame=0):
    for l in range(nlevels):
        assert np.array_equal(ex.pyramid_level(l, frame, padded=True), ref.pyramid_level(l)), "pyramid %d" % l
        assert np.array_equal(ex.fast_candidates(l, frame), ref.candidates(l)), "FAST %d" % l
        rk = ref.level_keypoints(l)
        want = np.stack([rk["x"], rk["y"], rk["response"]], 1) if len(rk) else np.zeros((0, 3), np.float32)
        assert np.array_equal(ex.level_keypoints(l, frame), want), "quad-tree %d" % l
        rb = ref.blurred_level(l)
        if rb is not None:
            assert np.array_equal(ex.blurred_level(l, frame), rb), "blur %d" % l


@pytest.mark.parametrize("rows,cols,nf,seed", [
    (480, 640, 1000, 0),      # TUM shape, benchmark configuration 0
    (376, 1241, 2000, 0),     # KITTI shape, config 1
    (480, 752, 1200, 0),      # EuRoC left eye, config 2
    (480, 752, 1200, 1),      # EuRoC right eye
    (240, 320, 500, 5),
    (211, 333, 300, 2),       # odd sizes
    (97, 131, 1500, 3),       # more features wanted than corners exist: every candidate survives
    (480, 640, 8000, 4),      # 2 x 4000 features, the reference's initialisation extractor (src/Tracking.cc:182)
])
def test_extract_matches_oracle(oracle, rows, cols, nf, seed):
    from orb_slam2_amd import ORBextractor
    from orb_slam2_amd.synth import synth_image
    nlevels = 8 if min(rows, cols) >= 240 else 2
    img = synth_image(rows, cols, seed)
    ex = ORBextractor(nf, 1.2, nlevels, 20, 7)
    ref = oracle.Extractor(nf, 1.2, nlevels, 20, 7)
    kps, desc = ex(img)
    rk, rd = ref(img)
    assert len(rk) > 50
    if nf == 1500:
        assert len(rk) < 1500
    check_against(kps, desc, rk, rd)
    check_stages(ex, ref, nlevels)


@pytest.mark.parametrize("nf", [4000, 9000])
def test_extract_4k_frame(oracle, nf):
    """One 3840x2160 frame at 4000 features (benchmark configuration 3 shape), and at 9000: the quad-tree
    then works with its larger node table on levels that hold tens of thousands of candidates."""
    from orb_slam2_amd import ORBextractor
    from orb_slam2_amd.synth import synth_image
    img = synth_image(2160, 3840, 7, n_shapes=20000)
    ex = ORBextractor(nf, 1.2, 8, 20, 7)
    ref = oracle.Extractor(nf, 1.2, 8, 20, 7)
    kps, desc = ex(img)
    rk, rd = ref(img)
    assert len(rk) >= nf
    check_against(kps, desc, rk, rd)
    for l in (0, 3, 7):
        assert np.array_equal(ex.fast_candidates(l), ref.candidates(l))


def test_extract_4k_batch_properties():
    """One rank's share of the 64-frame 3840x2160 batch (8 frames, device resident): every frame
    of the batch equals the same image extracted alone (which the test above checks against the
    oracle), whatever its place in the batch."""
    torch = pytest.importorskip("torch")
    from orb_slam2_amd import ORBextractor
    from orb_slam2_amd._lib import KP_DTYPE
    from orb_slam2_amd.synth import synth_image
    rows, cols, n = 2160, 3840, 8
    base = [synth_image(rows, cols, 7, n_shapes=20000), synth_image(rows, cols, 8, n_shapes=12000)]
    order = [0, 1, 1, 0, 0, 0, 1, 0]
    ex = ORBextractor(4000, 1.2, 8, 20, 7)
    singles = [ex(im) for im in base]
    batch = np.stack([base[o] for o in order])
    cap = ex.max_keypoints(rows, cols)
    d = torch.from_numpy(batch).cuda()
    kps = torch.zeros((n, cap, 6), dtype=torch.float32, device="cuda")
    desc = torch.zeros((n, cap, 32), dtype=torch.uint8, device="cuda")
    nout = torch.zeros(n, dtype=torch.int32, device="cuda")
    ex.extract_batch_device(d.data_ptr(), n, rows, cols, kps.data_ptr(), desc.data_ptr(), nout.data_ptr(), cap,
                            stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    hk, hd, hn = kps.cpu().numpy(), desc.cpu().numpy(), nout.cpu().numpy()
    for f in range(n):
        sk, sd = singles[order[f]]
        assert hn[f] == len(sk) >= 4000
        got = np.ascontiguousarray(hk[f, :hn[f]]).view(KP_DTYPE).reshape(-1)
        assert np.array_equal(got, sk), f
        assert np.array_equal(hd[f, :hn[f]], sd), f


@pytest.mark.parametrize("name", ["small_240x320", "tum_480x640", "kitti_376x1241", "euroc_480x752"])
def test_extract_matches_golden(name):
    from orb_slam2_amd import ORBextractor
    from orb_slam2_amd.synth import synth_image
    g = np.load(os.path.join(GOLD, name + ".npz"))
    rows, cols, seed, nf, nl = (int(v) for v in g["params"])
    img = synth_image(rows, cols, seed)
    assert zlib.crc32(img.tobytes()) == int(g["image_crc"])
    ex = ORBextractor(nf, 1.2, nl, 20, 7)
    kps, desc = ex(img)
    assert len(kps) == len(g["x"])
    for f in ("x", "y", "octave", "response", "size"):
        assert np.array_equal(kps[f], g[f]), f
    assert np.abs(kps["angle"] - g["angle"]).max() <= ANGLE_TOL_DEG
    assert (desc == g["desc"]).all(1).mean() >= DESC_MIN_IDENTICAL
    for l in range(nl):
        assert zlib.crc32(ex.pyramid_level(l, padded=True).tobytes()) == int(g["pyr_crc_%d" % l])
        assert zlib.crc32(ex.fast_candidates(l).tobytes()) == int(g["cand_crc_%d" % l])
        if int(g["blur_crc_%d" % l]):
            assert zlib.crc32(ex.blurred_level(l).tobytes()) == int(g["blur_crc_%d" % l])


def test_low_texture_uses_min_threshold(oracle):
    """Faint corners (contrast 12) are below iniThFAST = 20 and only found by the
    minThFAST = 7 retry of cells that came up empty."""
    from orb_slam2_amd import ORBextractor
    rng = np.random.default_rng(0)
    img = np.full((240, 320), 100, np.int32)
    for _ in range(150):
        y, x = rng.integers(0, 230), rng.integers(0, 310)
        img[y:y + rng.integers(4, 12), x:x + rng.integers(4, 12)] = 112
    img[:, :160] += rng.integers(-1, 2, (240, 160))
    # a few strong corners on the right half: those cells must NOT get the retry
    for _ in range(30):
        y, x = rng.integers(0, 230), rng.integers(170, 310)
        img[y:y + 6, x:x + 6] = 200
    img = np.clip(img, 0, 255).astype(np.uint8)
    ex = ORBextractor(500, 1.2, 8, 20, 7)
    ref = oracle.Extractor(500, 1.2, 8, 20, 7)
    kps, desc = ex(img)
    rk, rd = ref(img)
    assert (rk["response"] < 19).sum() > 10 and (rk["response"] >= 20).sum() > 10
    check_against(kps, desc, rk, rd)
    check_stages(ex, ref, 8)


@pytest.mark.parametrize("kind", ["noise", "salt", "checker"])
def test_dense_corners_take_the_overflow_paths(oracle, kind):
    """Images on which nearly every pixel passes FAST's quick test and a large part are corners:
    the per-wavefront corner lists overflow (dense sweep of the score map), queues re-queue
    both polarities, levels hold far more candidates than the quad-tree keeps in registers.
    One frame alone and the same frame in a batch (other launch shapes), stage by stage."""
    from orb_slam2_amd import ORBextractor
    rng = np.random.default_rng(5)
    rows, cols = 240, 416
    if kind == "noise":
        img = rng.integers(0, 256, (rows, cols), dtype=np.uint8)
    elif kind == "salt":
        img = np.full((rows, cols), 128, np.uint8)
        m = rng.random((rows, cols)) < 0.08
        img[m] = rng.choice(np.array([0, 255], np.uint8), int(m.sum()))
    else:
        yy, xx = np.mgrid[0:rows, 0:cols]
        img = (((yy // 5 + xx // 5) & 1) * 200 + 20 + rng.integers(0, 9, (rows, cols))).astype(np.uint8)
    ex = ORBextractor(1500, 1.2, 8, 20, 7)
    ref = oracle.Extractor(1500, 1.2, 8, 20, 7)
    kps, desc = ex(img)
    rk, rd = ref(img)
    assert len(rk) > 500
    check_against(kps, desc, rk, rd)
    check_stages(ex, ref, 8)
    bk, bd = ex.extract_batch(np.repeat(img[None], 10, 0))
    for f in (0, 9):
        check_against(bk[f], bd[f], rk, rd)
    check_stages(ex, ref, 8, frame=9)


def test_constant_and_empty_images():
    from orb_slam2_amd import ORBextractor
    ex = ORBextractor(500, 1.2, 8, 20, 7)
    k, d = ex(np.full((240, 320), 50, np.uint8))
    assert len(k) == 0 and d.shape == (0, 32)
    k, d = ex(np.zeros((0, 0), np.uint8))
    assert len(k) == 0 and d.shape == (0, 32)


def test_unsupported_inputs_fail_loudly():
    from orb_slam2_amd import ORBextractor
    from orb_slam2_amd._lib import ERR_UNSUPPORTED, OrbHipError
    ex = ORBextractor(500, 1.2, 8, 20, 7)
    with pytest.raises(OrbHipError) as e:   # level 7 would be 17 px high: the reference divides by zero
        ex(np.zeros((60, 400), np.uint8))
    assert e.value.code == ERR_UNSUPPORTED
    with pytest.raises(OrbHipError):        # aspect below 0.5: the reference's quad-tree has no root
        ORBextractor(500, 1.2, 2, 20, 7)(np.zeros((900, 200), np.uint8))
    with pytest.raises(OrbHipError):
        ORBextractor(50000, 1.2, 8, 20, 7)(np.zeros((480, 640), np.uint8))


@pytest.mark.parametrize("nfeatures,nlevels,rows,cols", [(5000, 2, 600, 800), (2900, 1, 480, 752), (13000, 8, 720, 1280)])
def test_more_than_two_thousand_keypoints_on_one_level(oracle, nfeatures, nlevels, rows, cols):
    """Levels that want 2000 to 3000 keypoints take the quad-tree kernel with the largest node table
    (one workgroup per compute unit): a frame alone and a batch of 9, every stage against the oracle."""
    from orb_slam2_amd import ORBextractor
    from orb_slam2_amd.synth import synth_sequence
    seq = synth_sequence(rows, cols, 9, seed=41, step=3)
    ex = ORBextractor(nfeatures, 1.2, nlevels, 20, 7)
    ref = oracle.Extractor(nfeatures, 1.2, nlevels, 20, 7)
    assert ex.features_per_level()[0].max() > 2040
    kps, desc = ex(seq[0])
    rk, rd = ref(seq[0], max_kp=2 * nfeatures + 4096)
    assert len(rk) > 0.5 * nfeatures, "scene too poor"
    check_against(kps, desc, rk, rd)
    check_stages(ex, ref, nlevels)
    bk, bd = ex.extract_batch(seq)
    for f in (0, 8):
        rk, rd = ref(seq[f], max_kp=2 * nfeatures + 4096)
        check_against(bk[f], bd[f], rk, rd)
        check_stages(ex, ref, nlevels, frame=f)


def test_batch_equals_singles_and_strided_input(oracle):
    from orb_slam2_amd import ORBextractor
    from orb_slam2_amd.synth import synth_sequence
    seq = synth_sequence(240, 320, 5, seed=3)
    ex = ORBextractor(500, 1.2, 8, 20, 7)
    bk, bd = ex.extract_batch(seq)
    ref = oracle.Extractor(500, 1.2, 8, 20, 7)
    for f in range(5):
        rk, rd = ref(seq[f])
        check_against(bk[f], bd[f], rk, rd)
        if f in (0, 4):
            check_stages(ex, ref, 8, frame=f)
    # a view with a row stride larger than its width (a cv::Mat ROI)
    big = np.zeros((260, 400), np.uint8)
    big[10:250, 40:360] = seq[2]
    import ctypes as C
    from orb_slam2_amd._lib import KP_DTYPE, check, lib, ptr
    view = big[10:250, 40:360]
    cap = ex.max_keypoints(240, 320)
    kps = np.zeros(cap, KP_DTYPE); desc = np.zeros((cap, 32), np.uint8); n = C.c_int()
    check(lib().orb_extract(ex._h, C.c_void_p(view.ctypes.data), 240, 320, 400, cap, ptr(kps), ptr(desc), C.byref(n)))
    check_against(kps[:n.value], desc[:n.value], bk[2], bd[2])


def test_device_resident_path_and_reuse(oracle):
    """Device-pointer entry on torch's current stream; a second call with a smaller batch
    and then another image size reuses / rebuilds the plan."""
    import torch
    from orb_slam2_amd import ORBextractor
    from orb_slam2_amd._lib import KP_DTYPE
    from orb_slam2_amd.synth import synth_sequence
    dev = torch.device("cuda", 0)
    ex = ORBextractor(500, 1.2, 8, 20, 7)
    ref = oracle.Extractor(500, 1.2, 8, 20, 7)
    for (rows, cols, n) in [(240, 320, 4), (240, 320, 2), (250, 330, 3)]:
        seq = synth_sequence(rows, cols, n, seed=rows + n)
        d_img = torch.from_numpy(seq).to(dev)
        cap = ex.max_keypoints(rows, cols)
        d_k = torch.zeros((n, cap, 6), dtype=torch.float32, device=dev)
        d_d = torch.zeros((n, cap, 32), dtype=torch.uint8, device=dev)
        d_n = torch.zeros(n, dtype=torch.int32, device=dev)
        ex.extract_batch_device(d_img.data_ptr(), n, rows, cols, d_k.data_ptr(), d_d.data_ptr(), d_n.data_ptr(), cap,
                                stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        cnt = d_n.cpu().numpy()
        for f in range(n):
            rk, rd = ref(seq[f])
            kp = d_k[f, :cnt[f]].cpu().numpy().copy().view(KP_DTYPE).reshape(-1)
            check_against(kp, d_d[f, :cnt[f]].cpu().numpy(), rk, rd)


@pytest.mark.parametrize("n", [3, 11])
def test_device_path_with_short_output_arrays(n):
    """max_kp below the number of keypoints found: the first max_kp keypoints and descriptors are
    written, nothing beyond them, and d_n_out still holds the number found (small launches take
    four keypoints per wavefront, larger ones eight)."""
    import torch
    from orb_slam2_amd import ORBextractor
    from orb_slam2_amd._lib import KP_DTYPE
    from orb_slam2_amd.synth import synth_sequence
    dev = torch.device("cuda", 0)
    rows, cols = 240, 320
    seq = synth_sequence(rows, cols, n, seed=50 + n)
    ex = ORBextractor(600, 1.2, 8, 20, 7)
    fk, fd = ex.extract_batch(seq)
    d_img = torch.from_numpy(seq).to(dev)
    st = torch.cuda.current_stream().cuda_stream
    for cap in (601, 333, 8, 1):
        d_k = torch.full((n, cap + 4, 6), -7.0, dtype=torch.float32, device=dev)
        d_d = torch.full((n, cap + 4, 32), 0xAB, dtype=torch.uint8, device=dev)
        d_n = torch.zeros(n, dtype=torch.int32, device=dev)
        # arrays of cap + 4 rows per frame, of which the library may write cap
        ex.extract_batch_device(d_img.data_ptr(), n, rows, cols, d_k.data_ptr(), d_d.data_ptr(), d_n.data_ptr(), cap + 4,
                                stream=st)
        torch.cuda.synchronize()
        full_k, full_d = d_k.cpu().numpy().copy(), d_d.cpu().numpy().copy()
        d_k.fill_(-7.0); d_d.fill_(0xAB)
        ex2 = ORBextractor(600, 1.2, 8, 20, 7)
        d_k2 = torch.full((n, cap, 6), -7.0, dtype=torch.float32, device=dev)
        d_d2 = torch.full((n, cap, 32), 0xAB, dtype=torch.uint8, device=dev)
        ex2.extract_batch_device(d_img.data_ptr(), n, rows, cols, d_k2.data_ptr(), d_d2.data_ptr(), d_n.data_ptr(), cap,
                                 stream=st)
        torch.cuda.synchronize()
        cnt = d_n.cpu().numpy()
        for f in range(n):
            assert cnt[f] == len(fk[f]) and cnt[f] > cap - 300
            m = min(cap, cnt[f])
            got = d_k2[f].cpu().numpy().copy().view(KP_DTYPE).reshape(-1)
            assert np.array_equal(got[:m], fk[f][:m]) and np.array_equal(d_d2[f, :m].cpu().numpy(), fd[f][:m])
            assert (d_k2[f, m:].cpu().numpy() == -7.0).all() and (d_d2[f, m:].cpu().numpy() == 0xAB).all()
            mm = min(cap + 4, cnt[f])
            assert (full_k[f, mm:] == -7.0).all() and (full_d[f, mm:] == 0xAB).all()


def test_full_size_properties_batch():
    """Properties that hold at the benchmark's batch size without the oracle: every frame of
    a batch of identical images gives identical output; output is level-major; keypoints
    respect the 19-px border; descriptors of a frame are those of the same image alone."""
    from orb_slam2_amd import ORBextractor
    from orb_slam2_amd.synth import synth_image
    img = synth_image(376, 1241, 42)
    ex = ORBextractor(2000, 1.2, 8, 20, 7)
    k1, d1 = ex(img)
    bk, bd = ex.extract_batch(np.repeat(img[None], 24, 0))
    sf = ex.GetScaleFactors()
    for f in range(24):
        assert np.array_equal(bk[f], k1) and np.array_equal(bd[f], d1)
    assert (np.diff(k1["octave"]) >= 0).all()
    lx, ly = k1["x"] / sf[k1["octave"]], k1["y"] / sf[k1["octave"]]
    assert (lx > 18.99).all() and (ly > 18.99).all()
    q, _ = ex.features_per_level()
    counts = np.bincount(k1["octave"], minlength=8)
    assert ((counts >= q) & (counts <= q + 3)).all()


def test_large_batch_equals_small_batches(oracle):
    """Batches over 128 frames take another dispatch order in the blur and the threaded packing
    of the host path: 139 distinct frames (not a multiple of 8) in one call equal the same
    frames in calls of 5, and a few of them are checked against the oracle."""
    from orb_slam2_amd import ORBextractor
    from orb_slam2_amd.synth import synth_sequence
    n = 139
    seq = synth_sequence(168, 224, n, seed=17, step=1)
    ex = ORBextractor(300, 1.2, 6, 20, 7)
    bk, bd = ex.extract_batch(seq)
    ex2 = ORBextractor(300, 1.2, 6, 20, 7)
    for f0 in range(0, n, 5):
        sk, sd = ex2.extract_batch(seq[f0:f0 + 5])
        for i in range(len(sk)):
            assert np.array_equal(bk[f0 + i], sk[i]) and np.array_equal(bd[f0 + i], sd[i]), f0 + i
    ref = oracle.Extractor(300, 1.2, 6, 20, 7)
    for f in (0, 127, 128, 136, 138):
        rk, rd = ref(seq[f])
        check_against(bk[f], bd[f], rk, rd)


@pytest.mark.parametrize("compat", [(4, 5), (4, 2)])
def test_blur_fused_into_descriptors(oracle, compat):
    """The descriptor kernel blurs each keypoint's patch in on-chip memory by default; with the
    levels blurred to memory first (set_blur_fusion(False), the reference's order) keypoints and
    descriptors are bit-identical, for both tap sets (the one summing to 257 saturates), single
    frames and small batches (four keypoints per wavefront) and batches (eight), at image borders and with bright
    patches that make the saturation matter."""
    from orb_slam2_amd import ORBextractor
    from orb_slam2_amd.synth import synth_sequence
    rng = np.random.default_rng(5)
    for rows, cols, nf, nlev in [(376, 1241, 2000, 8), (240, 320, 700, 6), (155, 411, 500, 4)]:
        seq = synth_sequence(rows, cols, 12, seed=rows, step=2).copy()
        seq[3:6] = np.clip(seq[3:6].astype(np.int32) * 3, 0, 255).astype(np.uint8)   # saturated regions
        seq[6] = rng.integers(0, 256, (rows, cols), dtype=np.uint8)                  # corners everywhere, up to the border
        seq[7] = np.where(rng.random((rows, cols)) < 0.5, 255, 250).astype(np.uint8)
        a, b = ORBextractor(nf, 1.2, nlev, 20, 7), ORBextractor(nf, 1.2, nlev, 20, 7)
        a.set_opencv_compat(*compat)
        b.set_opencv_compat(*compat)
        b.set_blur_fusion(False)
        ak, ad = a.extract_batch(seq)
        bk, bd = b.extract_batch(seq)
        total = 0
        for f in range(len(seq)):
            assert np.array_equal(ak[f], bk[f]) and np.array_equal(ad[f], bd[f]), (rows, f)
            total += len(ak[f])
        assert total > 1000
        for f in (0, 4, 6):
            k1, d1 = a(seq[f])
            assert np.array_equal(k1, bk[f]) and np.array_equal(d1, bd[f]), (rows, f)
            k2, d2 = b(seq[f])  # single frame with the levels blurred to memory
            assert np.array_equal(k2, bk[f]) and np.array_equal(d2, bd[f]), (rows, f)
        # the blurred level is made on demand after a fused call and equals the one written up front
        a.extract_batch(seq)
        b.extract_batch(seq)
        for l in (0, nlev - 1):
            assert np.array_equal(a.blurred_level(l, 7), b.blurred_level(l, 7))


def test_host_batch_in_pieces(oracle):
    """Host-buffer batches of 64 frames and more go through in pieces, each with an upload, its
    kernels and a download of its own: strided host images (rows and frames with padding), the
    64-frame threshold, the stages of frames in different pieces, a capacity too small for the
    results, and a smaller batch on the same handle afterwards."""
    from orb_slam2_amd import ORBextractor
    from orb_slam2_amd._lib import ERR_CAPACITY, KP_DTYPE, lib
    from orb_slam2_amd.synth import synth_sequence
    L = lib()
    rows, cols, n = 150, 203, 71
    seq = synth_sequence(rows, cols, n, seed=23, step=1)
    stride, fstride = cols + 13, (rows + 3) * (cols + 13) + 5
    buf = np.full(n * fstride, 77, np.uint8)
    for k in range(n):
        for r in range(rows):
            o = k * fstride + r * stride
            buf[o:o + cols] = seq[k, r]
    ex = ORBextractor(250, 1.2, 5, 20, 7)
    ref = oracle.Extractor(250, 1.2, 5, 20, 7)
    cap = ex.max_keypoints(rows, cols)
    for m in (n, 64, 63):
        kps = np.zeros((m, cap), KP_DTYPE)
        desc = np.zeros((m, cap, 32), np.uint8)
        nout = np.full(m, -1, np.int32)
        rc = L.orb_extract_batch(ex._h, m, buf.ctypes.data, rows, cols, stride, fstride, cap,
                                 kps.ctypes.data, desc.ctypes.data, nout.ctypes.data)
        assert rc == 0
        ex._frames = m
        ck, cd = ORBextractor(250, 1.2, 5, 20, 7).extract_batch(seq[:m][::-1])  # contiguous, another order
        for f in range(m):
            assert np.array_equal(kps[f, :nout[f]], ck[m - 1 - f]) and np.array_equal(desc[f, :nout[f]], cd[m - 1 - f]), (m, f)
        for f in (0, m // 2, m - 1):
            rk, rd = ref(seq[f])
            check_against(kps[f, :nout[f]], desc[f, :nout[f]], rk, rd)
            check_stages(ex, ref, 5, frame=f)
    # a capacity below what a frame produces: the counts still come back, with the error code
    small = int(nout.min()) - 1
    kps = np.zeros((n, small), KP_DTYPE)
    desc = np.zeros((n, small, 32), np.uint8)
    n2 = np.zeros(n, np.int32)
    rc = L.orb_extract_batch(ex._h, n, buf.ctypes.data, rows, cols, stride, fstride, small,
                             kps.ctypes.data, desc.ctypes.data, n2.ctypes.data)
    assert rc == ERR_CAPACITY and (n2 > small).all()
    bk, _ = ex.extract_batch(seq)
    assert [len(k) for k in bk] == list(n2)


@pytest.mark.parametrize("rows,cols,nlevels,scale", [(376, 1241, 8, 1.2), (480, 640, 8, 1.2), (211, 333, 5, 1.3),
                                                    (1080, 1920, 8, 1.2)])
def test_batch_stages_match_oracle(oracle, rows, cols, nlevels, scale):
    """Batches of more than 8 frames take their own launches (taller resize tiles, FAST in two
    launches by view height, four keypoints per wavefront): every stage of every level of a
    few frames of such a batch against the oracle, pixel for pixel."""
    from orb_slam2_amd import ORBextractor
    from orb_slam2_amd.synth import synth_sequence
    n = 11
    seq = synth_sequence(rows, cols, n, seed=23, step=3)
    ex = ORBextractor(1000, scale, nlevels, 20, 7)
    bk, bd = ex.extract_batch(seq)
    ref = oracle.Extractor(1000, scale, nlevels, 20, 7)
    for f in (0, 6, 10):
        rk, rd = ref(seq[f])
        check_against(bk[f], bd[f], rk, rd)
        check_stages(ex, ref, nlevels, frame=f)


@pytest.mark.parametrize("scale", [1.2, 1.1])
def test_batch_levels_0_and_1_for_every_width_around_a_tile_column_boundary(oracle, scale):
    """In batches levels 0 and 1 come out of one pass over the image, a workgroup per tile of level 1
    that also writes the 16-byte pieces of level 0 its column of tiles owns.  Where a tile column
    begins decides how many of its pieces take the aligned path - none, one, several - so every
    width of a 48-pixel range around the start of the third tile column (level-1 slab column 2048)
    is run: the padded bytes of levels 0 and 1 of the first and the last frame of a batch of 9
    against the oracle's."""
    from orb_slam2_amd import ORBextractor
    from orb_slam2_amd.synth import synth_image
    rows = 100
    w0 = int((2048 - 32) * scale)
    bad = []
    for cols in range(w0 - 8, w0 + 40):
        img = synth_image(rows, cols, seed=cols)
        batch = np.stack([img] + [np.roll(img, 3 * k, axis=1) for k in range(1, 8)] + [img])
        ex = ORBextractor(500, scale, 2, 20, 7)
        ex.extract_batch(batch)
        ref = oracle.Extractor(500, scale, 2, 20, 7)
        ref(img)
        for f in (0, 8):
            for l in (0, 1):
                if not np.array_equal(ex.pyramid_level(l, f, padded=True), ref.pyramid_level(l)):
                    bad.append((cols, f, l))
    assert not bad, bad


@pytest.mark.parametrize("n", [1, 9])
def test_wide_level_at_a_large_scale_factor_takes_the_generic_resize(oracle, n):
    """Scale factor 1.6 on an 1800-px-wide image: a 1024-column tile of level 1 would need more
    source columns than the tiled resize kernel stages, so that level is made by the generic
    kernel (a thread per four output pixels, taps from the tables) while the narrower levels above
    it stay tiled; one frame (host call) and a batch, every stage against the oracle."""
    from orb_slam2_amd import ORBextractor
    from orb_slam2_amd.synth import synth_sequence
    rows, cols, nlevels, scale = 300, 1800, 4, 1.6
    seq = synth_sequence(rows, cols, n, seed=31, step=3)
    ex = ORBextractor(1200, scale, nlevels, 20, 7)
    ref = oracle.Extractor(1200, scale, nlevels, 20, 7)
    if n == 1:
        kps, desc = ex(seq[0])
        bk, bd = [kps], [desc]
    else:
        bk, bd = ex.extract_batch(seq)
    for f in sorted({0, n - 1}):
        rk, rd = ref(seq[f])
        assert len(rk) > 300
        check_against(bk[f], bd[f], rk, rd)
        check_stages(ex, ref, nlevels, frame=f)


def test_level_read_back_into_a_strided_array(oracle):
    """orb_get_pyramid_level / orb_get_blurred_level write rows of the level's width into an array whose
    rows are longer (a cv::Mat with a step, as mvImagePyramid may be): the level lands in the left part
    of every row, the bytes to the right stay as they were, padded and unpadded, several frames."""
    import ctypes as C
    from orb_slam2_amd import ORBextractor
    from orb_slam2_amd._lib import check, lib, ptr
    from orb_slam2_amd.synth import synth_sequence
    seq = synth_sequence(211, 333, 3, seed=9, step=2)
    ex = ORBextractor(500, 1.2, 4, 20, 7)
    ex.extract_batch(seq)
    ref = oracle.Extractor(500, 1.2, 4, 20, 7)
    ref(seq[2])
    L = lib()
    for level in (0, 3):
        for padded in (0, 1):
            w, h = C.c_int(), C.c_int()
            check(L.orb_get_pyramid_level(ex._h, 2, level, padded, None, 0, C.byref(w), C.byref(h)))
            stride = w.value + 37
            out = np.full((h.value, stride), 0xA5, np.uint8)
            check(L.orb_get_pyramid_level(ex._h, 2, level, padded, ptr(out), stride, None, None))
            want = ref.pyramid_level(level)
            if not padded:
                want = want[19:-19, 19:-19]
            assert np.array_equal(out[:, :w.value], want), (level, padded)
            assert np.all(out[:, w.value:] == 0xA5), (level, padded)
        w, h = C.c_int(), C.c_int()
        check(L.orb_get_blurred_level(ex._h, 2, level, None, 0, C.byref(w), C.byref(h)))
        out = np.full((h.value, w.value + 5), 0x5A, np.uint8)
        check(L.orb_get_blurred_level(ex._h, 2, level, ptr(out), w.value + 5, None, None))
        assert np.array_equal(out[:, :w.value], ex.blurred_level(level, 2)), level
        assert np.all(out[:, w.value:] == 0x5A), level
    # a destination narrower than the level is refused
    assert L.orb_get_pyramid_level(ex._h, 0, 0, 0, ptr(out), 10, None, None) < 0


def _colour_image(rows, cols, ch, seed):
    """A colour image whose grey value has corners: the synthetic grey scene modulated per
    channel, plus noise, so that the three weights all matter."""
    from orb_slam2_amd.synth import synth_image
    rng = np.random.default_rng(seed)
    base = synth_image(rows, cols, seed).astype(np.int32)
    img = np.stack([np.clip(base + rng.integers(-40, 41, base.shape), 0, 255) for _ in range(3)], 2)
    if ch == 4:
        img = np.concatenate([img, rng.integers(0, 256, (rows, cols, 1))], 2)
    return np.ascontiguousarray(img.astype(np.uint8))


@pytest.mark.parametrize("rows,cols,ch,rgb", [(376, 1241, 3, True), (376, 1241, 3, False), (480, 640, 4, True),
                                              (480, 752, 4, False), (211, 333, 3, True)])
def test_colour_input_is_converted_in_level0(oracle, rows, cols, ch, rgb):
    """Colour frames through the extractor equal cvtColor + grey extraction, stage by stage."""
    from orb_slam2_amd import ORBextractor
    nlevels = 8 if min(rows, cols) >= 240 else 2
    img = _colour_image(rows, cols, ch, 5)
    gray = oracle.cvt_gray(img, rgb)
    assert len(np.unique(gray)) > 100
    ex = ORBextractor(1000, 1.2, nlevels, 20, 7)
    ex.mbRGB = rgb
    kps, desc = ex(img)
    assert np.array_equal(ex.pyramid_level(0), gray)
    ref = oracle.Extractor(1000, 1.2, nlevels, 20, 7)
    rk, rd = ref(gray)
    assert len(rk) > 50, "scene too poor"
    check_against(kps, desc, rk, rd)
    check_stages(ex, ref, nlevels)
    # the handle goes back to grey input without being rebuilt
    k2, d2 = ex(gray)
    assert np.array_equal(k2, kps) and np.array_equal(d2, desc)


def test_colour_device_path_strided_batch(oracle):
    """Device-resident colour frames with row padding and an odd (4-byte misaligned) start."""
    torch = pytest.importorskip("torch")
    from orb_slam2_amd import ORBextractor
    rows, cols, n = 240, 321, 3
    stride, fstride = cols * 3 + 5, rows * (cols * 3 + 5) + 7
    buf = np.zeros(1 + n * fstride, np.uint8)
    frames = [_colour_image(rows, cols, 3, 20 + k) for k in range(n)]
    for k in range(n):
        v = buf[1 + k * fstride:1 + k * fstride + rows * stride].reshape(rows, stride)
        v[:, :cols * 3] = frames[k].reshape(rows, cols * 3)
    ex = ORBextractor(800, 1.2, 8, 20, 7)
    ex.set_input_format(3, False)
    cap = ex.max_keypoints(rows, cols)
    d = torch.from_numpy(buf).cuda()
    kps = torch.zeros((n, cap, 6), dtype=torch.float32, device="cuda")
    desc = torch.zeros((n, cap, 32), dtype=torch.uint8, device="cuda")
    nout = torch.zeros(n, dtype=torch.int32, device="cuda")
    ex.extract_batch_device(d.data_ptr() + 1, n, rows, cols, kps.data_ptr(), desc.data_ptr(), nout.data_ptr(), cap,
                            stream=torch.cuda.current_stream().cuda_stream, stride=stride, frame_stride=fstride)
    torch.cuda.synchronize()
    ref = oracle.Extractor(800, 1.2, 8, 20, 7)
    for k in range(n):
        gray = oracle.cvt_gray(frames[k], False)
        assert np.array_equal(ex.pyramid_level(0, k), gray)
        rk, rd = ref(gray)
        m = int(nout[k])
        assert m == len(rk) and m > 50
        assert np.array_equal(kps[k, :m, 0].cpu().numpy(), rk["x"]) and np.array_equal(kps[k, :m, 1].cpu().numpy(), rk["y"])
        assert (desc[k, :m].cpu().numpy() == rd).all(1).mean() >= DESC_MIN_IDENTICAL
    # a stride that cannot hold a colour row is refused
    from orb_slam2_amd._lib import OrbHipError
    with pytest.raises(OrbHipError):
        ex.extract_batch_device(d.data_ptr(), n, rows, cols, kps.data_ptr(), desc.data_ptr(), nout.data_ptr(), cap,
                                stride=cols, frame_stride=fstride)


@pytest.mark.parametrize("major,minor", [(4, 2), (3, 4), (4, 8)])
def test_opencv_release_switch(oracle, major, minor):
    """orb_extractor_set_opencv_compat: the blur taps of OpenCV 3.4 / 4.0 - 4.4 (every tap rounded,
    sum 257, saturated results) and the 14-bit grey weights of 3.x, against the oracle set to the
    same release; 4.5+ is the default and gives the default's results."""
    from orb_slam2_amd import ORBextractor
    from orb_slam2_amd._lib import OrbHipError
    rows, cols = 300, 420
    img = _colour_image(rows, cols, 3, 9)
    img[40:80, 50:120] = 255  # a saturated patch: with taps summing to 257 the blur must clip at 255
    ex = ORBextractor(900, 1.2, 8, 20, 7)
    ex.set_opencv_compat(major, minor)
    try:
        oracle.set_opencv_compat(major, minor)
        taps = list(oracle.gaussian_taps())
        assert taps == ([18, 34, 48, 56, 48, 34, 18] if (major, minor) >= (4, 5) else [18, 34, 49, 55, 49, 34, 18])
        gray = oracle.cvt_gray(img, True)
        ref = oracle.Extractor(900, 1.2, 8, 20, 7)
        rk, rd = ref(gray)
        kps, desc = ex(img)
        assert np.array_equal(ex.pyramid_level(0), gray)
        assert len(rk) > 50
        check_against(kps, desc, rk, rd)
        check_stages(ex, ref, 8)
        assert ex.blurred_level(0)[45:75, 60:110].max() == 255
    finally:
        oracle.set_opencv_compat(4, 5)
    if (major, minor) < (4, 5):  # and the releases really differ on this image
        d = ORBextractor(900, 1.2, 8, 20, 7)
        d(img)
        assert not np.array_equal(d.blurred_level(0), ex.blurred_level(0))
    with pytest.raises(OrbHipError):
        ex.set_opencv_compat(2, 4)


def test_profiled_small_host_calls_upload_their_images(oracle):
    """With stage timing on, a host-buffer call of a few frames takes the plain launch sequence
    instead of the captured graph; it must still upload its own images: two different images one
    after the other on one handle, each equal to the unprofiled result, and stage times come back."""
    import ctypes as C
    from orb_slam2_amd import ORBextractor, _lib
    from orb_slam2_amd.synth import synth_image
    L = _lib.lib()
    imgs = [synth_image(240, 320, seed=s) for s in (31, 32)]
    plain = ORBextractor(500, 1.2, 6, 20, 7)
    want = [plain(im) for im in imgs]
    assert not np.array_equal(want[0][0], want[1][0])
    ex = ORBextractor(500, 1.2, 6, 20, 7)
    _lib.check(L.orb_extractor_set_profiling(ex._h, 1))
    for im, (wk, wd) in zip(imgs, want):   # the very first call on the handle is profiled too
        k, d = ex(im)
        assert np.array_equal(k, wk) and np.array_equal(d, wd)
    ms = np.zeros(5, np.float32)
    assert _lib.check(L.orb_extractor_stage_times(ex._h, ms.ctypes.data)) == 5 and ms.sum() > 0
    _lib.check(L.orb_extractor_set_profiling(ex._h, 0))
    # back on the graph path, and profiled again after it: still each call's own image
    for im, (wk, wd) in list(zip(imgs, want))[::-1]:
        k, d = ex(im)
        assert np.array_equal(k, wk) and np.array_equal(d, wd)
    _lib.check(L.orb_extractor_set_profiling(ex._h, 1))
    k, d = ex(imgs[1])
    assert np.array_equal(k, want[1][0]) and np.array_equal(d, want[1][1])


def test_blurred_level_follows_the_last_call_on_a_reused_handle(oracle):
    """The blurred levels are made on demand after a call with the blur fused into the descriptor
    kernel.  A second image of the same size on the same handle replays the captured graph; the
    blurred level asked for afterwards must be the second image's, not the first one's."""
    from orb_slam2_amd import ORBextractor
    from orb_slam2_amd.synth import synth_image
    a, b = synth_image(240, 320, seed=41), synth_image(240, 320, seed=42)
    ex = ORBextractor(500, 1.2, 6, 20, 7)
    ref = oracle.Extractor(500, 1.2, 6, 20, 7)
    ex(a)
    ref(a)
    assert np.array_equal(ex.blurred_level(0), ref.blurred_level(0))
    for img in (b, a, b):            # replays of the same graph
        ex(img)
        ref(img)
        for l in (0, 3):
            rb = ref.blurred_level(l)
            if rb is not None:
                assert np.array_equal(ex.blurred_level(l), rb), l
    # a batch of three after single frames (another graph), frame 2 asked for
    ex.extract_batch(np.stack([a, b, a]))
    ref(a)
    assert np.array_equal(ex.blurred_level(0, frame=2), ref.blurred_level(0))
    ex.extract_batch(np.stack([a, a, b]))
    ref(b)
    assert np.array_equal(ex.blurred_level(0, frame=2), ref.blurred_level(0))


def test_handle_changes_size_between_host_and_device_calls(oracle):
    """One handle through image sizes and both call forms in turn (every change reallocates the
    handle's buffers on its own stream and drops the captured graph): results stay those of a fresh
    handle."""
    import torch
    from orb_slam2_amd import ORBextractor
    from orb_slam2_amd._lib import KP_DTYPE
    from orb_slam2_amd.synth import synth_image
    dev = torch.device("cuda", 0)
    ex = ORBextractor(400, 1.2, 5, 20, 7)
    side = torch.cuda.Stream(device=dev)
    for it, (rows, cols) in enumerate([(240, 320), (200, 260), (240, 320), (300, 500), (200, 260)]):
        img = synth_image(rows, cols, seed=50 + it)
        wk, wd = ORBextractor(400, 1.2, 5, 20, 7)(img)
        if it % 2 == 0:
            k, d = ex(img)
        else:
            cap = ex.max_keypoints(rows, cols)
            d_img = torch.from_numpy(img).to(dev)
            d_k = torch.zeros((1, cap, 6), dtype=torch.float32, device=dev)
            d_d = torch.zeros((1, cap, 32), dtype=torch.uint8, device=dev)
            d_n = torch.zeros(1, dtype=torch.int32, device=dev)
            torch.cuda.synchronize()
            ex.set_input_format(1)
            ex.extract_batch_device(d_img.data_ptr(), 1, rows, cols, d_k.data_ptr(), d_d.data_ptr(), d_n.data_ptr(), cap,
                                    stream=side.cuda_stream)
            lvl = ex.pyramid_level(1)        # waits for the call on the side stream, not for the device
            assert lvl.shape[0] > 0
            n = int(d_n.item())
            k = d_k[0, :n].cpu().numpy().view(np.uint8).reshape(n, 24).copy().view(KP_DTYPE)[:, 0]
            d = d_d[0, :n].cpu().numpy()
        assert np.array_equal(k, wk) and np.array_equal(d, wd), it


def test_device_batch_in_two_staggered_halves(oracle):
    """orb_extract_batch_device runs batches of 64 frames and more as two halves on two streams, the
    second one stage behind the first, joined on the caller's stream.  Keypoints and descriptors must
    be those of the unsplit call, for an odd frame count, on a side stream, called twice in a row
    without a wait in between; frames of both halves are read back stage by stage."""
    import torch
    from orb_slam2_amd import ORBextractor, _lib
    from orb_slam2_amd._lib import KP_DTYPE
    from orb_slam2_amd.synth import synth_sequence
    L = _lib.lib()
    dev = torch.device("cuda", 0)
    rows, cols, n = 200, 333, 67
    seq = synth_sequence(rows, cols, n, seed=8, step=3)
    d_img = torch.from_numpy(seq).to(dev)
    outs = []
    side = torch.cuda.Stream(device=dev)
    for split in (0, 1):
        ex = ORBextractor(600, 1.2, 5, 20, 7)
        _lib.check(L.orb_extractor_set_batch_split(ex._h, split))
        cap = ex.max_keypoints(rows, cols)
        k = torch.zeros((n, cap, 6), dtype=torch.float32, device=dev)
        d = torch.zeros((n, cap, 32), dtype=torch.uint8, device=dev)
        c = torch.full((n,), -1, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        for _ in range(2):
            ex.extract_batch_device(d_img.data_ptr(), n, rows, cols, k.data_ptr(), d.data_ptr(), c.data_ptr(), cap,
                                    stream=side.cuda_stream)
        with torch.cuda.stream(side):
            total = c.sum()          # ordered after the call on the caller's stream only
        side.synchronize()
        outs.append((k.cpu().numpy(), d.cpu().numpy(), c.cpu().numpy(), int(total.item())))
        if split:
            ref = oracle.Extractor(600, 1.2, 5, 20, 7)
            for f in (0, n // 2 - 1, n // 2, n - 1):
                rk, rd = ref(seq[f])
                m = int(outs[-1][2][f])
                got = np.ascontiguousarray(outs[-1][0][f, :m]).view(KP_DTYPE).reshape(-1)
                check_against(got, outs[-1][1][f, :m], rk, rd)
                check_stages(ex, ref, 5, frame=f)
    a, b = outs
    assert a[3] == b[3] == int(a[2].sum()) and (a[2] > 50).all()
    assert np.array_equal(a[2], b[2])
    for f in range(n):
        m = int(a[2][f])
        assert np.array_equal(a[0][f, :m], b[0][f, :m]) and np.array_equal(a[1][f, :m], b[1][f, :m]), f


def test_device_batch_from_strided_grey_images(oracle):
    """Batches of more than 8 one-channel frames make pyramid levels 0 and 1 in one pass over the
    input image: rows and frames with padding between them (odd byte counts, so every row has
    another alignment), an odd image width and a width that is a multiple of 16; every stage of
    the first and last frame against the oracle."""
    import torch
    from orb_slam2_amd import ORBextractor
    from orb_slam2_amd._lib import KP_DTYPE
    from orb_slam2_amd.synth import synth_sequence
    dev = torch.device("cuda", 0)
    for rows, cols, pad_r, pad_f in [(201, 333, 7, 13), (160, 320, 1, 0), (190, 1105, 3, 5)]:
        n = 12
        seq = synth_sequence(rows, cols, n, seed=cols, step=3)
        stride, fstride = cols + pad_r, rows * (cols + pad_r) + pad_f
        buf = np.full(n * fstride + 64, 255, np.uint8)
        for f in range(n):
            view = buf[f * fstride:f * fstride + rows * stride].reshape(rows, stride)
            view[:, :cols] = seq[f]
        d_buf = torch.from_numpy(buf).to(dev)
        ex = ORBextractor(500, 1.2, 4, 20, 7)
        cap = ex.max_keypoints(rows, cols)
        k = torch.zeros((n, cap, 6), dtype=torch.float32, device=dev)
        d = torch.zeros((n, cap, 32), dtype=torch.uint8, device=dev)
        c = torch.zeros(n, dtype=torch.int32, device=dev)
        ex.set_input_format(1)
        ex.extract_batch_device(d_buf.data_ptr(), n, rows, cols, k.data_ptr(), d.data_ptr(), c.data_ptr(), cap,
                                stream=torch.cuda.current_stream().cuda_stream, stride=stride, frame_stride=fstride)
        torch.cuda.synchronize()
        ref = oracle.Extractor(500, 1.2, 4, 20, 7)
        for f in (0, n - 1):
            rk, rd = ref(seq[f])
            m = int(c[f])
            got = np.ascontiguousarray(k[f, :m].cpu().numpy()).view(KP_DTYPE).reshape(-1)
            check_against(got, d[f, :m].cpu().numpy(), rk, rd)
            check_stages(ex, ref, 4, frame=f)
