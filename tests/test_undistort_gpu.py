"""Frame::UndistortKeyPoints on the GPU against the oracle: bit-exact floats (same order of
double operations, no contraction), host-buffer and device-resident forms."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle
from orb_slam2_amd import _lib
from test_undistort_cpu import EUROC_D, EUROC_K, TUM1_D, TUM1_K

pytestmark = pytest.mark.gpu


def _cam(K, d):
    dd = np.zeros(5, np.float32)
    dd[:len(d)] = d
    return _lib.Camera(*K, *dd.tolist())


@pytest.mark.parametrize("K,d,rows,cols", [(EUROC_K, EUROC_D, 480, 752), (TUM1_K, TUM1_D, 480, 640),
                                            ((700.0, 705.0, 600.0, 180.0), (0.9, -3.0, 0.01, 0.02, 5.0), 376, 1241)])
def test_points_bit_exact(K, d, rows, cols):
    rng = np.random.default_rng(0)
    n = 20000
    xy = np.stack([rng.uniform(-20, cols + 20, n), rng.uniform(-20, rows + 20, n)], 1).astype(np.float32)
    xy[:4] = [[0, 0], [cols, 0], [0, rows], [cols, rows]]
    out = np.empty_like(xy)
    _lib.check(_lib.lib().orb_undistort_points(0, n, _lib.ptr(xy), _cam(K, d), _lib.ptr(out)))
    want = oracle.undistort_points(xy, K, d)
    assert np.array_equal(out.view(np.uint32), want.view(np.uint32))
    assert np.abs(out - xy).max() > 1.0, "scene too poor: the distortion moves nothing"


def test_extractor_output_undistorted_in_place_on_device():
    torch = pytest.importorskip("torch")
    from orb_slam2_amd import Frame, ORBextractor
    from orb_slam2_amd.synth import synth_sequence
    rows, cols, n = 480, 752, 3
    imgs = synth_sequence(rows, cols, n, seed=2)
    ex = ORBextractor(1200, 1.2, 8, 20, 7)
    cap = ex.max_keypoints(rows, cols)
    d = torch.from_numpy(imgs).cuda()
    kps = torch.zeros((n, cap, 6), dtype=torch.float32, device="cuda")
    desc = torch.zeros((n, cap, 32), dtype=torch.uint8, device="cuda")
    nout = torch.zeros(n, dtype=torch.int32, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    ex.extract_batch_device(d.data_ptr(), n, rows, cols, kps.data_ptr(), desc.data_ptr(), nout.data_ptr(), cap, stream=s)
    raw = kps.clone()
    un = torch.zeros_like(kps)
    L = _lib.lib()
    _lib.check(L.orb_undistort_keypoints_device(0, n, cap, C.c_void_p(kps.data_ptr()), C.c_void_p(nout.data_ptr()),
                                                _cam(EUROC_K, EUROC_D), C.c_void_p(un.data_ptr()), C.c_void_p(s)))
    # in place, and the k1 == 0 copy
    _lib.check(L.orb_undistort_keypoints_device(0, n, cap, C.c_void_p(kps.data_ptr()), C.c_void_p(nout.data_ptr()),
                                                _cam(EUROC_K, EUROC_D), C.c_void_p(kps.data_ptr()), C.c_void_p(s)))
    same = torch.zeros_like(kps)
    _lib.check(L.orb_undistort_keypoints_device(0, n, cap, C.c_void_p(raw.data_ptr()), C.c_void_p(nout.data_ptr()),
                                                _cam(EUROC_K, (0, 0.5, 0, 0)), C.c_void_p(same.data_ptr()), C.c_void_p(s)))
    torch.cuda.synchronize()
    raw_h, un_h, inplace_h, same_h = raw.cpu().numpy(), un.cpu().numpy(), kps.cpu().numpy(), same.cpu().numpy()
    for k in range(n):
        m = int(nout[k])
        assert m > 500
        want = oracle.undistort_points(raw_h[k, :m, :2], EUROC_K, EUROC_D)
        assert np.array_equal(un_h[k, :m, :2].view(np.uint32), want.view(np.uint32))
        assert np.array_equal(un_h[k, :m, 2:].view(np.uint32), raw_h[k, :m, 2:].view(np.uint32))  # size, angle, response, octave
        assert np.array_equal(inplace_h[k, :m].view(np.uint32), un_h[k, :m].view(np.uint32))
        assert np.array_equal(same_h[k, :m].view(np.uint32), raw_h[k, :m].view(np.uint32))
        assert not un_h[k, m:].any()  # rows beyond the count are not written
    # the host mirror: bounds and mvKeysUn of a frame built from raw keypoints
    k0, d0 = ex(imgs[0])
    fr = Frame.from_raw(k0, d0, ex.GetScaleFactors(), ex.GetScaleSigmaSquares(), EUROC_K, EUROC_D, rows, cols)
    assert (fr.mnMinX, fr.mnMaxX, fr.mnMinY, fr.mnMaxY) == tuple(float(np.float32(v)) for v in oracle.image_bounds(rows, cols, EUROC_K, EUROC_D))
    want = oracle.undistort_points(np.stack([k0["x"], k0["y"]], 1), EUROC_K, EUROC_D)
    assert np.array_equal(fr.mvKeysUn["x"], want[:, 0]) and np.array_equal(fr.mvKeysUn["y"], want[:, 1])
    assert np.array_equal(fr.mvKeys["x"], k0["x"]) and np.array_equal(fr.mvKeysUn["angle"], k0["angle"])
