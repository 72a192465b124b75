"""cv::undistortPoints as the reference's Frame uses it: oracle known answers, and the library's
host-side image bounds against the oracle (no GPU needed)."""
import numpy as np

from oracle import oracle
from orb_slam2_amd import _lib

EUROC_K = (458.654, 457.296, 367.215, 248.375)
EUROC_D = (-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05)
TUM1_K = (517.306408, 516.469215, 318.643040, 255.313989)
TUM1_D = (0.262383, -0.953104, -0.005358, 0.002628, 1.163314)


def _distort(xn, yn, d):
    """The forward model the iteration inverts (normalised coordinates)."""
    k1, k2, p1, p2, k3 = (list(d) + [0.0])[:5]
    r2 = xn * xn + yn * yn
    rad = 1 + k1 * r2 + k2 * r2 * r2 + k3 * r2 ** 3
    return (xn * rad + 2 * p1 * xn * yn + p2 * (r2 + 2 * xn * xn),
            yn * rad + p1 * (r2 + 2 * yn * yn) + 2 * p2 * xn * yn)


def test_principal_point_is_fixed_and_centre_inverts_forward_model():
    for K, d in ((EUROC_K, EUROC_D), (TUM1_K, TUM1_D)):
        fx, fy, cx, cy = K
        out = oracle.undistort_points([[cx, cy]], K, d)
        assert np.allclose(out, [[cx, cy]], atol=1e-4)
        # near the centre five iterations have converged: distorting the result gives the input back
        rng = np.random.default_rng(1)
        un = rng.uniform(-0.15, 0.15, (200, 2))
        xd, yd = _distort(un[:, 0], un[:, 1], d)
        pix = np.stack([xd * fx + cx, yd * fy + cy], 1).astype(np.float32)
        got = oracle.undistort_points(pix, K, d)
        want = np.stack([un[:, 0] * fx + cx, un[:, 1] * fy + cy], 1)
        assert np.abs(got - want).max() < 2e-2


def test_barrel_distortion_moves_corners_outwards():
    b = oracle.image_bounds(480, 752, EUROC_K, EUROC_D)
    assert b[0] < -50 and b[1] > 800 and b[2] < -30 and b[3] > 510
    assert oracle.image_bounds(480, 752, EUROC_K, (0, 0, 0, 0)) == (0.0, 752.0, 0.0, 480.0)


def test_library_image_bounds_equal_oracle():
    L = _lib.lib()
    for K, d, rows, cols in ((EUROC_K, EUROC_D, 480, 752), (TUM1_K, TUM1_D, 480, 640),
                             (EUROC_K, (0.0, 0.1, 0, 0), 480, 752), ((700, 700, 600, 180), (-0.1, 0.02, 1e-3, -1e-3, 0.003), 376, 1241)):
        dd = np.zeros(5, np.float32)
        dd[:len(d)] = d
        b = _lib.FrameBounds()
        _lib.check(L.orb_compute_image_bounds(rows, cols, _lib.Camera(*K, *dd.tolist()), b))
        want = oracle.image_bounds(rows, cols, K, d)
        assert (b.min_x, b.max_x, b.min_y, b.max_y) == tuple(np.float32(v) for v in want), (K, d)
    assert L.orb_compute_image_bounds(10, 10, _lib.Camera(0, 1, 0, 0, 0.1, 0, 0, 0, 0), b) == _lib.ERR_ARG


def test_regression_vector_euroc():
    """Bit patterns of the oracle's output for the EuRoC camera, recorded when the oracle was
    written (pins the oracle against later edits; the reference itself holds no vectors)."""
    pts = np.array([[0, 0], [752, 0], [0, 480], [752, 480], [100, 100], [600.5, 20.25]], np.float32)
    want = np.array([[3272068015, 3266844893], [1147134072, 3266953218], [3271921288, 1141703142],
                     [1147117037, 1141728102], [1113228145, 1117093491], [1143072551, 3251584121]], np.uint32)
    assert np.array_equal(oracle.undistort_points(pts, EUROC_K, EUROC_D).view(np.uint32), want)
