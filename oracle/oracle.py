"""ctypes binding of the CPU oracle (oracle/orb_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liborb_oracle.so")


class KeyPoint(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("size", C.c_float),
                ("angle", C.c_float), ("response", C.c_float), ("octave", C.c_int)]


KP_DTYPE = np.dtype([("x", "f4"), ("y", "f4"), ("size", "f4"), ("angle", "f4"),
                     ("response", "f4"), ("octave", "i4")])


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def _load():
    if not os.path.exists(_SO):
        build()
    return C.CDLL(_SO)


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _load()
        L = _lib
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int, C.c_int]
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_get_tables.argtypes = [C.c_void_p] + [C.c_void_p] * 6
        L.orc_extract.restype = C.c_int
        L.orc_extract.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_level_size.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_get_pyramid_level.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.orc_get_blurred_level.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.orc_get_candidates.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.orc_get_level_keypoints.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.orc_fast_atan2.restype = C.c_float
        L.orc_fast_atan2.argtypes = [C.c_float, C.c_float]
        L.orc_descriptor_distance.argtypes = [C.c_void_p, C.c_void_p]
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _u8(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


# ---- primitives -------------------------------------------------------------
def resize_linear(src, dw, dh):
    src = _u8(src)
    sh, sw = src.shape
    dst = np.empty((dh, dw), np.uint8)
    lib().orc_resize_linear_8u(_p(src), sw, sh, sw, _p(dst), dw, dh, dw)
    return dst


def copy_make_border_101(src, pad):
    src = _u8(src)
    sh, sw = src.shape
    dst = np.empty((sh + 2 * pad, sw + 2 * pad), np.uint8)
    lib().orc_copy_make_border_101(_p(src), sw, sh, sw, _p(dst), sw + 2 * pad, pad)
    return dst


def fast9_16(img, threshold):
    img = _u8(img)
    h, w = img.shape
    out = np.empty((w * h, 3), np.int32)
    n = lib().orc_fast9_16(_p(img), w, h, w, int(threshold), w * h, _p(out))
    return out[:n].copy()


def set_opencv_compat(major, minor):
    """Process-wide: the OpenCV release whose blur taps and grey weights are restated (default 4.5+)."""
    lib().orc_set_opencv_compat(C.c_int(int(major)), C.c_int(int(minor)))


def gaussian_taps():
    t = np.zeros(7, np.int32)
    lib().orc_gaussian_taps_7_s2(_p(t))
    return t


def gaussian_blur(img):
    img = _u8(img)
    h, w = img.shape
    dst = np.empty((h, w), np.uint8)
    lib().orc_gaussian_blur_7x7(_p(img), w, h, w, _p(dst), w)
    return dst


def fast_atan2(y, x):
    return float(lib().orc_fast_atan2(float(y), float(x)))


def distribute_octree(x, y, resp, min_x, max_x, min_y, max_y, n_target):
    x, y, resp = _f32(x), _f32(y), _f32(resp)
    out = np.empty(max(len(x), 1), np.int32)
    n = lib().orc_distribute_octree(_p(x), _p(y), _p(resp), len(x), int(min_x), int(max_x),
                                    int(min_y), int(max_y), int(n_target), _p(out))
    if n < 0:
        raise ValueError("reference behaviour undefined for this input (code %d)" % n)
    return out[:n].copy()


def descriptor_distance(a, b):
    a, b = _u8(a), _u8(b)
    return int(lib().orc_descriptor_distance(_p(a), _p(b)))


def three_maxima(sizes):
    sizes = _i32(sizes)
    i1, i2, i3 = C.c_int(), C.c_int(), C.c_int()
    lib().orc_three_maxima(_p(sizes), len(sizes), C.byref(i1), C.byref(i2), C.byref(i3))
    return i1.value, i2.value, i3.value


# ---- extractor ----------------------------------------------------------------
class Extractor:
    """Mirror of ORBextractor (reference include/ORBextractor.h:95,113) on the CPU."""

    def __init__(self, nfeatures, scale_factor, nlevels, ini_th, min_th):
        self.h = lib().orc_create(nfeatures, scale_factor, nlevels, ini_th, min_th)
        if not self.h:
            raise ValueError("bad extractor parameters")
        self.nlevels = nlevels
        self.nfeatures = nfeatures

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_destroy(self.h)
            self.h = None

    def tables(self):
        n = self.nlevels
        sf, inv, s2, is2 = (np.zeros(n, np.float32) for _ in range(4))
        quota = np.zeros(n, np.int32)
        umax = np.zeros(16, np.int32)
        lib().orc_get_tables(self.h, _p(sf), _p(inv), _p(s2), _p(is2), _p(quota), _p(umax))
        return dict(scale=sf, inv_scale=inv, sigma2=s2, inv_sigma2=is2, quota=quota, umax=umax)

    def __call__(self, img, max_kp=None):
        img = _u8(img)
        rows, cols = img.shape
        # (a level returns a few more keypoints than its quota, and at least one per root node of its
        # quad-tree, which on wide images with few features is more than the quota)
        cap = max_kp if max_kp is not None else self.nfeatures * 2 + 64 + 64 * self.nlevels + cols // 8
        kps = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = lib().orc_extract(self.h, _p(img), rows, cols, cols, cap, _p(kps), _p(desc))
        if n < 0:
            raise ValueError("reference behaviour undefined for this input (code %d)" % n)
        assert n <= cap
        return kps[:n].copy(), desc[:n].copy()

    def level_size(self, level):
        w, h = C.c_int(), C.c_int()
        if lib().orc_level_size(self.h, level, C.byref(w), C.byref(h)) != 0:
            raise ValueError("no such level")
        return w.value, h.value

    def pyramid_level(self, level):
        w, h = self.level_size(level)
        out = np.empty((h + 38, w + 38), np.uint8)
        lib().orc_get_pyramid_level(self.h, level, _p(out))
        return out

    def blurred_level(self, level):
        w, h = self.level_size(level)
        out = np.empty((h, w), np.uint8)
        if lib().orc_get_blurred_level(self.h, level, _p(out)) != 0:
            return None
        return out

    def candidates(self, level):
        w, h = self.level_size(level)
        out = np.empty((w * h // 4 + 16, 3), np.float32)
        n = lib().orc_get_candidates(self.h, level, len(out), _p(out))
        return out[:n].copy()

    def level_keypoints(self, level):
        out = np.zeros(self.nfeatures * 2 + 64, KP_DTYPE)
        n = lib().orc_get_level_keypoints(self.h, level, len(out), _p(out))
        return out[:n].copy()


# ---- matchers -------------------------------------------------------------------
def features_in_area(kx, ky, koct, bounds, x, y, r, min_level, max_level):
    kx, ky, koct = _f32(kx), _f32(ky), _i32(koct)
    out = np.empty(max(len(kx), 1), np.int32)
    n = lib().orc_features_in_area(len(kx), _p(kx), _p(ky), _p(koct),
                                   C.c_float(bounds[0]), C.c_float(bounds[1]), C.c_float(bounds[2]), C.c_float(bounds[3]),
                                   C.c_float(x), C.c_float(y), C.c_float(r), int(min_level), int(max_level), _p(out))
    return out[:n].copy()


def search_by_projection_frames(Tcw, K, bounds, scale_factors, last, cur, th, check_ori=True):
    """last: dict(mp_obs, xw, mp_desc, octave, angle); cur: dict(x, y, octave, angle, desc, obs).
    Returns (nmatches, cur_match)."""
    Tcw = _f32(Tcw).reshape(16)
    sf = _f32(scale_factors)
    l_obs, l_xw, l_desc = _i32(last["mp_obs"]), _f32(last["xw"]), _u8(last["mp_desc"])
    l_oct, l_ang = _i32(last["octave"]), _f32(last["angle"])
    c_x, c_y, c_oct, c_ang = _f32(cur["x"]), _f32(cur["y"]), _i32(cur["octave"]), _f32(cur["angle"])
    c_desc, c_obs = _u8(cur["desc"]), _i32(cur["obs"])
    match = np.empty(max(len(c_x), 1), np.int32)
    f = C.c_float
    n = lib().orc_search_by_projection_frames(
        _p(Tcw), f(K[0]), f(K[1]), f(K[2]), f(K[3]), f(bounds[0]), f(bounds[1]), f(bounds[2]), f(bounds[3]), _p(sf),
        len(l_obs), _p(l_obs), _p(l_xw), _p(l_desc), _p(l_oct), _p(l_ang),
        len(c_x), _p(c_x), _p(c_y), _p(c_oct), _p(c_ang), _p(c_desc), _p(c_obs), f(th), int(check_ori), _p(match))
    return n, match[:len(c_x)].copy()


def search_by_projection_mappoints(bounds, scale_factors, frame, mps, th, nnratio):
    """frame: dict(x, y, octave, desc, obs); mps: dict(track, level, viewcos, projx, projy, desc, obs)."""
    sf = _f32(scale_factors)
    f_x, f_y, f_oct, f_desc, f_obs = _f32(frame["x"]), _f32(frame["y"]), _i32(frame["octave"]), _u8(frame["desc"]), _i32(frame["obs"])
    m_tr, m_lv, m_vc = _u8(mps["track"]), _i32(mps["level"]), _f32(mps["viewcos"])
    m_px, m_py, m_desc, m_obs = _f32(mps["projx"]), _f32(mps["projy"]), _u8(mps["desc"]), _i32(mps["obs"])
    match = np.empty(max(len(f_x), 1), np.int32)
    f = C.c_float
    n = lib().orc_search_by_projection_mappoints(
        f(bounds[0]), f(bounds[1]), f(bounds[2]), f(bounds[3]), _p(sf),
        len(f_x), _p(f_x), _p(f_y), _p(f_oct), _p(f_desc), _p(f_obs),
        len(m_tr), _p(m_tr), _p(m_lv), _p(m_vc), _p(m_px), _p(m_py), _p(m_desc), _p(m_obs),
        f(th), f(nnratio), _p(match))
    return n, match[:len(f_x)].copy()


def match_windowed(bounds, train, queries, max_dist, check_ori, keyframe=False):
    """train: dict(x, y, octave, angle, desc, taken); queries: dict(active, u, v, r, minl, maxl, angle, desc).
    keyframe: the train side is a KeyFrame (queries subtract its int-truncated mnMinX / mnMinY)."""
    t_x, t_y, t_o, t_a = _f32(train["x"]), _f32(train["y"]), _i32(train["octave"]), _f32(train["angle"])
    t_d, t_t = _u8(train["desc"]), _u8(train["taken"])
    q_a, q_u, q_v, q_r = _u8(queries["active"]), _f32(queries["u"]), _f32(queries["v"]), _f32(queries["r"])
    q_lo, q_hi, q_ang, q_d = _i32(queries["minl"]), _i32(queries["maxl"]), _f32(queries["angle"]), _u8(queries["desc"])
    match = np.empty(max(len(t_x), 1), np.int32)
    f = C.c_float
    fn = lib().orc_match_windowed_keyframe if keyframe else lib().orc_match_windowed
    n = fn(f(bounds[0]), f(bounds[1]), f(bounds[2]), f(bounds[3]),
                                 len(t_x), _p(t_x), _p(t_y), _p(t_o), _p(t_a), _p(t_d), _p(t_t),
                                 len(q_a), _p(q_a), _p(q_u), _p(q_v), _p(q_r), _p(q_lo), _p(q_hi), _p(q_ang), _p(q_d),
                                 int(max_dist), int(check_ori), _p(match))
    return n, match[:len(t_x)].copy()


def search_for_initialization(bounds, f1, prev, f2, window, nnratio, check_ori=True):
    """f1: dict(octave, angle, desc); prev: (n1, 2) previous positions; f2: dict(x, y, octave, angle, desc)."""
    o1, a1, d1 = _i32(f1["octave"]), _f32(f1["angle"]), _u8(f1["desc"])
    prev = np.asarray(prev, np.float32).reshape(-1, 2)
    px, py = _f32(prev[:, 0]), _f32(prev[:, 1])
    x2, y2, o2, a2, d2 = _f32(f2["x"]), _f32(f2["y"]), _i32(f2["octave"]), _f32(f2["angle"]), _u8(f2["desc"])
    n1, n2 = len(o1), len(x2)
    m12 = np.full(max(n1, 1), -1, np.int32)
    f = C.c_float
    L = lib()
    L.orc_search_for_initialization.restype = C.c_int
    n = L.orc_search_for_initialization(f(bounds[0]), f(bounds[1]), f(bounds[2]), f(bounds[3]), C.c_int(n1), _p(o1), _p(a1), _p(d1),
                                        _p(px), _p(py), C.c_int(n2), _p(x2), _p(y2), _p(o2), _p(a2), _p(d2), f(window),
                                        f(nnratio), C.c_int(int(check_ori)), _p(m12))
    return n, m12[:n1].copy()


def window_best(bounds, train, queries, inv_level_sigma2=None, chi2=0.0, keyframe=False):
    """train: dict(x, y, octave, desc); queries: dict(active, u, v, r, level, desc).
    keyframe: the train side is a KeyFrame (queries subtract its int-truncated mnMinX / mnMinY).
    Returns (best_idx, best_dist); no candidate -> (-1, INT_MAX)."""
    t_x, t_y, t_o, t_d = _f32(train["x"]), _f32(train["y"]), _i32(train["octave"]), _u8(train["desc"])
    q_a, q_u, q_v, q_r = _u8(queries["active"]), _f32(queries["u"]), _f32(queries["v"]), _f32(queries["r"])
    q_l, q_d = _i32(queries["level"]), _u8(queries["desc"])
    nq = len(q_a)
    bi, bd = np.empty(max(nq, 1), np.int32), np.empty(max(nq, 1), np.int32)
    inv = _f32(inv_level_sigma2) if inv_level_sigma2 is not None else np.ones(16, np.float32)
    f = C.c_float
    L = lib()
    fn = L.orc_window_best_keyframe if keyframe else L.orc_window_best
    fn.restype = None
    fn(f(bounds[0]), f(bounds[1]), f(bounds[2]), f(bounds[3]), C.c_int(len(t_x)), _p(t_x), _p(t_y), _p(t_o),
                      _p(t_d), _p(inv), C.c_double(chi2), C.c_int(nq), _p(q_a), _p(q_u), _p(q_v), _p(q_r), _p(q_l), _p(q_d),
                      _p(bi), _p(bd))
    return bi[:nq].copy(), bd[:nq].copy()


def stereo_match(rows, left, right, scale_factors, min_d, max_d, th_dist=75):
    """left / right: dicts with x, y, octave, desc.  Returns (n, best_idx, best_dist)."""
    L = lib()
    nl, nr = len(left["x"]), len(right["x"])
    bi = np.full(max(nl, 1), -1, np.int32)
    bd = np.zeros(max(nl, 1), np.int32)
    lx, ly, lo, ld = _f32(left["x"]), _f32(left["y"]), _i32(left["octave"]), _u8(left["desc"])
    rx, ry, ro, rd = _f32(right["x"]), _f32(right["y"]), _i32(right["octave"]), _u8(right["desc"])
    sf = _f32(scale_factors)
    L.orc_stereo_match.restype = C.c_int
    n = L.orc_stereo_match(C.c_int(rows), C.c_int(nl), _p(lx), _p(ly), _p(lo), _p(ld), C.c_int(nr), _p(rx), _p(ry), _p(ro),
                           _p(rd), _p(sf), C.c_float(min_d), C.c_float(max_d), C.c_int(th_dist), _p(bi), _p(bd))
    return n, bi[:nl], bd[:nl]


def search_for_triangulation(kf1, kf2, fv1, fv2, F12, epipole, scale_factors2, level_sigma2_2, check_ori=False):
    """kf: dict(x, y, octave, angle, desc, has_mp); fv: dict(node, start, items)."""
    a = {}
    for tag, kf in (("1", kf1), ("2", kf2)):
        a["x" + tag], a["y" + tag] = _f32(kf["x"]), _f32(kf["y"])
        a["o" + tag], a["a" + tag] = _i32(kf["octave"]), _f32(kf["angle"])
        a["d" + tag], a["m" + tag] = _u8(kf["desc"]), _u8(kf["has_mp"])
    n1, n2 = len(a["x1"]), len(a["x2"])
    nd1, st1, it1 = _i32(fv1["node"]), _i32(fv1["start"]), _i32(fv1["items"])
    nd2, st2, it2 = _i32(fv2["node"]), _i32(fv2["start"]), _i32(fv2["items"])
    F12 = _f32(F12).reshape(9)
    sf2, ls2 = _f32(scale_factors2), _f32(level_sigma2_2)
    m12 = np.empty(max(n1, 1), np.int32)
    f = C.c_float
    n = lib().orc_search_for_triangulation(
        n1, _p(a["x1"]), _p(a["y1"]), _p(a["o1"]), _p(a["a1"]), _p(a["d1"]), _p(a["m1"]),
        n2, _p(a["x2"]), _p(a["y2"]), _p(a["o2"]), _p(a["a2"]), _p(a["d2"]), _p(a["m2"]),
        len(nd1), _p(nd1), _p(st1), _p(it1), len(nd2), _p(nd2), _p(st2), _p(it2),
        _p(F12), f(epipole[0]), f(epipole[1]), _p(sf2), _p(ls2), int(check_ori), _p(m12))
    return n, m12[:n1].copy()


def search_by_bow(s1, s2, fv1, fv2, nnratio, max_dist, check_ori=True):
    """s1 / s2: dict(angle, desc, ok); fv: dict(node, start, items).  Returns (n, matches12)."""
    a1, d1, k1 = _f32(s1["angle"]), _u8(s1["desc"]), _u8(s1["ok"])
    a2, d2, k2 = _f32(s2["angle"]), _u8(s2["desc"]), _u8(s2["ok"])
    n1, n2 = len(a1), len(a2)
    nd1, st1, it1 = _i32(fv1["node"]), _i32(fv1["start"]), _i32(fv1["items"])
    nd2, st2, it2 = _i32(fv2["node"]), _i32(fv2["start"]), _i32(fv2["items"])
    m12 = np.full(max(n1, 1), -1, np.int32)
    L = lib()
    L.orc_search_by_bow.restype = C.c_int
    n = L.orc_search_by_bow(C.c_int(n1), _p(a1), _p(d1), _p(k1), C.c_int(n2), _p(a2), _p(d2), _p(k2),
                            C.c_int(len(nd1)), _p(nd1), _p(st1), _p(it1), C.c_int(len(nd2)), _p(nd2), _p(st2), _p(it2),
                            C.c_float(nnratio), C.c_int(max_dist), C.c_int(int(check_ori)), _p(m12))
    return n, m12[:n1].copy()


def knn2_bruteforce(q, db, ratio):
    q, db = _u8(q), _u8(db)
    nq = len(q)
    best, second, idx = (np.empty(max(nq, 1), np.int32) for _ in range(3))
    acc = np.empty(max(nq, 1), np.uint8)
    lib().orc_knn2_bruteforce(_p(q), nq, _p(db), len(db), C.c_float(ratio), _p(best), _p(second), _p(idx), _p(acc))
    return best[:nq].copy(), second[:nq].copy(), idx[:nq].copy(), acc[:nq].copy()


def bow_transform(parent, is_leaf, node_desc, node_weight, levels, feats, levelsup):
    """Vocabulary in file-record form (parent id, leaf flag, descriptor, weight per node; node 0
    = root).  Returns (word_id, node_id, weight) per feature."""
    parent, leaf, nd = _i32(parent), _u8(is_leaf), _u8(node_desc)
    wt = np.ascontiguousarray(node_weight, np.float64)
    feats = _u8(feats)
    n = len(feats)
    word, node = np.empty(max(n, 1), np.int32), np.empty(max(n, 1), np.int32)
    w = np.empty(max(n, 1), np.float64)
    L = lib()
    L.orc_bow_transform.restype = C.c_int
    L.orc_bow_transform(C.c_int(len(parent)), _p(parent), _p(leaf), _p(nd), _p(wt), C.c_int(levels), C.c_int(n),
                        _p(feats), C.c_int(levelsup), _p(word), _p(node), _p(w))
    return word[:n].copy(), node[:n].copy(), w[:n].copy()


def bow_vector_l1(word_id, weight):
    """Returns (ids, values, keep): the L1-normalised BowVector in ascending word order and
    the per-feature 'not stopped' flags."""
    word_id = _i32(word_id)
    weight = np.ascontiguousarray(weight, np.float64)
    n = len(word_id)
    ids, vals, keep = np.empty(max(n, 1), np.int32), np.empty(max(n, 1), np.float64), np.empty(max(n, 1), np.uint8)
    L = lib()
    L.orc_bow_vector_l1.restype = C.c_int
    m = L.orc_bow_vector_l1(C.c_int(n), _p(word_id), _p(weight), _p(ids), _p(vals), _p(keep))
    return ids[:m].copy(), vals[:m].copy(), keep[:n].astype(bool)


def cvt_gray(img, rgb_order=True):
    """img: (rows, cols, 3|4) uint8 -> (rows, cols) grey, as cv::cvtColor computes it."""
    img = _u8(img)
    rows, cols, ch = img.shape
    out = np.empty((rows, cols), np.uint8)
    lib().orc_cvt_gray(_p(img), C.c_int(rows), C.c_int(cols), C.c_int(cols * ch), C.c_int(ch), C.c_int(int(rgb_order)), _p(out))
    return out


def distinctive_descriptor(desc):
    """-> (index of the least-median descriptor, that median)."""
    desc = _u8(desc).reshape(-1, 32)
    med = C.c_int()
    L = lib()
    L.orc_distinctive_descriptor.restype = C.c_int
    idx = L.orc_distinctive_descriptor(C.c_int(len(desc)), _p(desc), C.byref(med))
    return idx, med.value


def undistort_points(xy, K, dist):
    """xy: (n, 2) float32; K = (fx, fy, cx, cy); dist = (k1, k2, p1, p2[, k3])."""
    xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
    K4 = np.asarray(K, np.float32)
    d = np.zeros(5, np.float32)
    d[:len(dist)] = dist
    out = np.empty_like(xy)
    lib().orc_undistort_points(C.c_int(len(xy)), _p(xy), _p(K4), _p(d), _p(out))
    return out


def image_bounds(rows, cols, K, dist):
    """Frame::ComputeImageBounds (reference src/Frame.cc:560-596) -> (minX, maxX, minY, maxY)."""
    if len(dist) == 0 or np.float32(dist[0]) == 0:
        return (0.0, float(cols), 0.0, float(rows))
    c = undistort_points([[0, 0], [cols, 0], [0, rows], [cols, rows]], K, dist)
    return (float(min(c[0, 0], c[2, 0])), float(max(c[1, 0], c[3, 0])),
            float(min(c[0, 1], c[1, 1])), float(max(c[2, 1], c[3, 1])))
