"""Shared builders for the parity tests: synthetic frames, poses and map points."""
import numpy as np


def kp_struct(x, y, octave, angle):
    from orb_slam2_amd._lib import KP_DTYPE
    k = np.zeros(len(x), KP_DTYPE)
    k["x"], k["y"], k["octave"], k["angle"] = x, y, octave, angle
    k["size"], k["response"] = 31, 50
    return k


def random_frame_arrays(rng, n, rows, cols, nlevels=8):
    x = rng.uniform(19, cols - 20, n).astype(np.float32)
    y = rng.uniform(19, rows - 20, n).astype(np.float32)
    octv = rng.integers(0, nlevels, n).astype(np.int32)
    ang = rng.uniform(0, 360, n).astype(np.float32)
    desc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    return x, y, octv, ang, desc


def flip_bits(rng, desc, max_flips):
    out = desc.copy()
    for r in range(len(out)):
        for b in rng.integers(0, 256, rng.integers(0, max_flips + 1)):
            out[r, b >> 3] ^= np.uint8(1 << (b & 7))
    return out


def scale_tables(nlevels=8, f=1.2):
    sf = np.ones(nlevels, np.float32)
    for i in range(1, nlevels):
        sf[i] = np.float32(np.float64(sf[i - 1]) * np.float64(np.float32(f)))
    return sf, (sf * sf).astype(np.float32)


def projection_scene(rng, n_last, n_cur, rows, cols, frac_mp=0.8, conflict_clusters=0):
    """A last frame whose map points project near current-frame keypoints, with some
    descriptors close (planted matches).  Returns the dicts the oracle wrappers take plus
    the pose/intrinsics."""
    sf, _ = scale_tables()
    fx = fy = np.float32(700.0)
    cx, cy = np.float32(cols / 2), np.float32(rows / 2)
    cx_, cy_, coct, cang, cdesc = random_frame_arrays(rng, n_cur, rows, cols)
    if conflict_clusters:
        # crowd keypoints so that several queries compete for the same candidates
        for c in range(conflict_clusters):
            m = rng.integers(0, n_cur, 12)
            cx_[m] = cx_[m[0]] + rng.uniform(-3, 3, 12).astype(np.float32)
            cy_[m] = cy_[m[0]] + rng.uniform(-3, 3, 12).astype(np.float32)
            coct[m] = coct[m[0]]
            cdesc[m] = flip_bits(rng, np.repeat(cdesc[m[:1]], 12, 0), 6)
    Tcw = np.eye(4, dtype=np.float32)
    Tcw[0, 3] = np.float32(0.05)
    Tcw[2, 3] = np.float32(0.02)
    src = rng.integers(0, n_cur, n_last)
    depth = rng.uniform(4, 30, n_last).astype(np.float32)
    # back-project the chosen current keypoint (plus jitter) to a world point
    u = cx_[src] + rng.uniform(-4, 4, n_last).astype(np.float32)
    v = cy_[src] + rng.uniform(-4, 4, n_last).astype(np.float32)
    Xc = np.stack([(u - cx) / fx * depth, (v - cy) / fy * depth, depth], 1).astype(np.float32)
    Xw = (Xc - Tcw[:3, 3][None]).astype(np.float32)
    has = rng.uniform(0, 1, n_last) < frac_mp
    obs = np.where(has, rng.integers(0, 4, n_last), -1).astype(np.int32)
    behind = rng.uniform(0, 1, n_last) < 0.03
    Xw[behind, 2] = -5.0
    ldesc = flip_bits(rng, cdesc[src], 60)
    loct = np.clip(coct[src] + rng.integers(-1, 2, n_last), 0, 7).astype(np.int32)
    lang = ((cang[src] + rng.normal(0, 8, n_last)) % 360).astype(np.float32)
    far = rng.uniform(0, 1, n_last) < 0.2
    lang[far] = rng.uniform(0, 360, far.sum()).astype(np.float32)
    cobs = np.where(rng.uniform(0, 1, n_cur) < 0.1, rng.integers(0, 3, n_cur), -1).astype(np.int32)
    last = dict(mp_obs=obs, xw=Xw, mp_desc=ldesc, octave=loct, angle=lang)
    cur = dict(x=cx_, y=cy_, octave=coct, angle=cang, desc=cdesc, obs=cobs)
    return dict(Tcw=Tcw, K=(fx, fy, cx, cy), bounds=(0.0, float(cols), 0.0, float(rows)), sf=sf, last=last, cur=cur)


def triangulation_scene(rng, n1, n2, rows=480, cols=752, nnodes=60):
    """Two keyframes related by a pure x-translation (epipolar lines are image rows) with feature
    vectors over `nnodes` vocabulary nodes.  Returns the flat arrays the oracle takes and the
    pieces the KeyFrame mirror is built from."""
    sf, s2 = scale_tables()
    x2, y2, o2, a2, d2 = random_frame_arrays(rng, n2, rows, cols)
    src = rng.integers(0, n2, n1)
    x1 = (x2[src] + rng.uniform(5, 60, n1)).astype(np.float32)
    y1 = (y2[src] + rng.normal(0, 0.7, n1)).astype(np.float32)
    o1 = o2[src].copy()
    a1 = ((a2[src] + rng.normal(0, 6, n1)) % 360).astype(np.float32)
    d1 = flip_bits(rng, d2[src], 45)
    K = (np.float32(500), np.float32(500), np.float32(376), np.float32(240))
    F12 = np.array([[0, 0, 0], [0, 0, -1], [0, 1, 0]], np.float32) * np.float32(0.002)
    node_of2 = rng.integers(0, nnodes, n2)
    node_of1 = np.where(rng.uniform(0, 1, n1) < 0.8, node_of2[src], rng.integers(0, nnodes + 5, n1))
    fv1 = {int(k) * 3: [int(i) for i in np.nonzero(node_of1 == k)[0]] for k in np.unique(node_of1)}
    fv2 = {int(k) * 3: [int(i) for i in np.nonzero(node_of2 == k)[0]] for k in np.unique(node_of2)}
    has1 = rng.uniform(0, 1, n1) < 0.3
    has2 = rng.uniform(0, 1, n2) < 0.3
    T1 = np.eye(4, dtype=np.float32)
    T2 = np.eye(4, dtype=np.float32); T2[0, 3] = 0.5; T2[2, 3] = 0.3
    return dict(sf=sf, s2=s2, K=K, F12=F12, bounds=(0, cols, 0, rows), T1=T1, T2=T2, fv1=fv1, fv2=fv2,
                k1=dict(x=x1, y=y1, octave=o1, angle=a1, desc=d1, has_mp=has1.astype(np.uint8)),
                k2=dict(x=x2, y=y2, octave=o2, angle=a2, desc=d2, has_mp=has2.astype(np.uint8)))


def flat_feature_vector(fv):
    """{node: [feature indices]} (ordered) -> sorted node ids + CSR item lists."""
    nodes = np.array(list(fv.keys()), np.int32)
    start, items = [0], []
    for v in fv.values():
        items.extend(v)
        start.append(len(items))
    return dict(node=nodes, start=np.array(start, np.int32), items=np.array(items, np.int32))
