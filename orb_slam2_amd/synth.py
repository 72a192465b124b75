"""Seeded synthetic inputs of the benchmark shapes (images, sequences, descriptors).

Integer-only numpy so that the same seed gives the same bytes everywhere.
"""
import numpy as np


def synth_image(rows, cols, seed=0, n_shapes=None):
    """8-bit grayscale test image: smooth background + filled rectangles + pixel noise.

    The rectangles guarantee FAST corners at several scales; the background and the
    noise keep flat regions from being exactly constant.
    """
    rng = np.random.default_rng(seed)
    gh, gw = rows // 32 + 2, cols // 32 + 2
    coarse = rng.integers(60, 196, size=(gh, gw), dtype=np.int32)
    # integer bilinear upsample of the coarse grid (32-px lattice)
    ys, xs = np.arange(rows), np.arange(cols)
    y0, fy = ys // 32, ys % 32
    x0, fx = xs // 32, xs % 32
    top = coarse[y0][:, x0] * (32 - fx)[None, :] + coarse[y0][:, x0 + 1] * fx[None, :]
    bot = coarse[y0 + 1][:, x0] * (32 - fx)[None, :] + coarse[y0 + 1][:, x0 + 1] * fx[None, :]
    img = (top * (32 - fy)[:, None] + bot * fy[:, None]) >> 10
    if n_shapes is None:
        n_shapes = max(64, rows * cols // 400)
    ry = rng.integers(0, rows, size=n_shapes)
    rx = rng.integers(0, cols, size=n_shapes)
    rh = rng.integers(3, 40, size=n_shapes)
    rw = rng.integers(3, 40, size=n_shapes)
    rv = rng.integers(0, 256, size=n_shapes)
    for i in range(n_shapes):
        img[ry[i]:ry[i] + rh[i], rx[i]:rx[i] + rw[i]] = rv[i]
    img = img + rng.integers(-4, 5, size=(rows, cols))
    return np.clip(img, 0, 255).astype(np.uint8)


def synth_sequence(rows, cols, n_frames, seed=0, step=2):
    """n_frames views of one scene, each shifted `step` px to the left of the previous
    one (a camera translating to the right), plus fresh +-2 pixel noise per frame."""
    scene = synth_image(rows, cols + step * n_frames, seed)
    rng = np.random.default_rng(seed + 7919)
    out = np.empty((n_frames, rows, cols), np.uint8)
    for k in range(n_frames):
        view = scene[:, k * step:k * step + cols].astype(np.int32)
        view = view + rng.integers(-2, 3, size=(rows, cols))
        out[k] = np.clip(view, 0, 255).astype(np.uint8)
    return out


def synth_descriptors(n_db, n_q, seed=1, max_flips=40):
    """Random 256-bit database; half of the queries are database rows with 0..max_flips
    random bit flips (planted matches), half are fresh random rows."""
    rng = np.random.default_rng(seed)
    db = rng.integers(0, 256, size=(n_db, 32), dtype=np.uint8)
    rq = np.random.default_rng(seed + 1)
    q = rq.integers(0, 256, size=(n_q, 32), dtype=np.uint8)
    n_plant = n_q // 2
    src = rq.integers(0, n_db, size=n_plant)
    planted = db[src].copy()
    nflip = rq.integers(0, max_flips + 1, size=n_plant)
    bits = rq.integers(0, 256, size=(n_plant, max_flips))
    for k in range(max_flips):
        m = nflip > k
        b = bits[m, k]
        planted[m, b >> 3] ^= (1 << (b & 7)).astype(np.uint8)
    q[:n_plant] = planted
    return db, q


class SequenceScene:
    """Camera and map points that go with synth_sequence(rows, cols, n, step=2): a pinhole
    camera translating +x by `tstep` per frame in front of a fronto-parallel scene at depth
    Z, so that the image content moves 2 px per frame.  Frame p's map points are its own
    keypoints back-projected to depth Z; projected with the pose of frame p + 1 they land
    2 px to the left, which is what SearchByProjection(Cur = p + 1, Last = p) looks for.

    All arithmetic is numpy float32 on the host, so that every consumer (benchmark, tests,
    CPU baseline) builds bit-identical inputs from the same keypoints."""

    def __init__(self, rows, cols):
        self.rows, self.cols = int(rows), int(cols)
        self.fx = self.fy = np.float32(700.0)
        self.cx, self.cy = np.float32(cols / 2.0), np.float32(rows / 2.0)
        self.Z = np.float32(10.0)
        self.tstep = np.float32(2.0 * 10.0 / 700.0)
        self.bounds = (0.0, float(cols), 0.0, float(rows))
        self.K = (self.fx, self.fy, self.cx, self.cy)

    def map_points(self, kx, ky, frame):
        """World positions (n, 3) float32 of the keypoints (kx, ky) of frame `frame`."""
        kx, ky = np.asarray(kx, np.float32), np.asarray(ky, np.float32)
        cam = np.float32(frame) * self.tstep
        x = (kx - self.cx) / self.fx * self.Z + cam
        y = (ky - self.cy) / self.fy * self.Z
        return np.stack([x, y, np.full_like(x, self.Z)], axis=-1).astype(np.float32)

    def tcw(self, frame):
        """Pose (4x4 float32, world -> camera) of frame `frame`."""
        T = np.eye(4, dtype=np.float32)
        T[0, 3] = -(np.float32(frame) * self.tstep)
        return T
