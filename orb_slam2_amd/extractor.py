"""Host-side mirror of ORB_SLAM2::ORBextractor (reference include/ORBextractor.h:83-220)
over the C ABI of liborbhip.so."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import KP_DTYPE, check, ptr


class ORBextractor:
    """ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST)
    (reference include/ORBextractor.h:95); `device` selects the GPU."""

    def __init__(self, nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, device=0):
        self._lib = _lib.lib()
        h = C.c_void_p()
        check(self._lib.orb_extractor_create(int(nfeatures), float(scaleFactor), int(nlevels),
                                             int(iniThFAST), int(minThFAST), int(device), C.byref(h)))
        self._h = h
        self.nfeatures, self.nlevels, self.device = int(nfeatures), int(nlevels), int(device)
        self._scale = np.zeros(nlevels, np.float32)
        self._inv_scale = np.zeros(nlevels, np.float32)
        self._sigma2 = np.zeros(nlevels, np.float32)
        self._inv_sigma2 = np.zeros(nlevels, np.float32)
        check(self._lib.orb_extractor_get_scale_tables(h, ptr(self._scale), ptr(self._inv_scale),
                                                       ptr(self._sigma2), ptr(self._inv_sigma2)))
        self._scale_factor = float(np.float32(scaleFactor))
        self._frames = 0
        self._channels = 1
        self.mbRGB = True  # colour order of 3/4-channel input (reference Tracking::mbRGB, Camera.RGB)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.orb_extractor_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # getters, reference include/ORBextractor.h:121-163
    def GetLevels(self):
        return self.nlevels

    def GetScaleFactor(self):
        return self._scale_factor

    def GetScaleFactors(self):
        return self._scale.copy()

    def GetInverseScaleFactors(self):
        return self._inv_scale.copy()

    def GetScaleSigmaSquares(self):
        return self._sigma2.copy()

    def GetInverseScaleSigmaSquares(self):
        return self._inv_sigma2.copy()

    def features_per_level(self):
        q = np.zeros(self.nlevels, np.int32)
        u = np.zeros(16, np.int32)
        check(self._lib.orb_extractor_get_level_quotas(self._h, ptr(q), ptr(u)))
        return q, u

    def max_keypoints(self, rows, cols):
        return check(self._lib.orb_extractor_max_keypoints(self._h, int(rows), int(cols)))

    def __call__(self, image, mask=None):
        """operator()(image, mask, keypoints, descriptors) (reference
        src/ORBextractor.cc:1223).  Returns (keypoints, descriptors); the mask is
        ignored, as in the reference.  An empty image returns empty outputs."""
        image = np.asarray(image)
        if image.size == 0:
            return np.zeros(0, KP_DTYPE), np.zeros((0, 32), np.uint8)
        kps, desc = self.extract_batch(image[None])
        return kps[0], desc[0]

    def set_input_format(self, channels, rgb_order=True):
        """1 = grey; 3 / 4 = interleaved colour, converted while level 0 is written (the
        cvtColor of reference src/Tracking.cc:203-216)."""
        check(self._lib.orb_extractor_set_input_format(self._h, int(channels), int(bool(rgb_order))))
        self._channels = int(channels)

    def set_opencv_compat(self, major, minor):
        """Reproduce the arithmetic of this OpenCV release (blur taps, grey weights): the
        reference links whichever OpenCV its build finds."""
        check(self._lib.orb_extractor_set_opencv_compat(self._h, int(major), int(minor)))

    def set_blur_fusion(self, fused):
        """True (default): the descriptor kernel blurs each keypoint's patch itself; False: the
        levels are blurred to memory first, as the reference does (src/ORBextractor.cc:1297-1305).
        The results are the same."""
        check(self._lib.orb_extractor_set_blur_fusion(self._h, int(bool(fused))))

    def extract_batch(self, images):
        """images: (n, rows, cols) uint8 grey, or (n, rows, cols, 3|4) colour in the order
        self.mbRGB says.  Returns lists of per-frame keypoints/descriptors."""
        images = np.asarray(images)
        assert images.dtype == np.uint8 and images.ndim in (3, 4), "8-bit images expected"
        ch = images.shape[3] if images.ndim == 4 else 1
        self.set_input_format(ch, self.mbRGB)
        n, rows, cols = images.shape[:3]
        if not images.flags.c_contiguous:
            images = np.ascontiguousarray(images)
        cap = self.max_keypoints(rows, cols)
        kps = np.zeros((n, cap), KP_DTYPE)
        desc = np.zeros((n, cap, 32), np.uint8)
        nout = np.zeros(n, np.int32)
        check(self._lib.orb_extract_batch(self._h, n, ptr(images), rows, cols, cols * ch, rows * cols * ch, cap,
                                          ptr(kps), ptr(desc), ptr(nout)))
        self._frames = n
        return [kps[k, :nout[k]].copy() for k in range(n)], [desc[k, :nout[k]].copy() for k in range(n)]

    def extract_batch_device(self, d_images, n, rows, cols, d_kps, d_desc, d_nout, max_kp, stream=None,
                             stride=None, frame_stride=None):
        """Device-pointer form (ints holding device addresses); enqueues and returns."""
        stride = cols * self._channels if stride is None else stride
        frame_stride = rows * stride if frame_stride is None else frame_stride
        check(self._lib.orb_extract_batch_device(self._h, int(n), C.c_void_p(d_images), int(rows), int(cols),
                                                 stride, frame_stride, int(max_kp), C.c_void_p(d_kps),
                                                 C.c_void_p(d_desc), C.c_void_p(d_nout),
                                                 C.c_void_p(stream) if stream else None))
        self._frames = n

    def sync(self):
        check(self._lib.orb_extractor_sync(self._h))

    # mvImagePyramid, reference include/ORBextractor.h:166
    def pyramid_level(self, level, frame=0, padded=False):
        w, h = C.c_int(), C.c_int()
        check(self._lib.orb_get_pyramid_level(self._h, frame, level, int(padded), None, 0, C.byref(w), C.byref(h)))
        out = np.empty((h.value, w.value), np.uint8)
        check(self._lib.orb_get_pyramid_level(self._h, frame, level, int(padded), ptr(out), w.value, None, None))
        return out

    @property
    def mvImagePyramid(self):
        return [self.pyramid_level(l) for l in range(self.nlevels)]

    def blurred_level(self, level, frame=0):
        w, h = C.c_int(), C.c_int()
        check(self._lib.orb_get_blurred_level(self._h, frame, level, None, 0, C.byref(w), C.byref(h)))
        out = np.empty((h.value, w.value), np.uint8)
        check(self._lib.orb_get_blurred_level(self._h, frame, level, ptr(out), w.value, None, None))
        return out

    def fast_candidates(self, level, frame=0):
        n = check(self._lib.orb_get_fast_candidates(self._h, frame, level, 0, None))
        out = np.empty((max(n, 1), 3), np.float32)
        n = check(self._lib.orb_get_fast_candidates(self._h, frame, level, n, ptr(out)))
        return out[:n]

    def level_keypoints(self, level, frame=0):
        n = check(self._lib.orb_get_level_keypoints(self._h, frame, level, 0, None))
        out = np.empty((max(n, 1), 3), np.float32)
        n = check(self._lib.orb_get_level_keypoints(self._h, frame, level, n, ptr(out)))
        return out[:n]
