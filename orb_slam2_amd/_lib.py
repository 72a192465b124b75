"""ctypes loader for liborbhip.so (the HIP kernels + C ABI).  There is no fallback:
if the library is missing or cannot be loaded, importing a compute entry point fails."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liborbhip.so")
if os.environ.get("ORBHIP_LIB"):  # another build of the library (development: timing builds side by side)
    LIB_PATH = os.environ["ORBHIP_LIB"]

OK, ERR_ARG, ERR_UNSUPPORTED, ERR_HIP, ERR_CAPACITY = 0, -1, -2, -3, -4

KP_DTYPE = np.dtype([("x", "f4"), ("y", "f4"), ("size", "f4"), ("angle", "f4"),
                     ("response", "f4"), ("octave", "i4")])


class OrbHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("orbhip error %d: %s" % (code, msg))
        self.code = code


class Camera(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("fx", "fy", "cx", "cy", "k1", "k2", "p1", "p2", "k3")]


class FrameBounds(C.Structure):
    _fields_ = [("min_x", C.c_float), ("max_x", C.c_float), ("min_y", C.c_float), ("max_y", C.c_float)]


_lib = None

# every symbol include/orbhip.h declares
EXPORTS = [
    "orb_last_error", "orb_device_count", "orb_extractor_create", "orb_extractor_destroy",
    "orb_extractor_get_scale_tables", "orb_extractor_get_level_quotas", "orb_extractor_max_keypoints",
    "orb_extract", "orb_extract_batch", "orb_extract_batch_device", "orb_extractor_sync",
    "orb_extractor_set_profiling", "orb_extractor_set_input_format", "orb_extractor_set_opencv_compat", "orb_extractor_set_blur_fusion", "orb_extractor_set_batch_split", "orb_extractor_stage_times", "orb_match_scratch_bytes", "orb_match_scratch_bytes_for",
    "orb_search_by_projection_frames_device", "orb_search_by_projection_frames_noalloc", "orb_match_windowed", "orb_match_windowed_keyframe", "orb_window_best_keyframe", "orb_stereo_match", "orb_search_by_bow", "orb_window_best", "orb_search_for_initialization",
    "orb_get_pyramid_level", "orb_get_blurred_level", "orb_get_fast_candidates", "orb_get_level_keypoints",
    "orb_hamming_distance_pairs", "orb_search_by_projection_frames", "orb_search_by_projection_mappoints",
    "orb_search_for_triangulation", "orb_knn2_device", "orb_knn2_device_algo", "orb_knn2_merge_device", "orb_knn2_bruteforce", "orb_knn2_comm_unique_id", "orb_knn2_comm_create", "orb_knn2_comm_wrap", "orb_knn2_comm_destroy", "orb_knn2_sharded",
    "orb_knn2_pack_device", "orb_knn2_merge_packed_device", "orb_knn2_comm_info", "orb_knn2_sharded_times",
    "orb_vocabulary_create", "orb_vocabulary_destroy", "orb_bow_transform_device", "orb_bow_transform", "orb_distinctive_descriptors",
    "orb_undistort_keypoints_device", "orb_undistort_points", "orb_compute_image_bounds",
]


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("liborbhip.so not built: run `make -C orb_slam2_amd/csrc` "
                          "(or __graft_entry__.build()); there is no CPU fallback")
    # PyTorch-ROCm ships its own copy of the HIP runtime.  If it is going to be used in
    # this process it has to be loaded first, so that this library binds to the same
    # runtime instead of bringing in a second one from /opt/rocm (torch then finds no GPU).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, i, f, sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
    L.orb_last_error.restype = C.c_char_p
    L.orb_extractor_create.argtypes = [i, f, i, i, i, i, C.POINTER(vp)]
    L.orb_extractor_destroy.argtypes = [vp]
    L.orb_extractor_destroy.restype = None
    L.orb_extractor_get_scale_tables.argtypes = [vp, vp, vp, vp, vp]
    L.orb_extractor_get_level_quotas.argtypes = [vp, vp, vp]
    L.orb_extractor_max_keypoints.argtypes = [vp, i, i]
    L.orb_extract.argtypes = [vp, vp, i, i, sz, i, vp, vp, vp]
    L.orb_extract_batch.argtypes = [vp, i, vp, i, i, sz, sz, i, vp, vp, vp]
    L.orb_extract_batch_device.argtypes = [vp, i, vp, i, i, sz, sz, i, vp, vp, vp, vp]
    L.orb_extractor_sync.argtypes = [vp]
    L.orb_extractor_set_profiling.argtypes = [vp, i]
    L.orb_extractor_set_input_format.argtypes = [vp, i, i]
    L.orb_extractor_set_opencv_compat.argtypes = [vp, i, i]
    L.orb_extractor_set_blur_fusion.argtypes = [vp, i]
    L.orb_extractor_set_batch_split.argtypes = [vp, i]
    L.orb_extractor_stage_times.argtypes = [vp, vp]
    L.orb_match_scratch_bytes.argtypes = [i, i]
    L.orb_match_scratch_bytes.restype = sz
    L.orb_match_scratch_bytes_for.argtypes = [i, i, i]
    L.orb_match_scratch_bytes_for.restype = sz
    L.orb_search_by_projection_frames_device.argtypes = (
        [i, i, i, i, vp, f, f, f, f, FrameBounds, vp, i] + [vp] * 9 + [f, i, vp, sz, vp, vp, vp])
    L.orb_search_by_projection_frames_noalloc.argtypes = (
        [i, i, i, i, vp, f, f, f, f, FrameBounds, vp, i] + [vp] * 9 + [f, i, vp, vp, vp])
    L.orb_get_pyramid_level.argtypes = [vp, i, i, i, vp, sz, C.POINTER(i), C.POINTER(i)]
    L.orb_get_blurred_level.argtypes = [vp, i, i, vp, sz, C.POINTER(i), C.POINTER(i)]
    L.orb_get_fast_candidates.argtypes = [vp, i, i, i, vp]
    L.orb_get_level_keypoints.argtypes = [vp, i, i, i, vp]
    L.orb_hamming_distance_pairs.argtypes = [i, vp, vp, i, vp]
    L.orb_search_by_projection_frames.argtypes = (
        [i, i, i, i, vp, f, f, f, f, FrameBounds, vp, i] + [vp] * 13 + [f, i, vp, vp])
    L.orb_search_by_projection_mappoints.argtypes = (
        [i, FrameBounds, vp, i, i] + [vp] * 5 + [i] + [vp] * 7 + [f, f, vp, vp])
    L.orb_match_windowed.argtypes = [i, FrameBounds, i] + [vp] * 6 + [i] + [vp] * 8 + [i, i, vp, vp]
    L.orb_match_windowed_keyframe.argtypes = L.orb_match_windowed.argtypes
    L.orb_search_for_initialization.argtypes = [i, FrameBounds, i, vp, vp, vp, vp, vp, i, vp, vp, vp, vp, vp, f, f, i, vp, vp]
    L.orb_window_best.argtypes = [i, FrameBounds, i, vp, vp, vp, vp, vp, i, C.c_double, i, vp, vp, vp, vp, vp, vp, vp, vp]
    L.orb_window_best_keyframe.argtypes = L.orb_window_best.argtypes
    L.orb_search_by_bow.argtypes = [i, i, vp, vp, vp, i, vp, vp, vp, i, vp, vp, vp, i, vp, vp, vp, f, i, i, vp, vp]
    L.orb_stereo_match.argtypes = [i, i, i] + [vp] * 4 + [i] + [vp] * 5 + [i, f, f, i, vp, vp, vp]
    L.orb_search_for_triangulation.argtypes = (
        [i, i] + [vp] * 6 + [i] + [vp] * 6 + [i, vp, vp, vp, i, vp, vp, vp, vp, f, f, vp, vp, i, i, vp, vp])
    L.orb_knn2_device.argtypes = [i, vp, i, vp, i, i, f, vp, vp, vp, vp, vp]
    L.orb_knn2_device_algo.argtypes = [i, vp, i, vp, i, i, f, vp, vp, vp, vp, vp, i]
    L.orb_knn2_merge_device.argtypes = [i, i, i, vp, vp, vp, f, vp, vp, vp, vp, vp]
    L.orb_knn2_comm_unique_id.argtypes = [vp]
    L.orb_knn2_comm_create.argtypes = [vp, i, i, i, C.POINTER(vp)]
    L.orb_knn2_comm_wrap.argtypes = [vp, i, i, i, C.POINTER(vp)]
    L.orb_knn2_comm_destroy.argtypes = [vp]
    L.orb_knn2_comm_destroy.restype = None
    L.orb_knn2_sharded.argtypes = [vp, vp, i, vp, i, i, f, vp, vp, vp, vp, vp]
    L.orb_knn2_pack_device.argtypes = [i, i, vp, vp, vp, vp, vp]
    L.orb_knn2_merge_packed_device.argtypes = [i, i, i, vp, f, vp, vp, vp, vp, vp]
    L.orb_knn2_comm_info.argtypes = [vp, C.POINTER(i), C.POINTER(i), C.POINTER(i)]
    L.orb_knn2_sharded_times.argtypes = [vp, C.POINTER(f), C.POINTER(f)]
    L.orb_knn2_bruteforce.argtypes = [i, vp, i, vp, i, f, vp, vp, vp, vp]
    L.orb_vocabulary_create.argtypes = [i, i, i, vp, vp, vp, vp, vp, C.POINTER(vp)]
    L.orb_vocabulary_destroy.argtypes = [vp]
    L.orb_vocabulary_destroy.restype = None
    L.orb_bow_transform_device.argtypes = [vp, i, vp, i, vp, vp, vp]
    L.orb_bow_transform.argtypes = [vp, i, vp, i, vp, vp]
    L.orb_distinctive_descriptors.argtypes = [i, i, vp, vp, vp, vp]
    L.orb_undistort_keypoints_device.argtypes = [i, i, i, vp, vp, Camera, vp, vp]
    L.orb_undistort_points.argtypes = [i, i, vp, Camera, vp]
    L.orb_compute_image_bounds.argtypes = [i, i, Camera, C.POINTER(FrameBounds)]
    _lib = L
    return L


def check(rc):
    if rc < 0:
        raise OrbHipError(rc, lib().orb_last_error().decode("utf-8", "replace"))
    return rc


def ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)
