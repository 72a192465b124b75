"""Compiled consumers of the C ABI: tests/cabi/abi_smoke.c (C99, gcc) and abi_smoke.cpp (C++17,
g++) are built against include/orbhip.h and the in-tree liborbhip.so with no Python, ctypes or
torch in their process - the way a C++ host such as the reference links the library.  Each is run
as a child process on a directory of raw arrays written here: the image and expected keypoints /
descriptors of a committed golden vector, a frame pair with the CPU oracle's SearchByProjection
result, and the committed kNN-2 golden vector.  The programs compare byte for byte (angles within
1e-4 rad, at least 99.9 % of the descriptors identical) and exit non-zero on any difference."""
import os
import subprocess

import numpy as np
import pytest

from helpers import projection_scene

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
BUILD = os.path.join(ROOT, "tests", "cabi", "_build")


def built(prog):
    """Path of a consumer program; the two small gcc / g++ builds are made here if a fresh tree lacks them
    (__graft_entry__.build() makes them too)."""
    exe = os.path.join(BUILD, prog)
    if not os.path.exists(exe):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "cabi")], check=True, timeout=120)
    assert os.path.exists(exe), "tests/cabi did not build: make -C tests/cabi"
    return exe


def write_case(d, oracle):
    from orb_slam2_amd._lib import KP_DTYPE
    from orb_slam2_amd.synth import synth_descriptors, synth_image
    g = np.load(os.path.join(GOLD, "small_240x320.npz"))
    rows, cols, seed, nf, nl = (int(v) for v in g["params"])
    img = synth_image(rows, cols, seed)
    kp = np.zeros(len(g["x"]), KP_DTYPE)
    for f in ("x", "y", "size", "angle", "response", "octave"):
        kp[f] = g[f]

    def put(name, a, dtype):
        np.ascontiguousarray(a, dtype=dtype).tofile(os.path.join(d, name))
    put("extract_params.raw", [rows, cols, nf, nl, 20, 7], np.int32)
    put("image.raw", img, np.uint8)
    kp.tofile(os.path.join(d, "expect_kps.raw"))
    put("expect_desc.raw", g["desc"], np.uint8)
    # one frame pair, arrays exactly as orb_search_by_projection_frames takes them
    rng = np.random.default_rng(77)
    nlast, ncur = 1500, 1800
    sc = projection_scene(rng, nlast, ncur, 376, 1241, conflict_clusters=20)
    want_n, want = oracle.search_by_projection_frames(sc["Tcw"], sc["K"], sc["bounds"], sc["sf"], sc["last"], sc["cur"], 7.0, True)
    assert want_n > 100
    L, Cu, b = sc["last"], sc["cur"], sc["bounds"]
    put("match_ints.raw", [nlast, ncur, 8, 1], np.int32)
    put("match_floats.raw", list(sc["K"]) + [b[0], b[1], b[2], b[3], 7.0], np.float32)
    put("match_tcw.raw", sc["Tcw"].reshape(16), np.float32)
    put("match_scale.raw", sc["sf"], np.float32)
    put("match_last_obs.raw", L["mp_obs"], np.int32)
    put("match_last_xw.raw", L["xw"], np.float32)
    put("match_last_desc.raw", L["mp_desc"], np.uint8)
    put("match_last_octave.raw", L["octave"], np.int32)
    put("match_last_angle.raw", L["angle"], np.float32)
    put("match_cur_x.raw", Cu["x"], np.float32)
    put("match_cur_y.raw", Cu["y"], np.float32)
    put("match_cur_octave.raw", Cu["octave"], np.int32)
    put("match_cur_angle.raw", Cu["angle"], np.float32)
    put("match_cur_desc.raw", Cu["desc"], np.uint8)
    put("match_cur_obs.raw", Cu["obs"], np.int32)
    put("expect_match.raw", want, np.int32)
    put("expect_nmatch.raw", [want_n], np.int32)
    gk = np.load(os.path.join(GOLD, "knn2_3000x800.npz"))
    db, q = synth_descriptors(3000, 800, seed=4)
    put("knn_q.raw", q, np.uint8)
    put("knn_db.raw", db, np.uint8)
    put("knn_ratio.raw", [0.7], np.float32)
    put("expect_knn_best.raw", gk["best"], np.int32)
    put("expect_knn_second.raw", gk["second"], np.int32)
    put("expect_knn_idx.raw", gk["idx"], np.int32)
    put("expect_knn_accept.raw", gk["accept"], np.uint8)


@pytest.mark.gpu
@pytest.mark.parametrize("prog", ["abi_smoke_c", "abi_smoke_cpp"])
def test_compiled_consumer(prog, oracle, tmp_path):
    exe = built(prog)
    write_case(str(tmp_path), oracle)
    p = subprocess.run([exe, str(tmp_path)], capture_output=True, timeout=170)
    out, err = p.stdout.decode(), p.stderr.decode()
    assert p.returncode == 0, (p.returncode, out, err)
    assert out.strip().endswith("abi_smoke ok") and "extract:" in out and "SearchByProjection:" in out and "kNN-2:" in out


def test_consumers_are_built_and_link_only_the_library():
    """(CPU) both programs exist, need liborbhip.so and no HIP / torch library of their own."""
    for prog in ("abi_smoke_c", "abi_smoke_cpp"):
        exe = built(prog)
        needed = subprocess.run(["readelf", "-d", exe], capture_output=True, check=True).stdout.decode()
        libs = [l.split("[")[1].split("]")[0] for l in needed.splitlines() if "(NEEDED)" in l]
        assert "liborbhip.so" in libs
        assert not any("hip" in l and l != "liborbhip.so" for l in libs) and not any("torch" in l for l in libs), libs
