"""orb_search_by_projection_frames_noalloc (the entry point for callers that cannot allocate:
its work arrays are a fixed array inside the library) against
orb_search_by_projection_frames_device on the same inputs, through the C ABI: bit-exact match
indices and counts, on two calls in a row, the second with a smaller input; and the size limit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROWS, COLS = 376, 1241


def _inputs(rng, npairs, max_last, max_cur, n_last, n_cur):
    from helpers import kp_struct, projection_scene
    from orb_slam2_amd import _lib
    lk = np.zeros((npairs, max_last), _lib.KP_DTYPE)
    ck = np.zeros((npairs, max_cur), _lib.KP_DTYPE)
    lobs = np.full((npairs, max_last), -1, np.int32)
    lxw = np.zeros((npairs, max_last, 3), np.float32)
    ldesc = np.zeros((npairs, max_last, 32), np.uint8)
    cdesc = np.zeros((npairs, max_cur, 32), np.uint8)
    cobs = np.full((npairs, max_cur), -1, np.int32)
    tcw = np.zeros((npairs, 16), np.float32)
    sc = None
    for p in range(npairs):
        nl, nc = int(n_last[p]), int(n_cur[p])
        sc = projection_scene(rng, nl, nc, ROWS, COLS, conflict_clusters=6)
        la, cu = sc["last"], sc["cur"]
        lk[p, :nl] = kp_struct(np.zeros(nl), np.zeros(nl), la["octave"], la["angle"])
        ck[p, :nc] = kp_struct(cu["x"], cu["y"], cu["octave"], cu["angle"])
        lobs[p, :nl], lxw[p, :nl], ldesc[p, :nl] = la["mp_obs"], la["xw"], la["mp_desc"]
        cdesc[p, :nc], cobs[p, :nc] = cu["desc"], cu["obs"]
        tcw[p] = sc["Tcw"].reshape(16)
    return dict(tcw=tcw, nl=np.asarray(n_last, np.int32), lk=lk.view(np.uint8), lobs=lobs, lxw=lxw, ldesc=ldesc,
                nc=np.asarray(n_cur, np.int32), ck=ck.view(np.uint8), cdesc=cdesc, cobs=cobs), sc


def _both(npairs, max_last, max_cur, n_last, n_cur, seed, ori):
    import torch
    from orb_slam2_amd import _lib
    from orb_slam2_amd._lib import FrameBounds, check
    L = _lib.lib()
    dev = torch.device("cuda", 0)
    host, sc = _inputs(np.random.default_rng(seed), npairs, max_last, max_cur, n_last, n_cur)
    d = {k: torch.from_numpy(v).to(dev) for k, v in host.items()}
    K, sf, bounds = sc["K"], sc["sf"], FrameBounds(*sc["bounds"])
    head = [0, npairs, max_last, max_cur, d["tcw"].data_ptr(), float(K[0]), float(K[1]), float(K[2]), float(K[3]),
            bounds, sf.ctypes.data, 8, d["nl"].data_ptr(), d["lk"].data_ptr(), d["lobs"].data_ptr(), d["lxw"].data_ptr(),
            d["ldesc"].data_ptr(), d["nc"].data_ptr(), d["ck"].data_ptr(), d["cdesc"].data_ptr(), d["cobs"].data_ptr(),
            7.0, int(ori)]
    out = []
    for noalloc in (False, True):
        m = torch.full((npairs, max_cur), -9999, dtype=torch.int32, device=dev)
        nm = torch.full((npairs,), -5, dtype=torch.int32, device=dev)
        if noalloc:
            check(L.orb_search_by_projection_frames_noalloc(*head, m.data_ptr(), nm.data_ptr(), None))
        else:
            nbytes = L.orb_match_scratch_bytes(npairs, max_last)
            scratch = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
            check(L.orb_search_by_projection_frames_device(*head, scratch.data_ptr(), nbytes, m.data_ptr(),
                                                           nm.data_ptr(), None))
        torch.cuda.synchronize()
        out.append((m.cpu().numpy(), nm.cpu().numpy()))
    return out, host


def test_noalloc_equals_device_form_twice_second_smaller():
    # first call: 6 pairs x 2100 queries; second: 3 pairs x 700 on the array the first one left behind
    for npairs, max_last, max_cur, n_last, n_cur, seed, ori in [
            (6, 2100, 2050, [2100, 1999, 2100, 17, 2000, 1500], [2050, 2000, 1800, 2050, 40, 2050], 5, True),
            (3, 700, 900, [700, 512, 1], [900, 640, 900], 6, False)]:
        (dev_res, na_res), host = _both(npairs, max_last, max_cur, n_last, n_cur, seed, ori)
        assert np.array_equal(dev_res[1], na_res[1])
        assert dev_res[1].sum() > 100
        for p in range(npairs):
            nc = int(n_cur[p])
            assert np.array_equal(dev_res[0][p, :nc], na_res[0][p, :nc]), p
            assert (na_res[0][p, :nc] != -9999).all()


def test_noalloc_rejects_sizes_beyond_its_array():
    import torch
    from orb_slam2_amd import _lib
    from orb_slam2_amd._lib import FrameBounds
    L = _lib.lib()
    t = torch.zeros(4096, dtype=torch.uint8, device="cuda:0")
    sf = np.ones(8, np.float32)
    p = t.data_ptr()
    rc = L.orb_search_by_projection_frames_noalloc(0, 33, 2000, 2000, p, 1.0, 1.0, 0.0, 0.0, FrameBounds(0, 1, 0, 1),
                                                   sf.ctypes.data, 8, p, p, p, p, p, p, p, p, p, 7.0, 1, p, p, None)
    assert rc == _lib.ERR_CAPACITY
    assert b"ORBHIP_NOALLOC_MAX_QUERIES" in L.orb_last_error()
