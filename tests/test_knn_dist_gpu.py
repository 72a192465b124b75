"""orb_knn2_comm_* / orb_knn2_sharded (shard scan -> ncclAllGather -> merge inside the library)
through the C ABI.  A one-GPU box can only form a communicator of one rank; that still runs the
whole path (RCCL bound at run time, exchange buffers, the strided merge) and must equal both the
plain one-GPU search and the CPU oracle, bit for bit.  The rank-order merge rule for several
shards is covered by test_knn2_merge_of_shards and, with world_size 2, by tests/test_dist_cpu.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_sharded_entry_point_single_rank_equals_plain_search_and_oracle(oracle):
    import torch
    from orb_slam2_amd.dist import RcclKnn
    from orb_slam2_amd import knn2_bruteforce
    from orb_slam2_amd.synth import synth_descriptors
    dev = torch.device("cuda", 0)
    knn = RcclKnn(0, rank=0, world=1)
    try:
        for nq, ndb, base in [(3000, 5000, 0), (700, 2100, 40000), (5000, 300, 7)]:  # growing, then shrinking, then growing n_q
            db, q = synth_descriptors(ndb, nq, seed=nq)
            db[5] = db[2]
            d_q, d_db = torch.from_numpy(q).to(dev), torch.from_numpy(db).to(dev)
            out, acc = knn.search(d_q, d_db, base, 0.7, torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            b, s, i, a = knn2_bruteforce(q, db, 0.7)
            rb, rs, ri, ra = oracle.knn2_bruteforce(q, db, 0.7)
            o = out.cpu().numpy()
            assert np.array_equal(o[0], rb) and np.array_equal(o[1], rs) and np.array_equal(acc.cpu().numpy(), ra)
            assert np.array_equal(o[2], np.where(ri >= 0, ri + base, -1))
            assert np.array_equal(o[0], b) and np.array_equal(o[1], s) and np.array_equal(o[2] - base, i)
    finally:
        knn.close()


def test_sharded_entry_point_rejects_bad_arguments():
    import ctypes as C
    from orb_slam2_amd import _lib
    L = _lib.lib()
    h = C.c_void_p()
    uid = np.zeros(128, np.uint8)
    assert L.orb_knn2_comm_create(uid.ctypes.data, 2, 2, 0, C.byref(h)) == _lib.ERR_ARG
    assert L.orb_knn2_comm_wrap(None, 0, 1, 0, C.byref(h)) == _lib.ERR_ARG
    assert L.orb_knn2_sharded(None, None, 5, None, 0, 0, 0.7, None, None, None, None, None) == _lib.ERR_ARG
