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


def test_packed_records_merge_equals_the_int32_merge(oracle):
    """The 8-byte exchange records: packing on the device equals the tensor arithmetic of
    orb_slam2_amd.dist.pack_records, and the merge of eight shards' records equals both the merge of
    the three int32 arrays and the single scan."""
    import torch
    from orb_slam2_amd import _lib
    from orb_slam2_amd.dist import hip_merge, hip_merge_packed, pack_records, shard_bounds
    from orb_slam2_amd.synth import synth_descriptors
    L = _lib.lib()
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream().cuda_stream
    nq, ndb, world = 3001, 8000, 8
    db, q = synth_descriptors(ndb, nq, seed=9)
    db[ndb - 1] = db[0]
    q[0] = db[0]
    q[1] = 255 - db[17]
    d_q, d_db = torch.from_numpy(q).to(dev), torch.from_numpy(db).to(dev)
    shards = torch.zeros((world, 3, nq), dtype=torch.int32, device=dev)
    recs = torch.zeros((world, nq, 2), dtype=torch.int32, device=dev)
    for r in range(world):
        lo, hi = shard_bounds(ndb, world, r)
        _lib.check(L.orb_knn2_device(0, d_q.data_ptr(), nq, d_db[lo:hi].data_ptr(), hi - lo, lo, 0.7,
                                     shards[r, 0].data_ptr(), shards[r, 1].data_ptr(), shards[r, 2].data_ptr(), None, st))
        _lib.check(L.orb_knn2_pack_device(0, nq, shards[r, 0].data_ptr(), shards[r, 1].data_ptr(), shards[r, 2].data_ptr(),
                                          recs[r].data_ptr(), st))
        torch.cuda.synchronize()
        assert torch.equal(recs[r], pack_records(shards[r]))
    o12, a12 = hip_merge(0.7, 0, st)(shards.permute(1, 0, 2).contiguous())
    o8, a8 = hip_merge_packed(0.7, 0, st)(recs)
    torch.cuda.synchronize()
    assert torch.equal(o8, o12) and torch.equal(a8, a12)
    rb, rs, ri, ra = oracle.knn2_bruteforce(q, db, 0.7)
    o = o8.cpu().numpy()
    assert np.array_equal(o[0], rb) and np.array_equal(o[1], rs) and np.array_equal(o[2], ri)
    assert np.array_equal(a8.cpu().numpy(), ra)
    assert rb[0] == 0 and ri[0] == 0 and rb[1] <= 256


def test_sharded_entry_point_reports_its_ranks_and_times():
    import torch
    from orb_slam2_amd.dist import RcclKnn
    from orb_slam2_amd.synth import synth_descriptors
    dev = torch.device("cuda", 0)
    knn = RcclKnn(0, rank=0, world=1)
    try:
        assert knn.info() == (0, 1, 1)
        db, q = synth_descriptors(4000, 1000, seed=2)
        knn.search(torch.from_numpy(q).to(dev), torch.from_numpy(db).to(dev), 0, 0.7, torch.cuda.current_stream().cuda_stream)
        scan_ms, exchange_ms = knn.times()
        assert scan_ms > 0 and exchange_ms > 0
    finally:
        knn.close()
