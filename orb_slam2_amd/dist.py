"""Multi-GPU Hamming kNN-2: the database is split into contiguous row shards (one per
rank, ascending), every rank scans its shard for all queries, the per-query
(best, second, index) triples are exchanged with one all-gather and merged.  Shards are
in index order, so "earlier shard keeps ties" reproduces the single scan's first-minimum
rule and the merged result equals a one-GPU search bit for bit.

The reference has no multi-device code; this is the one path here with a real exchange
step.  Frame batches need no communication (each rank extracts its own frames).
"""
import torch
import torch.distributed as dist


def shard_bounds(n_rows, world, rank):
    """Contiguous, ascending, near-equal row ranges: [lo, hi) of `rank`."""
    return n_rows * rank // world, n_rows * (rank + 1) // world


def pack_records(res):
    """[3, n_q] int32 (best, second, global index) -> [n_q, 2] int32: the 8-byte records the ranks
    exchange, word 0 = best | second << 16 (distances are at most 256), word 1 = index (-1 = none).
    Plain tensor arithmetic, so the same code serves host (gloo) and device tensors."""
    return torch.stack([res[0] | (res[1] << 16), res[2]], dim=1).contiguous()


def unpack_records(rec):
    """[world, n_q, 2] int32 records -> [3, world, n_q] int32 (best, second, index)."""
    w0, w1 = rec[..., 0], rec[..., 1]
    return torch.stack([w0 & 0xFFFF, (w0 >> 16) & 0xFFFF, w1], dim=0).contiguous()


def knn2_sharded(local_search, merge, n_q, device, world, group=None, codec="packed8", merge_packed=None,
                 timings=None):
    """local_search() -> int32 tensor [3, n_q] (best, second, global index) for this rank's
    shard; merge(g) with g int32 [3, world, n_q] -> (int32 [3, n_q], uint8 [n_q] accept).
    Returns the merged tensors (identical on every rank).

    codec: what one rank sends per query - "packed8": one 8-byte record (pack_records), the default;
    "int32x3": the three int32 values as they are (12 bytes).  Both give the same result.
    merge_packed(rec) with rec int32 [world, n_q, 2], if given, merges the records directly (the HIP
    kernel does); otherwise they are unpacked and handed to merge.
    timings: a dict that receives "scan_s" and "exchange_s" (host clock; the caller synchronises the
    device in local_search / merge if it wants device time)."""
    import time
    t0 = time.perf_counter()
    res = local_search()
    assert res.shape == (3, n_q) and res.dtype == torch.int32
    if timings is not None:
        if res.is_cuda:
            torch.cuda.synchronize(res.device)
        timings["scan_s"] = time.perf_counter() - t0
        t0 = time.perf_counter()
    send = pack_records(res) if codec == "packed8" else res.contiguous()
    if world > 1:
        # (the ranks' blocks are concatenated along the first dimension: the form every backend takes)
        shape = (world * send.shape[0],) + tuple(send.shape[1:])
        if send.is_cuda and dist.get_backend(group) == "gloo":
            # gloo carries host tensors only: stage through the host (CPU rehearsals of the GPU path)
            gathered = torch.empty(shape, dtype=torch.int32)
            dist.all_gather_into_tensor(gathered, send.cpu(), group=group)
            gathered = gathered.to(device)
        else:
            gathered = torch.empty(shape, dtype=torch.int32, device=device)
            dist.all_gather_into_tensor(gathered, send, group=group)
        gathered = gathered.view((world,) + tuple(send.shape))
    else:
        gathered = send.unsqueeze(0)
    if codec == "packed8":
        out = merge_packed(gathered) if merge_packed is not None else merge(unpack_records(gathered))
    else:
        out = merge(gathered.permute(1, 0, 2).contiguous())
    if timings is not None:
        if out[0].is_cuda:
            torch.cuda.synchronize(out[0].device)
        timings["exchange_s"] = time.perf_counter() - t0
    return out


def hip_local_search(d_q, d_db_shard, index_base, ratio, device_index, stream=0):
    """Closure running orb_knn2_device on this rank's shard."""
    from . import _lib
    L = _lib.lib()
    n_q, n_db = d_q.shape[0], d_db_shard.shape[0]
    res = torch.empty((3, n_q), dtype=torch.int32, device=d_q.device)

    def run():
        _lib.check(L.orb_knn2_device(device_index, d_q.data_ptr(), n_q, d_db_shard.data_ptr(), n_db, index_base,
                                     ratio, res[0].data_ptr(), res[1].data_ptr(), res[2].data_ptr(), None, stream))
        return res
    return run


def hip_merge(ratio, device_index, stream=0):
    """Closure running orb_knn2_merge_device."""
    from . import _lib
    L = _lib.lib()

    def run(g):
        _, world, n_q = g.shape
        out = torch.empty((3, n_q), dtype=torch.int32, device=g.device)
        acc = torch.empty(n_q, dtype=torch.uint8, device=g.device)
        _lib.check(L.orb_knn2_merge_device(device_index, world, n_q, g[0].data_ptr(), g[1].data_ptr(), g[2].data_ptr(),
                                           ratio, out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(),
                                           acc.data_ptr(), stream))
        return out, acc
    return run


def hip_merge_packed(ratio, device_index, stream=0):
    """Closure running orb_knn2_merge_packed_device on gathered 8-byte records [world, n_q, 2]."""
    from . import _lib
    L = _lib.lib()

    def run(rec):
        world, n_q, _ = rec.shape
        out = torch.empty((3, n_q), dtype=torch.int32, device=rec.device)
        acc = torch.empty(n_q, dtype=torch.uint8, device=rec.device)
        _lib.check(L.orb_knn2_merge_packed_device(device_index, world, n_q, rec.data_ptr(), ratio, out[0].data_ptr(),
                                                  out[1].data_ptr(), out[2].data_ptr(), acc.data_ptr(), stream))
        return out, acc
    return run


class RcclKnn:
    """The same exchange through the library's own C entry points (orb_knn2_comm_* /
    orb_knn2_sharded: shard scan, ncclAllGather, merge, all enqueued on one stream), for hosts
    that are not Python; torch.distributed only carries the 128-byte communicator id from
    rank 0 to the other ranks here."""

    def __init__(self, device_index, rank=0, world=1, group=None):
        import ctypes as C

        import numpy as np

        from . import _lib
        self._lib = _lib
        self.L = _lib.lib()
        self.rank, self.world, self.device_index = rank, world, device_index
        uid = np.zeros(128, np.uint8)
        if rank == 0:
            _lib.check(self.L.orb_knn2_comm_unique_id(uid.ctypes.data))
        if world > 1:
            t = torch.from_numpy(uid).to(torch.device("cuda", device_index) if dist.get_backend(group) == "nccl" else "cpu")
            dist.broadcast(t, src=0, group=group)
            uid = t.cpu().numpy()
        h = C.c_void_p()
        _lib.check(self.L.orb_knn2_comm_create(uid.ctypes.data, rank, world, device_index, C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self.L.orb_knn2_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def search(self, d_q, d_db_shard, index_base, ratio, stream=0):
        """-> (int32 [3, n_q] best / second / global index, uint8 [n_q] accept), identical on every rank."""
        n_q = d_q.shape[0]
        out = torch.empty((3, n_q), dtype=torch.int32, device=d_q.device)
        acc = torch.empty(n_q, dtype=torch.uint8, device=d_q.device)
        self._lib.check(self.L.orb_knn2_sharded(self._h, d_q.data_ptr(), n_q, d_db_shard.data_ptr(), d_db_shard.shape[0],
                                                index_base, ratio, out[0].data_ptr(), out[1].data_ptr(),
                                                out[2].data_ptr(), acc.data_ptr(), stream))
        return out, acc

    def info(self):
        """(rank, world, ranks the communicator itself reports)"""
        import ctypes as C
        r, w, n = C.c_int(), C.c_int(), C.c_int()
        self._lib.check(self.L.orb_knn2_comm_info(self._h, C.byref(r), C.byref(w), C.byref(n)))
        return r.value, w.value, n.value

    def times(self):
        """(scan_ms, exchange_ms) of the last search, by HIP events on its stream"""
        import ctypes as C
        a, b = C.c_float(), C.c_float()
        self._lib.check(self.L.orb_knn2_sharded_times(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value
