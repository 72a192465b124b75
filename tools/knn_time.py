import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from orb_slam2_amd import _lib
from orb_slam2_amd._lib import check
from orb_slam2_amd.synth import synth_descriptors
L = _lib.lib()
n = 1000000
db, q = synth_descriptors(n, n, seed=1)
dev = torch.device("cuda", 0)
d_db = torch.from_numpy(db).to(dev); d_q = torch.from_numpy(q).to(dev)
res = torch.zeros((3, n), dtype=torch.int32, device=dev)
st = torch.cuda.current_stream().cuda_stream
def run():
    check(L.orb_knn2_device(0, d_q.data_ptr(), n, d_db.data_ptr(), n, 0, 0.7, res[0].data_ptr(), res[1].data_ptr(), res[2].data_ptr(), None, st))
run(); torch.cuda.synchronize()
ts = []
for _ in range(3):
    t = time.perf_counter(); run(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
print("%-28s knn2 1M x 1M: %.4f s  checksum %d" % (os.path.basename(os.environ.get("ORBHIP_LIB", "liborbhip.so")), min(ts), int(res.sum(dtype=torch.int64).item())))
