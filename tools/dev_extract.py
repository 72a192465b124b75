import sys, time, ctypes as C
import numpy as np
sys.path.insert(0, '/root/repo')
import orb_slam2_amd._lib as _l
# dev: bind without matcher
src = open(_l.__file__).read().split("# MATCHER-BIND")[0]
from oracle import oracle as O
from orb_slam2_amd.synth import synth_image
import orb_slam2_amd._lib as L
def lib_nomatch():
    import ctypes as C
    lib = C.CDLL(L.LIB_PATH)
    return lib
from orb_slam2_amd.extractor import ORBextractor
for (rows, cols, nf) in [(480, 640, 1000), (376, 1241, 2000)]:
    img = synth_image(rows, cols, 3)
    ex = ORBextractor(nf, 1.2, 8, 20, 7)
    ref = O.Extractor(nf, 1.2, 8, 20, 7)
    t0 = time.time(); kps, desc = ex(img); t1 = time.time()
    rk, rd = ref(img)
    print(rows, cols, "gpu n", len(kps), "ref n", len(rk), "time", t1 - t0)
    for l in range(8):
        pg = ex.pyramid_level(l, padded=True); pr = ref.pyramid_level(l)
        cg = ex.fast_candidates(l); cr = ref.candidates(l)
        kg = ex.level_keypoints(l); kr = ref.level_keypoints(l)
        krr = np.stack([kr['x'], kr['y'], kr['response']], 1) if len(kr) else np.zeros((0,3))
        bg = ex.blurred_level(l); br = ref.blurred_level(l)
        print(" L%d pyr %s cand %d/%d eq %s  kps %d/%d eq %s blur %s" % (l, np.array_equal(pg, pr), len(cg), len(cr),
              cg.shape == cr.shape and np.array_equal(cg, cr), len(kg), len(krr), kg.shape == krr.shape and np.array_equal(kg, krr),
              br is None or np.array_equal(bg, br)))
    if len(kps) == len(rk):
        for fld in ('x', 'y', 'size', 'response', 'octave'):
            print("  ", fld, np.array_equal(kps[fld], rk[fld]))
        print("   angle maxdiff", np.abs(kps['angle'] - rk['angle']).max(), "exact", np.array_equal(kps['angle'], rk['angle']))
        print("   desc identical rows", (desc == rd).all(1).mean())
