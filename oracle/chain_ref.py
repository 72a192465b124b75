"""CPU oracle (test infrastructure) of the benchmark's unit of work: ORBextractor::operator()
on every frame of a synthetic sequence (reference src/ORBextractor.cc:1223-1339), then
ORBmatcher::SearchByProjection(Cur, Last, th, mono) on consecutive frames
(src/ORBmatcher.cc:1633-1788), with the map points SequenceScene defines.  Imported only by
tests and by the cpu_baseline leg of bench.py."""
import numpy as np

from . import oracle as O


def match_pair(scene, sf, last_kps, last_desc, cur_kps, cur_desc, last_frame, th=7.0, check_ori=True):
    """Oracle SearchByProjection for the pair (last_frame, last_frame + 1) on given keypoints."""
    xw = scene.map_points(last_kps["x"], last_kps["y"], last_frame)
    last = dict(mp_obs=np.ones(len(last_kps), np.int32), xw=xw, mp_desc=last_desc, octave=last_kps["octave"],
                angle=last_kps["angle"])
    cur = dict(x=cur_kps["x"], y=cur_kps["y"], octave=cur_kps["octave"], angle=cur_kps["angle"], desc=cur_desc,
               obs=np.full(len(cur_kps), -1, np.int32))
    return O.search_by_projection_frames(scene.tcw(last_frame + 1), scene.K, scene.bounds, sf, last, cur, th, check_ori)


def chain_reference(scene, seq, nfeatures=2000, nlevels=8, first_frame=0, th=7.0, check_ori=True, max_kp=None):
    """Returns (frames, pairs): frames[k] = (keypoints, descriptors) of seq[k]; pairs[p] =
    (nmatches, cur_match) for Last = frame p, Cur = frame p + 1."""
    ref = O.Extractor(nfeatures, 1.2, nlevels, 20, 7)
    sf = ref.tables()["scale"]
    frames, pairs = [], []
    for k in range(len(seq)):
        kps, desc = ref(seq[k], max_kp)
        if frames:
            pk, pd = frames[-1]
            pairs.append(match_pair(scene, sf, pk, pd, kps, desc, first_frame + k - 1, th, check_ori))
        frames.append((kps, desc))
    return frames, pairs


def chain_reference_parallel(scene, seq, nfeatures=2000, nlevels=8, threads=16, th=7.0, check_ori=True):
    """The same work frame-parallel on `threads` host threads (the oracle's C functions hold no state
    and run outside the interpreter lock): every frame extracted by some thread, then every pair
    matched by some thread.  Returns the same (frames, pairs) as chain_reference."""
    import threading
    from concurrent.futures import ThreadPoolExecutor
    local = threading.local()

    def extractor():
        if not hasattr(local, "ref"):
            local.ref = O.Extractor(nfeatures, 1.2, nlevels, 20, 7)
        return local.ref

    sf = O.Extractor(nfeatures, 1.2, nlevels, 20, 7).tables()["scale"]
    with ThreadPoolExecutor(max_workers=threads) as pool:
        frames = list(pool.map(lambda k: extractor()(seq[k], None), range(len(seq))))
        pairs = list(pool.map(lambda p: match_pair(scene, sf, frames[p][0], frames[p][1], frames[p + 1][0], frames[p + 1][1],
                                                   p, th, check_ori), range(len(seq) - 1)))
    return frames, pairs
