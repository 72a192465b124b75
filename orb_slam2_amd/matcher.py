"""Host-side mirror of ORB_SLAM2::ORBmatcher (reference include/ORBmatcher.h:41-221) over
the C ABI of liborbhip.so.  The candidate scans and Hamming distances run on the GPU; this
layer only flattens the caller's objects into arrays and writes the matches back."""
import ctypes as C

import numpy as np

from . import _lib
from .frame import KeyFrame
from ._lib import FrameBounds, check, ptr


def _f32(a):
    return np.ascontiguousarray(a, np.float32)


def _i32(a):
    return np.ascontiguousarray(a, np.int32)


def _u8(a):
    return np.ascontiguousarray(a, np.uint8)


def _mat3_mul_add(R, x, t):
    """R*x + t the way OpenCV's small-matrix gemm evaluates it for CV_32F operands:
    three-term float dot products accumulated left to right, then one float add
    (used by the reference at src/ORBmatcher.cc:867, 1675)."""
    R, x, t = np.asarray(R, np.float32), np.asarray(x, np.float32), np.asarray(t, np.float32)
    out = np.empty(3, np.float32)
    for r in range(3):
        acc = np.float32(R[r, 0] * x[0])
        acc = np.float32(acc + np.float32(R[r, 1] * x[1]))
        acc = np.float32(acc + np.float32(R[r, 2] * x[2]))
        out[r] = np.float32(acc + t[r])
    return out


class ORBmatcher:
    TH_LOW = 50
    TH_HIGH = 100
    HISTO_LENGTH = 30

    def __init__(self, nnratio=0.6, checkOri=True, device=0):
        self._lib = _lib.lib()
        self.mfNNratio = float(np.float32(nnratio))
        self.mbCheckOrientation = bool(checkOri)
        self.device = int(device)

    # ---- DescriptorDistance (reference src/ORBmatcher.cc:1991-2011) ----------------
    @staticmethod
    def DescriptorDistance(a, b, device=0):
        return int(ORBmatcher.DescriptorDistances(np.asarray(a).reshape(1, 32), np.asarray(b).reshape(1, 32), device)[0])

    @staticmethod
    def DescriptorDistances(a, b, device=0):
        a, b = _u8(a).reshape(-1, 32), _u8(b).reshape(-1, 32)
        assert a.shape == b.shape
        out = np.zeros(len(a), np.int32)
        check(_lib.lib().orb_hamming_distance_pairs(int(device), ptr(a), ptr(b), len(a), ptr(out)))
        return out

    # ---- SearchByProjection(Frame&, const Frame&, th, bMono) (:1633) ----------------
    def SearchByProjection(self, *args):
        """The four overloads of the reference (include/ORBmatcher.h:73,93,107,121),
        told apart by their argument types."""
        if len(args) >= 3 and isinstance(args[2], (set, frozenset)):
            return self._search_reloc(*args)            # (Frame&, KeyFrame*, set, th, ORBdist)
        if len(args) >= 2 and isinstance(args[1], np.ndarray):
            return self._search_sim3(*args)             # (KeyFrame*, Scw, points, matched, th)
        if len(args) >= 2 and hasattr(args[1], "mvpMapPoints"):
            return self._search_frames(*args)           # (Frame&, const Frame&, th, bMono)
        return self._search_mappoints(*args)            # (Frame&, vector<MapPoint*>, th)

    def _windowed(self, F, taken, q, max_dist, check_ori):
        k = F.mvKeysUn
        nq = len(q["active"])
        match = np.full(max(F.N, 1), -1, np.int32)
        nm = C.c_int(0)
        b = FrameBounds(F.mnMinX, F.mnMaxX, F.mnMinY, F.mnMaxY)
        # a KeyFrame answers GetFeaturesInArea with its int-truncated origin (reference
        # include/KeyFrame.h:411-414, src/KeyFrame.cc:732-776), a Frame with the float one
        from .frame import KeyFrame
        fn = self._lib.orb_match_windowed_keyframe if isinstance(F, KeyFrame) else self._lib.orb_match_windowed
        check(fn(
            self.device, b, F.N, ptr(_f32(k["x"])), ptr(_f32(k["y"])), ptr(_i32(k["octave"])), ptr(_f32(k["angle"])),
            ptr(F.mDescriptors), ptr(_u8(taken)), nq, ptr(_u8(q["active"])), ptr(_f32(q["u"])), ptr(_f32(q["v"])),
            ptr(_f32(q["r"])), ptr(_i32(q["minl"])), ptr(_i32(q["maxl"])), ptr(_f32(q["angle"])), ptr(_u8(q["desc"])),
            int(max_dist), int(check_ori), ptr(match), C.byref(nm)))
        return nm.value, match

    @staticmethod
    def _blank_queries(n):
        n = max(n, 1)
        return dict(active=np.zeros(n, np.uint8), u=np.zeros(n, np.float32), v=np.zeros(n, np.float32),
                    r=np.zeros(n, np.float32), minl=np.zeros(n, np.int32), maxl=np.zeros(n, np.int32),
                    angle=np.zeros(n, np.float32), desc=np.zeros((n, 32), np.uint8))

    # ---- SearchByProjection(Frame&, KeyFrame*, const set<MapPoint*>&, th, ORBdist) (:1800) ----
    def _search_reloc(self, CurrentFrame, pKF, sAlreadyFound, th, ORBdist):
        f32 = np.float32
        Rcw, tcw = CurrentFrame.mTcw[:3, :3], CurrentFrame.mTcw[:3, 3]
        Ow = (-(Rcw.T.astype(f32) @ tcw)).astype(f32)
        vpMPs = pKF.GetMapPointMatches()
        q = self._blank_queries(len(vpMPs))
        for i, pMP in enumerate(vpMPs):
            if pMP is None or pMP.isBad() or pMP in sAlreadyFound:
                continue
            x3Dw = pMP.GetWorldPos()
            x3Dc = _mat3_mul_add(Rcw, x3Dw, tcw)
            invzc = f32(1.0 / np.float64(x3Dc[2])) if x3Dc[2] != 0 else f32(np.inf)
            u = f32(f32(f32(f32(CurrentFrame.fx) * x3Dc[0]) * invzc) + f32(CurrentFrame.cx))
            v = f32(f32(f32(f32(CurrentFrame.fy) * x3Dc[1]) * invzc) + f32(CurrentFrame.cy))
            if u < CurrentFrame.mnMinX or u > CurrentFrame.mnMaxX or v < CurrentFrame.mnMinY or v > CurrentFrame.mnMaxY:
                continue
            PO = (x3Dw - Ow).astype(f32)
            dist3D = f32(np.sqrt(np.float64(PO[0]) ** 2 + np.float64(PO[1]) ** 2 + np.float64(PO[2]) ** 2))  # cv::norm
            if dist3D < pMP.GetMinDistanceInvariance() or dist3D > pMP.GetMaxDistanceInvariance():
                continue
            n = pMP.PredictScale(dist3D, CurrentFrame)
            q["active"][i] = 1
            q["u"][i], q["v"][i] = u, v
            q["r"][i] = f32(f32(th) * CurrentFrame.mvScaleFactors[n])
            q["minl"][i], q["maxl"][i] = n - 1, n + 1
            q["angle"][i] = pKF.mvKeysUn["angle"][i]
            q["desc"][i] = pMP.GetDescriptor()
        taken = np.array([mp is not None for mp in CurrentFrame.mvpMapPoints] or [0], np.uint8)
        nm, match = self._windowed(CurrentFrame, taken, q, ORBdist, self.mbCheckOrientation)
        for i2 in range(CurrentFrame.N):
            if match[i2] >= 0:
                CurrentFrame.mvpMapPoints[i2] = vpMPs[match[i2]]
            elif match[i2] == -2:
                CurrentFrame.mvpMapPoints[i2] = None
        return nm

    # ---- SearchByProjection(KeyFrame*, cv::Mat Scw, points, matched, th) (:394) ----------
    def _search_sim3(self, pKF, Scw, vpPoints, vpMatched, th):
        f32 = np.float32
        Scw = np.asarray(Scw, f32).reshape(4, 4)
        sRcw = Scw[:3, :3]
        scw = f32(np.sqrt(np.float64(f32(sRcw[0, 0] * sRcw[0, 0]) + f32(sRcw[0, 1] * sRcw[0, 1]) + f32(sRcw[0, 2] * sRcw[0, 2]))))
        Rcw = (sRcw / scw).astype(f32)
        tcw = (Scw[:3, 3] / scw).astype(f32)
        Ow = (-(Rcw.T @ tcw)).astype(f32)
        found = {id(mp) for mp in vpMatched if mp is not None}
        q = self._blank_queries(len(vpPoints))
        for i, pMP in enumerate(vpPoints):
            if pMP.isBad() or id(pMP) in found:
                continue
            p3Dw = pMP.GetWorldPos()
            p3Dc = _mat3_mul_add(Rcw, p3Dw, tcw)
            if p3Dc[2] < 0.0:
                continue
            invz = f32(1.0) / p3Dc[2]
            u = f32(f32(f32(pKF.fx) * f32(p3Dc[0] * invz)) + f32(pKF.cx))
            v = f32(f32(f32(pKF.fy) * f32(p3Dc[1] * invz)) + f32(pKF.cy))
            if not pKF.IsInImage(u, v):
                continue
            PO = (p3Dw - Ow).astype(f32)
            dist = f32(np.sqrt(np.float64(PO[0]) ** 2 + np.float64(PO[1]) ** 2 + np.float64(PO[2]) ** 2))
            if dist < pMP.GetMinDistanceInvariance() or dist > pMP.GetMaxDistanceInvariance():
                continue
            Pn = pMP.GetNormal()
            if np.float64(f32(f32(f32(PO[0] * Pn[0]) + f32(PO[1] * Pn[1])) + f32(PO[2] * Pn[2]))) < 0.5 * np.float64(dist):
                continue
            n = pMP.PredictScale(dist, pKF)
            q["active"][i] = 1
            q["u"][i], q["v"][i] = u, v
            q["r"][i] = f32(f32(th) * pKF.mvScaleFactors[n])
            q["minl"][i], q["maxl"][i] = n - 1, n
            q["desc"][i] = pMP.GetDescriptor()
        taken = np.array([mp is not None for mp in vpMatched] or [0], np.uint8)
        nm, match = self._windowed(pKF, taken, q, self.TH_LOW, False)
        for idx in range(pKF.N):
            if match[idx] >= 0:
                vpMatched[idx] = vpPoints[match[idx]]
        return nm

    def _search_frames(self, CurrentFrame, LastFrame, th, bMono=True):
        n, cur_match = self.search_by_projection_frames_batch([CurrentFrame], [LastFrame], th)
        m = cur_match[0]
        for i2 in range(CurrentFrame.N):
            if m[i2] >= 0:
                CurrentFrame.mvpMapPoints[i2] = LastFrame.mvpMapPoints[m[i2]]
            elif m[i2] == -2:
                CurrentFrame.mvpMapPoints[i2] = None
        return int(n[0])

    def search_by_projection_frames_batch(self, cur_frames, last_frames, th):
        """Independent (Last, Cur) pairs with equal intrinsics in one launch.  Returns
        (n_matches[pairs], cur_match[pairs][max_cur]) with the C ABI's encoding."""
        npairs = len(cur_frames)
        max_last = max(1, max(f.N for f in last_frames))
        max_cur = max(1, max(f.N for f in cur_frames))
        c0 = cur_frames[0]
        tcw = np.zeros((npairs, 16), np.float32)
        n_last, n_cur = np.zeros(npairs, np.int32), np.zeros(npairs, np.int32)
        l_obs = np.full((npairs, max_last), -1, np.int32)
        l_xw = np.zeros((npairs, max_last, 3), np.float32)
        l_desc = np.zeros((npairs, max_last, 32), np.uint8)
        l_oct = np.zeros((npairs, max_last), np.int32)
        l_ang = np.zeros((npairs, max_last), np.float32)
        c_x, c_y, c_ang = (np.zeros((npairs, max_cur), np.float32) for _ in range(3))
        c_oct = np.zeros((npairs, max_cur), np.int32)
        c_desc = np.zeros((npairs, max_cur, 32), np.uint8)
        c_obs = np.full((npairs, max_cur), -1, np.int32)
        for p, (cur, last) in enumerate(zip(cur_frames, last_frames)):
            tcw[p] = cur.mTcw.reshape(16)
            n_last[p], n_cur[p] = last.N, cur.N
            for i, mp in enumerate(last.mvpMapPoints):
                if mp is not None:
                    l_obs[p, i] = mp.Observations()
                    l_xw[p, i] = mp.GetWorldPos()
                    l_desc[p, i] = mp.GetDescriptor()
            l_oct[p, :last.N] = last.mvKeys["octave"]
            l_ang[p, :last.N] = last.mvKeysUn["angle"]
            c_x[p, :cur.N], c_y[p, :cur.N] = cur.mvKeysUn["x"], cur.mvKeysUn["y"]
            c_oct[p, :cur.N], c_ang[p, :cur.N] = cur.mvKeysUn["octave"], cur.mvKeysUn["angle"]
            c_desc[p, :cur.N] = cur.mDescriptors
            for i2, mp in enumerate(cur.mvpMapPoints):
                if mp is not None:
                    c_obs[p, i2] = mp.Observations()
        sf = _f32(c0.mvScaleFactors)
        cur_match = np.zeros((npairs, max_cur), np.int32)
        nm = np.zeros(npairs, np.int32)
        b = FrameBounds(c0.mnMinX, c0.mnMaxX, c0.mnMinY, c0.mnMaxY)
        check(self._lib.orb_search_by_projection_frames(
            self.device, npairs, max_last, max_cur, ptr(tcw), c0.fx, c0.fy, c0.cx, c0.cy, b, ptr(sf), len(sf),
            ptr(n_last), ptr(l_obs), ptr(l_xw), ptr(l_desc), ptr(l_oct), ptr(l_ang),
            ptr(n_cur), ptr(c_x), ptr(c_y), ptr(c_oct), ptr(c_ang), ptr(c_desc), ptr(c_obs),
            float(th), int(self.mbCheckOrientation), ptr(cur_match), ptr(nm)))
        return nm, cur_match

    # ---- SearchByProjection(Frame&, const vector<MapPoint*>&, th) (:64) --------------
    def _search_mappoints(self, F, vpMapPoints, th=3.0):
        nmp = len(vpMapPoints)
        track = np.zeros(max(nmp, 1), np.uint8)
        level = np.zeros(max(nmp, 1), np.int32)
        vcos, px, py = (np.zeros(max(nmp, 1), np.float32) for _ in range(3))
        desc = np.zeros((max(nmp, 1), 32), np.uint8)
        obs = np.zeros(max(nmp, 1), np.int32)
        for i, mp in enumerate(vpMapPoints):
            track[i] = 1 if (mp.mbNeedTrackInView and not mp.isBad()) else 0
            level[i], vcos[i] = mp.mnTrackedScaleLevel, mp.mTrackedViewCosine
            px[i], py[i] = mp.mTrackedProjX, mp.mTrackedProjY
            desc[i], obs[i] = mp.GetDescriptor(), mp.Observations()
        f_obs = np.full(max(F.N, 1), -1, np.int32)
        for idx, mp in enumerate(F.mvpMapPoints):
            if mp is not None:
                f_obs[idx] = mp.Observations()
        f_x, f_y = _f32(F.mvKeysUn["x"]), _f32(F.mvKeysUn["y"])
        f_oct = _i32(F.mvKeysUn["octave"])
        sf = _f32(F.mvScaleFactors)
        match = np.full(max(F.N, 1), -1, np.int32)
        nm = C.c_int(0)
        b = FrameBounds(F.mnMinX, F.mnMaxX, F.mnMinY, F.mnMaxY)
        check(self._lib.orb_search_by_projection_mappoints(
            self.device, b, ptr(sf), len(sf), F.N, ptr(f_x), ptr(f_y), ptr(f_oct), ptr(F.mDescriptors), ptr(f_obs),
            nmp, ptr(track), ptr(level), ptr(vcos), ptr(px), ptr(py), ptr(desc), ptr(obs),
            float(th), self.mfNNratio, ptr(match), C.byref(nm)))
        for idx in range(F.N):
            if match[idx] >= 0:
                F.mvpMapPoints[idx] = vpMapPoints[match[idx]]
        return nm.value

    # ---- SearchForTriangulation (:853) ---------------------------------------------
    def SearchForInitialization(self, F1, F2, vbPrevMatched, vnMatches12, windowSize=10):
        """reference src/ORBmatcher.cc:543-686.  vbPrevMatched: (N1, 2) float32 array of the
        previously matched positions, updated in place (:681-683); vnMatches12: list, replaced."""
        prev = np.asarray(vbPrevMatched, np.float32)
        k1, k2 = F1.mvKeysUn, F2.mvKeysUn
        m12 = np.full(max(F1.N, 1), -1, np.int32)
        nm = C.c_int(0)
        b = FrameBounds(F2.mnMinX, F2.mnMaxX, F2.mnMinY, F2.mnMaxY)
        check(self._lib.orb_search_for_initialization(
            self.device, b, F1.N, ptr(_i32(k1["octave"])), ptr(_f32(k1["angle"])), ptr(F1.mDescriptors),
            ptr(_f32(prev[:, 0])), ptr(_f32(prev[:, 1])), F2.N, ptr(_f32(k2["x"])), ptr(_f32(k2["y"])),
            ptr(_i32(k2["octave"])), ptr(_f32(k2["angle"])), ptr(F2.mDescriptors), float(windowSize),
            float(self.mfNNratio), int(self.mbCheckOrientation), ptr(m12), C.byref(nm)))
        vnMatches12.clear()
        vnMatches12.extend(int(v) for v in m12[:F1.N])
        for i1 in range(F1.N):
            if m12[i1] >= 0:
                vbPrevMatched[i1, 0] = k2["x"][m12[i1]]
                vbPrevMatched[i1, 1] = k2["y"][m12[i1]]
        return nm.value

    def SearchByBoW(self, pKF, other, vpMatches):
        """Both overloads of the reference (include/ORBmatcher.h): `other` is a Frame
        (src/ORBmatcher.cc:222-381, vpMatches indexed by the frame's features) or a KeyFrame
        (:698-841, vpMatches indexed by pKF's features)."""
        to_frame = not isinstance(other, KeyFrame)
        mps1 = pKF.GetMapPointMatches()
        ok1 = np.array([mp is not None and not mp.isBad() for mp in mps1] or [0], np.uint8)
        if to_frame:
            ok2 = np.ones(max(other.N, 1), np.uint8)
            a2 = other.mvKeys["angle"]
            max_dist = self.TH_LOW
        else:
            mps2 = other.GetMapPointMatches()
            ok2 = np.array([mp is not None and not mp.isBad() for mp in mps2] or [0], np.uint8)
            a2 = other.mvKeysUn["angle"]
            max_dist = self.TH_LOW - 1
        nd1, st1, it1 = _flat_featvec(pKF.mFeatVec)
        nd2, st2, it2 = _flat_featvec(other.mFeatVec)
        m12 = np.full(max(pKF.N, 1), -1, np.int32)
        nm = C.c_int(0)
        check(self._lib.orb_search_by_bow(
            self.device, pKF.N, ptr(_f32(pKF.mvKeysUn["angle"])), ptr(pKF.mDescriptors), ptr(ok1),
            other.N, ptr(_f32(a2)), ptr(other.mDescriptors), ptr(ok2),
            len(nd1), ptr(nd1), ptr(st1), ptr(it1), len(nd2), ptr(nd2), ptr(st2), ptr(it2),
            float(self.mfNNratio), int(max_dist), int(self.mbCheckOrientation), ptr(m12), C.byref(nm)))
        vpMatches.clear()
        if to_frame:
            vpMatches.extend([None] * other.N)
            for i1 in range(pKF.N):
                if m12[i1] >= 0:
                    vpMatches[m12[i1]] = mps1[i1]
        else:
            vpMatches.extend(mps2[m12[i1]] if m12[i1] >= 0 else None for i1 in range(pKF.N))
        return nm.value

    # ---- Fuse / SearchBySim3: independent best-candidate scans + the reference's host logic ----
    def _window_best(self, KF, q, inv_sigma2=None, chi2=0.0):
        k = KF.mvKeysUn
        nq = len(q["active"])
        bi = np.full(max(nq, 1), -1, np.int32)
        bd = np.zeros(max(nq, 1), np.int32)
        inv = _f32(inv_sigma2) if inv_sigma2 is not None else np.ones(1, np.float32)
        b = FrameBounds(KF.mnMinX, KF.mnMaxX, KF.mnMinY, KF.mnMaxY)
        check(self._lib.orb_window_best_keyframe(  # every caller queries a KeyFrame (src/KeyFrame.cc:732-776)
            self.device, b, KF.N, ptr(_f32(k["x"])), ptr(_f32(k["y"])), ptr(_i32(k["octave"])), ptr(KF.mDescriptors),
            ptr(inv), len(inv), float(chi2), nq, ptr(_u8(q["active"])), ptr(_f32(q["u"])), ptr(_f32(q["v"])),
            ptr(_f32(q["r"])), ptr(_i32(q["minl"])), ptr(_u8(q["desc"])), ptr(bi), ptr(bd)))
        return bi, bd

    @staticmethod
    def _project_for_fuse(pKF, Rcw, tcw, Ow, pMP, th, q, i, camera_dist=None):
        """Gates of reference src/ORBmatcher.cc:1089-1131 (= :1255-1299; SearchBySim3 :1438-1476
        measures the distance in the camera frame and has no viewing-angle gate): fills query i,
        returns False if the point is rejected before the candidate scan."""
        f32 = np.float32
        p3Dw = pMP.GetWorldPos()
        p3Dc = _mat3_mul_add(Rcw, p3Dw, tcw) if camera_dist is None else camera_dist
        if p3Dc[2] < 0.0:
            return False
        invz = f32(1.0 / np.float64(p3Dc[2])) if p3Dc[2] != 0 else f32(np.inf)
        u = f32(f32(f32(pKF.fx) * f32(p3Dc[0] * invz)) + f32(pKF.cx))
        v = f32(f32(f32(pKF.fy) * f32(p3Dc[1] * invz)) + f32(pKF.cy))
        if not pKF.IsInImage(u, v):
            return False
        if camera_dist is None:
            PO = (p3Dw - Ow).astype(f32)
        else:
            PO = p3Dc
        dist3D = f32(np.sqrt(np.float64(PO[0]) ** 2 + np.float64(PO[1]) ** 2 + np.float64(PO[2]) ** 2))  # cv::norm
        if dist3D < pMP.GetMinDistanceInvariance() or dist3D > pMP.GetMaxDistanceInvariance():
            return False
        if camera_dist is None:
            Pn = pMP.GetNormal()
            if np.float64(f32(f32(f32(PO[0] * Pn[0]) + f32(PO[1] * Pn[1])) + f32(PO[2] * Pn[2]))) < 0.5 * np.float64(dist3D):
                return False
        n = pMP.PredictScale(dist3D, pKF)
        q["active"][i] = 1
        q["u"][i], q["v"][i] = u, v
        q["r"][i] = f32(f32(th) * pKF.mvScaleFactors[n])
        q["minl"][i] = n  # the predicted level; candidates of level n - 1 and n pass
        q["desc"][i] = pMP.GetDescriptor()
        return True

    def Fuse(self, pKF, *args):
        """Both overloads (include/ORBmatcher.h): Fuse(pKF, vpMapPoints, th=3.0) and
        Fuse(pKF, Scw, vpPoints, th, vpReplacePoint)."""
        if len(args) >= 2 and isinstance(args[0], np.ndarray):
            return self._fuse_sim3(pKF, *args)
        return self._fuse(pKF, *args)

    def _fuse(self, pKF, vpMapPoints, th=3.0):
        # reference src/ORBmatcher.cc:1060-1205.  The scan of a point does not depend on what
        # earlier points did to the map; the gates that do (isBad, IsInKeyFrame) are evaluated
        # again in the loop below, in the reference's order.
        Rcw, tcw, Ow = pKF.GetRotation(), pKF.GetTranslation(), pKF.GetCameraCenter()
        n = len(vpMapPoints)
        q = self._blank_queries(n)
        for i, pMP in enumerate(vpMapPoints):
            if pMP is not None:
                self._project_for_fuse(pKF, Rcw, tcw, Ow, pMP, th, q, i)
        bi, bd = self._window_best(pKF, q, pKF.mvInvLevelSigma2, 5.99)
        nFused = 0
        for i, pMP in enumerate(vpMapPoints):
            if pMP is None or pMP.isBad() or pMP.IsInKeyFrame(pKF) or not q["active"][i]:
                continue
            if bi[i] >= 0 and bd[i] <= self.TH_LOW:
                pMPinKF = pKF.GetMapPoint(int(bi[i]))
                if pMPinKF is not None:
                    if not pMPinKF.isBad():
                        if pMPinKF.Observations() > pMP.Observations():
                            pMP.Replace(pMPinKF)
                        else:
                            pMPinKF.Replace(pMP)
                else:
                    pMP.AddObservation(pKF, int(bi[i]))
                    pKF.AddMapPoint(pMP, int(bi[i]))
                nFused += 1
        return nFused

    def _fuse_sim3(self, pKF, Scw, vpPoints, th, vpReplacePoint):
        # reference src/ORBmatcher.cc:1218-1354
        f32 = np.float32
        Scw = np.asarray(Scw, f32).reshape(4, 4)
        sRcw = Scw[:3, :3]
        scw = f32(np.sqrt(np.float64(f32(f32(sRcw[0, 0] * sRcw[0, 0]) + f32(sRcw[0, 1] * sRcw[0, 1])) + f32(sRcw[0, 2] * sRcw[0, 2]))))
        Rcw = (sRcw / scw).astype(f32)
        tcw = (Scw[:3, 3] / scw).astype(f32)
        Ow = (-(Rcw.T @ tcw)).astype(f32)
        spAlreadyFound = pKF.GetMapPoints()
        q = self._blank_queries(len(vpPoints))
        for i, pMP in enumerate(vpPoints):
            if not pMP.isBad() and pMP not in spAlreadyFound:
                self._project_for_fuse(pKF, Rcw, tcw, Ow, pMP, th, q, i)
        bi, bd = self._window_best(pKF, q)
        nFused = 0
        for i, pMP in enumerate(vpPoints):
            if pMP.isBad() or pMP in spAlreadyFound or not q["active"][i]:
                continue
            if bi[i] >= 0 and bd[i] <= self.TH_LOW:
                pMPinKF = pKF.GetMapPoint(int(bi[i]))
                if pMPinKF is not None:
                    if not pMPinKF.isBad():
                        vpReplacePoint[i] = pMPinKF
                else:
                    pMP.AddObservation(pKF, int(bi[i]))
                    pKF.AddMapPoint(pMP, int(bi[i]))
                nFused += 1
        return nFused

    def SearchBySim3(self, pKF1, pKF2, vpMatches12, s12, R12, t12, th):
        # reference src/ORBmatcher.cc:1368-1615
        f32 = np.float32
        R1w, t1w = pKF1.GetRotation(), pKF1.GetTranslation()
        R2w, t2w = pKF2.GetRotation(), pKF2.GetTranslation()
        R12 = np.asarray(R12, f32).reshape(3, 3)
        t12 = np.asarray(t12, f32).reshape(3)
        sR12 = (f32(s12) * R12).astype(f32)
        sR21 = (np.float64(1.0) / np.float64(f32(s12)) * R12.T.astype(np.float64)).astype(f32)
        t21 = (-_mat3_mul_add(sR21, t12, np.zeros(3, f32))).astype(f32)
        mps1, mps2 = pKF1.GetMapPointMatches(), pKF2.GetMapPointMatches()
        N1, N2 = len(mps1), len(mps2)
        done1, done2 = [False] * N1, [False] * N2
        for i in range(N1):
            pMP = vpMatches12[i]
            if pMP is not None:
                done1[i] = True
                idx2 = pMP.GetIndexInKeyFrame(pKF2)
                if 0 <= idx2 < N2:
                    done2[idx2] = True

        def one_way(mps, done, Rw, tw, sR, t, dst):
            q = self._blank_queries(len(mps))
            for i, pMP in enumerate(mps):
                if pMP is None or done[i] or pMP.isBad():
                    continue
                pc = _mat3_mul_add(Rw, pMP.GetWorldPos(), tw)
                pd = _mat3_mul_add(sR, pc, t)
                self._project_for_fuse(dst, None, None, None, pMP, th, q, i, camera_dist=pd)
            bi, bd = self._window_best(dst, q)
            return [int(bi[i]) if q["active"][i] and bi[i] >= 0 and bd[i] <= self.TH_HIGH else -1 for i in range(len(mps))]

        vnMatch1 = one_way(mps1, done1, R1w, t1w, sR21, t21, pKF2)
        vnMatch2 = one_way(mps2, done2, R2w, t2w, sR12, t12, pKF1)
        nFound = 0
        for i1 in range(N1):
            idx2 = vnMatch1[i1]
            if idx2 >= 0 and vnMatch2[idx2] == i1:
                vpMatches12[i1] = mps2[idx2]
                nFound += 1
        return nFound

    def SearchForTriangulation(self, pKF1, pKF2, F12, vMatchedPairs, bOnlyStereo=False):
        # epipole of KF1's centre in KF2, reference src/ORBmatcher.cc:862-871
        Cw = pKF1.GetCameraCenter()
        C2 = _mat3_mul_add(pKF2.GetRotation(), Cw, pKF2.GetTranslation())
        invz = np.float32(1.0) / C2[2]
        ex = np.float32(np.float32(np.float32(np.float32(pKF2.fx) * C2[0]) * invz) + np.float32(pKF2.cx))
        ey = np.float32(np.float32(np.float32(np.float32(pKF2.fy) * C2[1]) * invz) + np.float32(pKF2.cy))

        nd1, st1, it1 = _flat_featvec(pKF1.mFeatVec)
        nd2, st2, it2 = _flat_featvec(pKF2.mFeatVec)
        k1, k2 = pKF1.mvKeysUn, pKF2.mvKeysUn
        has1 = np.array([mp is not None for mp in pKF1.mvpMapPoints] or [0], np.uint8)
        has2 = np.array([mp is not None for mp in pKF2.mvpMapPoints] or [0], np.uint8)
        m12 = np.full(max(pKF1.N, 1), -1, np.int32)
        nm = C.c_int(0)
        sf2, ls2 = _f32(pKF2.mvScaleFactors), _f32(pKF2.mvLevelSigma2)
        F12 = _f32(F12).reshape(9)
        check(self._lib.orb_search_for_triangulation(
            self.device, pKF1.N, ptr(_f32(k1["x"])), ptr(_f32(k1["y"])), ptr(_i32(k1["octave"])), ptr(_f32(k1["angle"])),
            ptr(pKF1.mDescriptors), ptr(has1),
            pKF2.N, ptr(_f32(k2["x"])), ptr(_f32(k2["y"])), ptr(_i32(k2["octave"])), ptr(_f32(k2["angle"])),
            ptr(pKF2.mDescriptors), ptr(has2),
            len(nd1), ptr(nd1), ptr(st1), ptr(it1), len(nd2), ptr(nd2), ptr(st2), ptr(it2),
            ptr(F12), float(ex), float(ey), ptr(sf2), ptr(ls2), len(sf2), int(self.mbCheckOrientation),
            ptr(m12), C.byref(nm)))
        vMatchedPairs.clear()
        vMatchedPairs.extend((i, int(m12[i])) for i in range(pKF1.N) if m12[i] >= 0)
        return nm.value


def _flat_featvec(fv):
    """DBoW2::FeatureVector (ordered map node id -> feature indices) as sorted node ids + CSR."""
    nodes = np.array(list(fv.keys()), np.int32)
    start = np.zeros(len(nodes) + 1, np.int32)
    items = []
    for k, v in enumerate(fv.values()):
        items.extend(v)
        start[k + 1] = len(items)
    return nodes, start, np.array(items, np.int32) if items else np.zeros(1, np.int32)


def stereo_match(rows, left, right, scale_factors, min_d, max_d, th_dist=75, device=0):
    """Coarse stereo association (row-band Hamming search; this fork of the reference has no
    stereo matcher, see include/orbhip.h).  left / right: (keypoints KP_DTYPE, descriptors).
    Returns (n_matches, best_idx[int32, -1 = none], best_dist[int32])."""
    lk, ld = left
    rk, rd = right
    nl, nr = len(lk), len(rk)
    sf = _f32(scale_factors)
    bi = np.full(max(nl, 1), -1, np.int32)
    bd = np.zeros(max(nl, 1), np.int32)
    nm = C.c_int(0)
    check(_lib.lib().orb_stereo_match(
        device, int(rows), nl, ptr(_f32(lk["x"])), ptr(_f32(lk["y"])), ptr(_i32(lk["octave"])), ptr(_u8(ld)), nr,
        ptr(_f32(rk["x"])), ptr(_f32(rk["y"])), ptr(_i32(rk["octave"])), ptr(_u8(rd)), ptr(sf), len(sf),
        float(min_d), float(max_d), int(th_dist), ptr(bi), ptr(bd), C.byref(nm)))
    return nm.value, bi[:nl], bd[:nl]


def knn2_bruteforce(q, db, ratio=0.7, device=0):
    """Brute-force Hamming 2-nearest-neighbour search with ratio test on one GPU.
    Returns (best, second, idx, accept)."""
    q, db = _u8(q).reshape(-1, 32), _u8(db).reshape(-1, 32)
    nq = len(q)
    best, second, idx = (np.zeros(max(nq, 1), np.int32) for _ in range(3))
    acc = np.zeros(max(nq, 1), np.uint8)
    check(_lib.lib().orb_knn2_bruteforce(int(device), ptr(q), nq, ptr(db), len(db), float(ratio),
                                         ptr(best), ptr(second), ptr(idx), ptr(acc)))
    return best[:nq], second[:nq], idx[:nq], acc[:nq]
