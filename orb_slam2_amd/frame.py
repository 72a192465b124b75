"""Minimal host-side stand-ins for the reference's Frame / KeyFrame / MapPoint: just the
state the hot matchers read and write (reference include/Frame.h, KeyFrame.h, MapPoint.h).
They exist so that ORBmatcher below keeps the reference's call signatures."""
import numpy as np


class MapPoint:
    def __init__(self, world_pos, descriptor, observations=1):
        self._pos = np.asarray(world_pos, np.float32).reshape(3)
        self._desc = np.asarray(descriptor, np.uint8).reshape(32)
        self._nobs = int(observations)
        self._bad = False
        self.mObservations = {}  # keyframe -> feature index (reference include/MapPoint.h)
        self.mpReplaced = None
        # filled by Frame::isInFrustum in the reference (src/Frame.cc:343-355)
        self.mbNeedTrackInView = False
        self.mnTrackedScaleLevel = 0
        self.mTrackedViewCosine = 1.0
        self.mTrackedProjX = 0.0
        self.mTrackedProjY = 0.0

    def GetWorldPos(self):
        return self._pos

    def GetDescriptor(self):
        return self._desc

    def Observations(self):
        return self._nobs

    # observation bookkeeping used by Fuse / SearchBySim3 (reference src/MapPoint.cc:137-175,
    # 276-334, 471-492); the monocular subset: one observation counts once
    def AddObservation(self, pKF, idx):
        if pKF in self.mObservations:
            return
        self.mObservations[pKF] = int(idx)
        self._nobs += 1

    def IsInKeyFrame(self, pKF):
        return pKF in self.mObservations

    def GetIndexInKeyFrame(self, pKF):
        return self.mObservations.get(pKF, -1)

    def Replace(self, pMP):
        if pMP is self:
            return
        obs, self.mObservations = self.mObservations, {}
        self._bad = True
        self.mpReplaced = pMP
        for pKF, idx in obs.items():
            if not pMP.IsInKeyFrame(pKF):
                pKF.ReplaceMapPointMatch(idx, pMP)
                pMP.AddObservation(pKF, idx)
            else:
                pKF.EraseMapPointMatch(idx)

    def isBad(self):
        return self._bad

    def ComputeDistinctiveDescriptors(self):
        """reference src/MapPoint.cc:386-469; see compute_distinctive_descriptors for many points"""
        compute_distinctive_descriptors([self])

    # scale-invariance range and viewing direction (reference src/MapPoint.cc:527-568);
    # set by UpdateNormalAndDepth in the reference, plain attributes here
    mfMinDistance = 0.0
    mfMaxDistance = float("inf")
    mNormalVector = np.array([0, 0, 1], np.float32)

    def GetMinDistanceInvariance(self):
        return np.float32(np.float32(0.8) * np.float32(self.mfMinDistance))

    def GetMaxDistanceInvariance(self):
        return np.float32(np.float32(1.2) * np.float32(self.mfMaxDistance))

    def GetNormal(self):
        return np.asarray(self.mNormalVector, np.float32)

    def PredictScale(self, currentDist, frame):
        """reference src/MapPoint.cc:593-633"""
        ratio = np.float32(np.float32(self.mfMaxDistance) / np.float32(currentDist))
        nScale = int(np.ceil(np.float32(np.log(ratio)) / np.float32(frame.mfLogScaleFactor)))
        return min(max(nScale, 0), frame.mnScaleLevels - 1)


def compute_distinctive_descriptors(map_points, device_id=0):
    """MapPoint::ComputeDistinctiveDescriptors (reference src/MapPoint.cc:386-469) for a list of
    map points in one GPU call: each point that is not bad takes, of the descriptors of the
    keyframes (not bad) that observe it, the one with the least median distance to the others.
    Observations are visited in insertion order (the reference's std::map orders them by
    keyframe address, which decides nothing but ties between equal medians)."""
    from . import _lib
    rows, start, todo = [], [0], []
    for mp in map_points:
        if mp.isBad():
            continue
        obs = [pKF.mDescriptors[idx] for pKF, idx in mp.mObservations.items() if not pKF.isBad()]
        if not obs:
            continue
        rows.extend(obs)
        start.append(len(rows))
        todo.append(mp)
    if not todo:
        return
    desc = np.ascontiguousarray(np.stack(rows), np.uint8)
    start = np.asarray(start, np.int32)
    best = np.empty(len(todo), np.int32)
    _lib.check(_lib.lib().orb_distinctive_descriptors(int(device_id), len(todo), _lib.ptr(start), _lib.ptr(desc),
                                                      _lib.ptr(best), None))
    for mp, s0, b in zip(todo, start[:-1], best):
        mp._desc = desc[s0 + b].copy()


class Frame:
    """Keypoints, descriptors, pose and map-point slots of one image."""

    def __init__(self, keypoints, descriptors, scale_factors, level_sigma2, K, bounds, Tcw=None, feat_vec=None):
        """keypoints: structured array with x, y, octave, angle (undistorted = raw: the
        reference's UndistortKeyPoints is the identity for zero distortion, src/Frame.cc:506-510);
        K = (fx, fy, cx, cy); bounds = (mnMinX, mnMaxX, mnMinY, mnMaxY)."""
        self.mvKeys = keypoints
        self.mvKeysUn = keypoints
        self.N = len(keypoints)
        self.mDescriptors = np.ascontiguousarray(descriptors, np.uint8).reshape(self.N, 32)
        self.mvScaleFactors = np.asarray(scale_factors, np.float32)
        self.mvLevelSigma2 = np.asarray(level_sigma2, np.float32)
        self.fx, self.fy, self.cx, self.cy = (float(v) for v in K)
        self.mnMinX, self.mnMaxX, self.mnMinY, self.mnMaxY = (float(v) for v in bounds)
        self.mTcw = np.eye(4, dtype=np.float32) if Tcw is None else np.asarray(Tcw, np.float32).reshape(4, 4)
        self.mvpMapPoints = [None] * self.N
        # DBoW2::FeatureVector: vocabulary node id -> feature indices, ordered by node id
        self.mFeatVec = dict(sorted((feat_vec or {}).items()))
        self.mBowVec = {}
        self.mnScaleLevels = len(self.mvScaleFactors)
        self.mfScaleFactor = float(self.mvScaleFactors[1]) if self.mnScaleLevels > 1 else 1.2
        self.mfLogScaleFactor = np.float32(np.log(np.float32(self.mfScaleFactor)))

    @classmethod
    def from_raw(cls, keypoints, descriptors, scale_factors, level_sigma2, K, dist_coef, rows, cols, device_id=0,
                 **kw):
        """The part of the reference's Frame constructor between ExtractORB and
        AssignFeaturesToGrid (src/Frame.cc:502-596): mvKeys stay as extracted, mvKeysUn are the
        undistorted copies, the bounds come from the undistorted image corners."""
        from . import _lib
        d = np.zeros(5, np.float32)
        d[:len(dist_coef)] = dist_coef
        cam = _lib.Camera(*(float(v) for v in K), *(float(v) for v in d))
        b = _lib.FrameBounds()
        _lib.check(_lib.lib().orb_compute_image_bounds(int(rows), int(cols), cam, b))
        f = cls(keypoints, descriptors, scale_factors, level_sigma2, K, (b.min_x, b.max_x, b.min_y, b.max_y), **kw)
        f.mDistCoef = d
        if d[0] != 0 and f.N:
            xy = np.ascontiguousarray(np.stack([keypoints["x"], keypoints["y"]], 1), np.float32)
            un = np.empty_like(xy)
            _lib.check(_lib.lib().orb_undistort_points(int(device_id), f.N, _lib.ptr(xy), cam, _lib.ptr(un)))
            f.mvKeysUn = keypoints.copy()
            f.mvKeysUn["x"], f.mvKeysUn["y"] = un[:, 0], un[:, 1]
        return f

    def ComputeBoW(self, vocabulary):
        """reference src/Frame.cc:483-497 / src/KeyFrame.cc:120-131: once per frame, levelsup 4"""
        if not self.mBowVec:
            self.mBowVec, self.mFeatVec = vocabulary.transform(self.mDescriptors, 4)


class KeyFrame(Frame):
    """Adds the accessors the keyframe searches use (map point list, pose, image bounds)."""

    mbBad = False

    def isBad(self):
        return self.mbBad

    def GetMapPoint(self, idx):
        return self.mvpMapPoints[idx]

    def GetMapPointMatches(self):
        return list(self.mvpMapPoints)

    def GetMapPoints(self):
        """reference src/KeyFrame.cc: the set of map points that are not bad"""
        return {mp for mp in self.mvpMapPoints if mp is not None and not mp.isBad()}

    def AddMapPoint(self, pMP, idx):
        self.mvpMapPoints[idx] = pMP

    def ReplaceMapPointMatch(self, idx, pMP):
        self.mvpMapPoints[idx] = pMP

    def EraseMapPointMatch(self, idx):
        self.mvpMapPoints[idx] = None

    @property
    def mvInvLevelSigma2(self):
        return (np.float32(1.0) / self.mvLevelSigma2).astype(np.float32)

    def IsInImage(self, x, y):
        """reference src/KeyFrame.cc:779-782; a KeyFrame keeps the frame's bounds as const int
        (include/KeyFrame.h:411-414), i.e. truncated towards zero - only different from the
        frame's own bounds when the camera has distortion"""
        return int(self.mnMinX) <= x < int(self.mnMaxX) and int(self.mnMinY) <= y < int(self.mnMaxY)

    def GetRotation(self):
        return self.mTcw[:3, :3]

    def GetTranslation(self):
        return self.mTcw[:3, 3]

    def GetCameraCenter(self):
        R, t = self.mTcw[:3, :3], self.mTcw[:3, 3]
        return (-(R.T.astype(np.float32) @ t)).astype(np.float32)
