/*
 * orbhip.h -- C ABI of liborbhip.so: ORB extraction and Hamming matching on AMD
 * MI355X (gfx950).  Plain C types, caller-owned buffers, int status codes, opaque
 * handles; no exceptions cross this boundary.
 *
 * Each entry point names the ORB-SLAM2 interface it stands in for
 * (zhangshaos/ORB_SLAM2, file:line).  Pointers marked "device" are HIP device
 * pointers on the handle's GPU; everything else is host memory.
 *
 * Threading: an extractor handle serves one caller thread and one stream at a time (like an
 * ORBextractor instance, which keeps its pyramid between calls): its pyramid, candidate and
 * keypoint buffers are shared by every call on it, so two calls that overlap (two threads, or
 * orb_extract_batch_device on two streams without an event between them) race.  Distinct handles
 * are independent.  The first call of a handle at a new image size, or with more frames than any
 * call before, allocates and waits for the handle's own earlier work; later calls only enqueue.
 * The host-buffer matcher entry points (and orb_distinctive_descriptors, orb_bow_transform,
 * orb_undistort_points) keep no state and may be called concurrently from several threads (the
 * reference calls them from its tracking, mapping and loop-closing threads): each call runs on a
 * stream of its own and waits only for that stream.  Device memory of handles and calls comes from
 * the device's stream-ordered pool (hipMallocAsync / hipFreeAsync on the handle's or call's own
 * stream), so neither a first call nor a growing one waits for other threads' work; on a device
 * without memory pools (or with ORBHIP_ASYNC_ALLOC=0 in the environment) plain hipMalloc / hipFree
 * are used instead, and a first or growing call then waits for the whole device.
 */
#ifndef ORBHIP_H_
#define ORBHIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#define ORBHIP_OK 0
#define ORBHIP_ERR_ARG (-1)        /* bad argument (null pointer, size out of range) */
#define ORBHIP_ERR_UNSUPPORTED (-2) /* input for which the reference's behaviour is undefined,
                                       or a size beyond this build's limits */
#define ORBHIP_ERR_HIP (-3)        /* a HIP runtime call failed; see orb_last_error() */
#define ORBHIP_ERR_CAPACITY (-4)   /* a caller-provided buffer was too small */

/* Same fields as cv::KeyPoint as filled by ORBextractor::operator()
 * (src/ORBextractor.cc:1164-1179,1323-1337): pt in level-0 pixels, size =
 * int(31*scale[octave]), angle in degrees [0,360), response = FAST score. */
typedef struct OrbKeyPoint {
  float x, y;
  float size;
  float angle;
  float response;
  int32_t octave;
} OrbKeyPoint;

typedef struct OrbExtractor OrbExtractor;

/* Text of the last error on the calling thread ("" if none). */
const char* orb_last_error(void);

/* Number of HIP devices visible, or a negative status. */
int orb_device_count(void);

/* ---- extractor ------------------------------------------------------------ */

/* ORBextractor::ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST)
 * (include/ORBextractor.h:95, src/ORBextractor.cc:471-573) on GPU `device_id`. */
int orb_extractor_create(int nfeatures, float scale_factor, int nlevels, int ini_th_fast,
                         int min_th_fast, int device_id, OrbExtractor** out);
void orb_extractor_destroy(OrbExtractor* h);

/* GetLevels / GetScaleFactors / GetInverseScaleFactors / GetScaleSigmaSquares /
 * GetInverseScaleSigmaSquares (include/ORBextractor.h:121-163).  Each array holds
 * nlevels floats; null pointers are skipped.  Returns nlevels. */
int orb_extractor_get_scale_tables(const OrbExtractor* h, float* scale, float* inv_scale,
                                   float* sigma2, float* inv_sigma2);
/* mnFeaturesPerLevel (src/ORBextractor.cc:511-528) and umax[0..15] (:544-572). */
int orb_extractor_get_level_quotas(const OrbExtractor* h, int* quotas /* nlevels */, int* umax /* 16 */);
/* Upper bound on keypoints per frame for the given image size (the quad-tree may
 * return a few more than nfeatures, src/ORBextractor.cc:985).  <0 on error. */
int orb_extractor_max_keypoints(OrbExtractor* h, int rows, int cols);

/* ORBextractor::operator()(image, mask, keypoints, descriptors)
 * (src/ORBextractor.cc:1223-1339) for one host image (8-bit, one channel,
 * `stride` bytes per row).  The mask is ignored by the reference and has no
 * parameter here.  Writes at most max_kp keypoints and max_kp*32 descriptor bytes;
 * *n_out receives the number found (ORBHIP_ERR_CAPACITY if it exceeds max_kp).
 * An empty image (rows or cols == 0) returns ORBHIP_OK with *n_out = 0. */
int orb_extract(OrbExtractor* h, const uint8_t* image, int rows, int cols, size_t stride,
                int max_kp, OrbKeyPoint* keypoints, uint8_t* descriptors, int* n_out);

/* The same for n_frames host images of one size; frame k starts at
 * images + k*frame_stride.  keypoints/descriptors/n_out are [n_frames][max_kp],
 * [n_frames][max_kp][32], [n_frames]. */
int orb_extract_batch(OrbExtractor* h, int n_frames, const uint8_t* images, int rows, int cols,
                      size_t stride, size_t frame_stride, int max_kp, OrbKeyPoint* keypoints,
                      uint8_t* descriptors, int* n_out);

/* The same with every buffer resident on the handle's GPU.  The work is enqueued on
 * `stream` (a hipStream_t; null = the device's default stream) and not waited for; the
 * call allocates only when the image size or n_frames grows beyond what the handle
 * has already seen.  d_n_out[k] is the count found for frame k (may exceed max_kp,
 * in which case only max_kp entries were written). */
int orb_extract_batch_device(OrbExtractor* h, int n_frames, const uint8_t* d_images, int rows,
                             int cols, size_t stride, size_t frame_stride, int max_kp,
                             OrbKeyPoint* d_keypoints, uint8_t* d_descriptors, int* d_n_out,
                             void* stream);

/* Pixel format of the images handed to the orb_extract* calls that follow: 1 = 8-bit grey
 * (default, what ORBextractor::operator() asserts, src/ORBextractor.cc:1229), 3 or 4 =
 * interleaved colour, converted to grey while level 0 of the pyramid is written - the
 * cvtColor(CV_RGB2GRAY / CV_BGR2GRAY / CV_RGBA2GRAY / CV_BGRA2GRAY) that
 * Tracking::GrabImageMonocular / GrabImageStereo / GrabImageRGBD run first
 * (src/Tracking.cc:203-216); rgb_order = the reference's mbRGB (Camera.RGB).  `stride` and
 * `frame_stride` stay in bytes (stride >= cols * channels).  OpenCV's 8-bit conversion is
 * (R*9798 + G*19235 + B*3735 + 2^14) >> 15; alpha is ignored. */
int orb_extractor_set_input_format(OrbExtractor* h, int channels, int rgb_order);

/* The OpenCV release whose arithmetic the handle reproduces from now on.  The reference links
 * whatever OpenCV its build finds (CMakeLists.txt:20, no version pinned), and two primitives on
 * this path changed between releases:
 *   cv::GaussianBlur 7x7 sigma 2, 8-bit (src/ORBextractor.cc:1300): fixed-point taps
 *     18 34 48 56 48 34 18 (sum 256) from 4.5 on; 18 34 49 55 49 34 18 (every tap rounded, sum 257,
 *     results saturated) in 3.4 and 4.0 - 4.4;
 *   cv::cvtColor to grey (src/Tracking.cc:203-216, see orb_extractor_set_input_format): 15-bit
 *     weights from 4.0 on, 14-bit weights (R*4899 + G*9617 + B*1868 + 2^13) >> 14 before.
 * Default: 4.5 and later.  Releases before 3.4 blur in floating point and are refused with
 * ORBHIP_ERR_UNSUPPORTED.  No counterpart in the reference.  The release boundaries above are
 * taken from OpenCV's history as restated here, not from anything the reference holds: they are
 * unverified against a real OpenCV build (a late 3.4.x may already carry the 4.5 taps). */
int orb_extractor_set_opencv_compat(OrbExtractor* h, int major, int minor);

/* Where the 7x7 blur of src/ORBextractor.cc:1297-1305 is done.  fused != 0 (the default): the
 * descriptor kernel blurs the 37 x 37 patch around each keypoint itself, in on-chip memory, and no
 * blurred level is written; fused == 0: every level is blurred to memory first, as the reference
 * does.  The descriptors are bit-identical either way (the same fixed-point arithmetic on the same
 * pixels).  No counterpart in the reference. */
int orb_extractor_set_blur_fusion(OrbExtractor* h, int fused);

/* orb_extract_batch_device with 64 frames or more runs the batch as two halves: the first on the
 * caller's stream, the second on a stream of the handle's own, started one stage behind (after the
 * first half's pyramid) and joined back into the caller's stream before the call returns, so that
 * the bandwidth-bound pyramid of one half runs beside the issue-bound stages of the other.  Same
 * results, same frame slots (every accessor works as before); the caller's stream sees the whole
 * call as one piece of work.  enable == 0 keeps the batch on the caller's stream alone; with stage
 * timing on, and on a stream under graph capture, it is kept there anyway.  Default: enabled.  No
 * counterpart in the reference. */
int orb_extractor_set_batch_split(OrbExtractor* h, int enable);

/* Waits for everything the handle has queued: its own streams (which the host-buffer entry
 * points and the accessors below use) and the last orb_extract_batch_device call on the caller's
 * stream. */
int orb_extractor_sync(OrbExtractor* h);

/* Stage timing of the batch calls (no counterpart in the reference, which has no timers
 * on this path).  While enabled, every batch call records HIP events on its stream
 * between the stages; orb_extractor_stage_times waits for those calls and returns the
 * milliseconds each stage took, averaged over the calls since the times were last
 * read, in the order {pyramid, FAST, quad-tree, blur, orientation+descriptors}.
 * Returns the stage count. */
#define ORBHIP_NUM_STAGES 5
int orb_extractor_set_profiling(OrbExtractor* h, int enable);
int orb_extractor_stage_times(OrbExtractor* h, float* ms /* ORBHIP_NUM_STAGES */);

/* mvImagePyramid[level] of the last call (include/ORBextractor.h:166), frame `frame`
 * of the batch.  padded != 0 returns the (h+38) x (w+38) image including the 19-px
 * BORDER_REFLECT_101 frame (src/ORBextractor.cc:1385-1408), else the h x w level.
 * dst may be null to query the size only. */
int orb_get_pyramid_level(OrbExtractor* h, int frame, int level, int padded, uint8_t* dst,
                          size_t dst_stride, int* width, int* height);
/* The 7x7 sigma-2 blurred copy of a level that the descriptors are sampled from
 * (src/ORBextractor.cc:1297-1305).  With blur fusion on, the levels of the last call's frames are
 * blurred when this is first called after that call. */
int orb_get_blurred_level(OrbExtractor* h, int frame, int level, uint8_t* dst, size_t dst_stride,
                          int* width, int* height);
/* vToDistributeKeys of a level (src/ORBextractor.cc:1061-1146): FAST corners after
 * cell-local non-maximum suppression, as (x, y, response) with x, y relative to the
 * (16,16) border origin, in the reference's cell-row-major / raster order.
 * Returns the count (writes at most max_n triplets). */
int orb_get_fast_candidates(OrbExtractor* h, int frame, int level, int max_n, float* xyr);
/* Keypoints of one level after quad-tree distribution (src/ORBextractor.cc:1156-1179),
 * level coordinates, in the reference's list order: (x, y, response) triplets. */
int orb_get_level_keypoints(OrbExtractor* h, int frame, int level, int max_n, float* xyr);

/* ---- matcher ---------------------------------------------------------------- */

/* ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:1991-2011) for n pairs:
 * out[i] = popcount(a[i] ^ b[i]) over 256 bits.  Host buffers; the pairs are
 * evaluated on the GPU `device_id`. */
int orb_hamming_distance_pairs(int device_id, const uint8_t* a, const uint8_t* b, int n, int* out);

/* Window geometry shared by the projection matchers: Frame::mnMinX.. and the 64x48
 * grid of Frame::AssignFeaturesToGrid / GetFeaturesInArea (src/Frame.cc:190-213,
 * 371-449, 461-477). */
typedef struct OrbFrameBounds {
  float min_x, max_x, min_y, max_y;
} OrbFrameBounds;

/* Pinhole camera with the 4- or 5-coefficient distortion model of the settings file
 * (Frame::mK, Frame::mDistCoef: k1, k2, p1, p2 and optionally k3). */
typedef struct OrbCamera {
  float fx, fy, cx, cy;
  float k1, k2, p1, p2, k3;
} OrbCamera;

/* Frame::UndistortKeyPoints (src/Frame.cc:502-553) for a batch of frames in the layout the
 * extractor writes: d_kps_un[f][i] = d_kps[f][i] with pt replaced by
 * cv::undistortPoints(pt, K, distCoef, R = I, P = K) (five iterations in double, result rounded
 * to float), for i < min(d_n[f], max_kp); a plain copy if k1 == 0, as in the reference.  Device
 * pointers (d_kps_un may equal d_kps); enqueued on `stream`, not waited for. */
int orb_undistort_keypoints_device(int device_id, int n_frames, int max_kp, const OrbKeyPoint* d_kps,
                                   const int* d_n, OrbCamera cam, OrbKeyPoint* d_kps_un, void* stream);
/* The same map for n host points (x, y interleaved). */
int orb_undistort_points(int device_id, int n, const float* xy, OrbCamera cam, float* xy_un);
/* Frame::ComputeImageBounds (src/Frame.cc:560-596): mnMinX, mnMaxX, mnMinY, mnMaxY from the
 * undistorted image corners (the image size itself if k1 == 0).  Host arithmetic only. */
int orb_compute_image_bounds(int rows, int cols, OrbCamera cam, OrbFrameBounds* out);

/* ORBmatcher::SearchByProjection(Frame& Cur, const Frame& Last, th, bMono)
 * (src/ORBmatcher.cc:1633-1788), for n_pairs independent (Last, Cur) frame pairs
 * with the same intrinsics; pair p uses slices [p*max_last, ...) / [p*max_cur, ...)
 * of the arrays.  Host buffers.
 *   tcw        [n_pairs][16]  Cur.mTcw, row major
 *   n_last     [n_pairs]      LastFrame.N
 *   last_mp_obs[..max_last]   -1 where LastFrame.mvpMapPoints[i] is null, else its Observations()
 *   last_xw    [..max_last][3] GetWorldPos()
 *   last_mp_desc[..][32]      GetDescriptor()
 *   last_octave, last_angle   LastFrame.mvKeys[i].octave, mvKeysUn[i].angle
 *   n_cur, cur_x, cur_y, cur_octave, cur_angle, cur_desc   Cur.mvKeysUn / mDescriptors
 *   cur_obs    [..max_cur]    -1 where Cur.mvpMapPoints[i2] is null, else its Observations()
 *   scale_factors [nlevels]   Cur.mvScaleFactors
 * Outputs:
 *   cur_match  [..max_cur]    -1 untouched; i >= 0: Cur.mvpMapPoints[i2] = LastFrame.mvpMapPoints[i];
 *                             -2: set and then nulled by the rotation-consistency check
 *   n_matches  [n_pairs]      the function's return value
 */
int orb_search_by_projection_frames(
    int device_id, int n_pairs, int max_last, int max_cur, const float* tcw, float fx, float fy,
    float cx, float cy, OrbFrameBounds bounds, const float* scale_factors, int nlevels,
    const int* n_last, const int* last_mp_obs, const float* last_xw, const uint8_t* last_mp_desc,
    const int* last_octave, const float* last_angle, const int* n_cur, const float* cur_x,
    const float* cur_y, const int* cur_octave, const float* cur_angle, const uint8_t* cur_desc,
    const int* cur_obs, float th, int check_orientation, int* cur_match, int* n_matches);

/* The same search with every array resident on GPU `device_id`, keypoints given as the
 * OrbKeyPoint records the extractor writes (only x, y, octave, angle are read; the
 * last-frame records supply mvKeys[i].octave and mvKeysUn[i].angle).  Pair p uses
 * d_last_* + p*max_last and d_cur_* + p*max_cur.  scale_factors is a HOST array.
 * d_scratch must hold orb_match_scratch_bytes(n_pairs, max_last) bytes.  Enqueued on
 * `stream`, not waited for; allocates nothing.
 * Size limits of the matchers: up to about 7900 train keypoints per frame (current-frame
 * keypoints here; about 6500 for orb_search_for_initialization) the keypoints' grid lives in
 * on-chip memory.  Larger frames, up to 65535 keypoints, take a second form of the kernel that
 * keeps part of it in global memory: the host-buffer entry points do that by themselves, the
 * device form needs d_scratch sized with orb_match_scratch_bytes_for(n_pairs, max_last, max_cur)
 * (equal to orb_match_scratch_bytes while max_cur fits on chip) and returns ORBHIP_ERR_CAPACITY
 * otherwise; the allocation-free form below is limited to frames that fit on chip. */
size_t orb_match_scratch_bytes(int n_pairs, int max_queries);
size_t orb_match_scratch_bytes_for(int n_pairs, int max_last, int max_cur);
int orb_search_by_projection_frames_device(
    int device_id, int n_pairs, int max_last, int max_cur, const float* d_tcw, float fx, float fy,
    float cx, float cy, OrbFrameBounds bounds, const float* scale_factors, int nlevels,
    const int* d_n_last, const OrbKeyPoint* d_last_kps, const int* d_last_mp_obs,
    const float* d_last_xw, const uint8_t* d_last_mp_desc, const int* d_n_cur,
    const OrbKeyPoint* d_cur_kps, const uint8_t* d_cur_desc, const int* d_cur_obs, float th,
    int check_orientation, void* d_scratch, size_t scratch_bytes, int* d_cur_match,
    int* d_n_matches, void* stream);

/* The same search as orb_search_by_projection_frames_device for callers that must not
 * allocate and have no workspace to hand over (code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  The per-query work
 * arrays live in a fixed array inside the library, ORBHIP_NOALLOC_MAX_QUERIES * 60 + 64 bytes
 * (3.75 MiB) on every GPU the library is used on, held until the process ends; the call fails
 * with ORBHIP_ERR_CAPACITY if n_pairs * max_last exceeds ORBHIP_NOALLOC_MAX_QUERIES.  It calls no
 * hipMalloc / hipFree; the first call on a device looks the array's address up once
 * (hipGetSymbolAddress) and raises the kernel's dynamic-LDS limit (hipFuncSetAttribute): make one
 * call per device outside stream capture before capturing one.  All calls
 * on one GPU share that array: calls that overlap in time on one GPU (different streams or
 * threads) race and must be serialised by the caller.  Inputs, outputs and results as in
 * orb_search_by_projection_frames_device. */
#define ORBHIP_NOALLOC_MAX_QUERIES 65536
int orb_search_by_projection_frames_noalloc(
    int device_id, int n_pairs, int max_last, int max_cur, const float* d_tcw, float fx, float fy,
    float cx, float cy, OrbFrameBounds bounds, const float* scale_factors, int nlevels,
    const int* d_n_last, const OrbKeyPoint* d_last_kps, const int* d_last_mp_obs,
    const float* d_last_xw, const uint8_t* d_last_mp_desc, const int* d_n_cur,
    const OrbKeyPoint* d_cur_kps, const uint8_t* d_cur_desc, const int* d_cur_obs, float th,
    int check_orientation, int* d_cur_match, int* d_n_matches, void* stream);

/* ORBmatcher::SearchByProjection(Frame& F, const std::vector<MapPoint*>&, th)
 * (src/ORBmatcher.cc:64-158) for one frame.  Per map point: mp_track =
 * mbNeedTrackInView && !isBad(), mnTrackedScaleLevel, mTrackedViewCosine,
 * mTrackedProjX/Y, GetDescriptor(), Observations().  f_obs as cur_obs above.
 * f_match[idx] = index of the map point stored in F.mvpMapPoints[idx], or -1. */
int orb_search_by_projection_mappoints(
    int device_id, OrbFrameBounds bounds, const float* scale_factors, int nlevels, int n_f,
    const float* f_x, const float* f_y, const int* f_octave, const uint8_t* f_desc,
    const int* f_obs, int n_mp, const uint8_t* mp_track, const int* mp_level,
    const float* mp_viewcos, const float* mp_projx, const float* mp_projy, const uint8_t* mp_desc,
    const int* mp_obs, float th, float nnratio, int* f_match, int* n_matches);

/* The candidate scan shared by the two remaining projection searches,
 *   SearchByProjection(Frame& Cur, KeyFrame*, const std::set<MapPoint*>&, th, ORBdist)
 *     (src/ORBmatcher.cc:1800-1932: relocalisation; levels n-1..n+1, threshold ORBdist,
 *      rotation check), and
 *   SearchByProjection(KeyFrame*, cv::Mat Scw, const std::vector<MapPoint*>&, std::vector<MapPoint*>&, th)
 *     (:394-524: loop closing; levels n-1..n, threshold TH_LOW, no rotation check).
 * The caller projects its map points (pose algebra, distance gates and
 * MapPoint::PredictScale, src/MapPoint.cc:593-633, stay on the host) and passes, per
 * query in the reference's loop order, the window (u, v, radius), the level range of
 * GetFeaturesInArea, the descriptor and q_active = "reached the candidate scan".
 * t_taken marks train keypoints that already hold a map point; every accepted match
 * (best distance <= max_dist, first minimum wins) takes its keypoint for later queries.
 * t_match[idx] = query index, -1 untouched, -2 set and then nulled by the rotation check. */
int orb_match_windowed(int device_id, OrbFrameBounds bounds, int n_t, const float* t_x,
                       const float* t_y, const int* t_octave, const float* t_angle,
                       const uint8_t* t_desc, const uint8_t* t_taken, int n_q,
                       const uint8_t* q_active, const float* q_u, const float* q_v,
                       const float* q_radius, const int* q_min_level, const int* q_max_level,
                       const float* q_angle, const uint8_t* q_desc, int max_dist,
                       int check_orientation, int* t_match, int* n_matches);

/* orb_match_windowed with the keypoints held by a KeyFrame instead of a Frame.
 * KeyFrame::GetFeaturesInArea (src/KeyFrame.cc:732-776) walks the grid that the Frame built with
 * its float bounds (src/KeyFrame.cc:45-60), but subtracts the KeyFrame's own mnMinX / mnMinY,
 * which are const int (include/KeyFrame.h:411-414, truncated from the Frame's floats at
 * src/KeyFrame.cc:70): with a distorted camera the cell range of a window can differ at cell
 * edges.  This is what the loop-closing overload SearchByProjection(KeyFrame*, Scw, ...)
 * (src/ORBmatcher.cc:470) sees; `bounds` are the Frame's float bounds as everywhere else. */
int orb_match_windowed_keyframe(int device_id, OrbFrameBounds bounds, int n_t, const float* t_x,
                                const float* t_y, const int* t_octave, const float* t_angle,
                                const uint8_t* t_desc, const uint8_t* t_taken, int n_q,
                                const uint8_t* q_active, const float* q_u, const float* q_v,
                                const float* q_radius, const int* q_min_level, const int* q_max_level,
                                const float* q_angle, const uint8_t* q_desc, int max_dist,
                                int check_orientation, int* t_match, int* n_matches);

/* ORBmatcher::SearchForInitialization(const Frame& F1, const Frame& F2,
 * vector<cv::Point2f>& vbPrevMatched, vector<int>& vnMatches12, int windowSize)
 * (src/ORBmatcher.cc:543-686).  Per F1 feature of octave 0: the level-0 features of F2
 * within windowSize of (prev_x, prev_y), best and second-best distance over those whose
 * recorded match distance is larger (:596), accepted if best <= TH_LOW and
 * best < second * nnratio; an accepted match takes the F2 feature away from the F1 feature
 * that held it (:619-623); rotation check over every accepted query (:653-677).
 * matches12[i1] = i2 or -1.  Updating vbPrevMatched (:681-683) is left to the caller. */
int orb_search_for_initialization(int device_id, OrbFrameBounds bounds, int n1,
                                  const int* octave1, const float* angle1, const uint8_t* desc1,
                                  const float* prev_x, const float* prev_y, int n2,
                                  const float* x2, const float* y2, const int* octave2,
                                  const float* angle2, const uint8_t* desc2, float window_size,
                                  float nnratio, int check_orientation, int* matches12,
                                  int* n_matches);

/* The candidate scan of ORBmatcher::Fuse (both overloads, src/ORBmatcher.cc:1133-1175,
 * :1302-1329) and of the two directions of ORBmatcher::SearchBySim3 (:1479-1506,
 * :1560-1586): per query GetFeaturesInArea(u, v, radius), the level test
 * kpLevel < q_level - 1 || kpLevel > q_level -> skip, optionally (chi2 > 0; the first Fuse
 * overload passes 5.99 and mvInvLevelSigma2) the reprojection gate
 * e2 * inv_level_sigma2[kpLevel] > chi2 -> skip, and the first minimum of the descriptor
 * distance.  Queries do not interact; what the callers do with the result (threshold
 * TH_LOW / TH_HIGH, Replace / AddObservation, the agreement check) stays on the host.
 * best_idx[i] = -1 and best_dist[i] = INT_MAX when no candidate is left. */
int orb_window_best(int device_id, OrbFrameBounds bounds, int n_t, const float* t_x,
                    const float* t_y, const int* t_octave, const uint8_t* t_desc,
                    const float* inv_level_sigma2, int nlevels, double chi2, int n_q,
                    const uint8_t* q_active, const float* q_u, const float* q_v,
                    const float* q_radius, const int* q_level, const uint8_t* q_desc,
                    int* best_idx, int* best_dist);

/* orb_window_best on a KeyFrame's grid (int-truncated query origin, see
 * orb_match_windowed_keyframe): every caller in the reference - Fuse x2 (src/ORBmatcher.cc:1133,
 * :1302) and SearchBySim3 (:1479, :1560) - queries a KeyFrame. */
int orb_window_best_keyframe(int device_id, OrbFrameBounds bounds, int n_t, const float* t_x,
                             const float* t_y, const int* t_octave, const uint8_t* t_desc,
                             const float* inv_level_sigma2, int nlevels, double chi2, int n_q,
                             const uint8_t* q_active, const float* q_u, const float* q_v,
                             const float* q_radius, const int* q_level, const uint8_t* q_desc,
                             int* best_idx, int* best_dist);

/* ORBmatcher::SearchByBoW, both overloads:
 *   SearchByBoW(KeyFrame* pKF, Frame& F, std::vector<MapPoint*>& vpMapPointMatches)
 *     (src/ORBmatcher.cc:222-381; accepts bestDist1 <= TH_LOW: pass max_dist = TH_LOW), and
 *   SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12)
 *     (:698-841; accepts bestDist1 < TH_LOW: pass max_dist = TH_LOW - 1).
 * Side 1 is the keyframe whose features are visited (ok1 = has a map point that is not
 * bad), side 2 the frame / second keyframe (ok2 = may be matched: 1 for every feature of a
 * Frame, "has a good map point" for a keyframe).  Feature vectors as in
 * orb_search_for_triangulation.  Best and second best are taken over the side-2 features of
 * the same node not yet matched; accepted if best <= max_dist and
 * (float)best < nnratio * (float)second.  matches12[idx1] = idx2 or -1 (the first overload
 * stores the same pairs under idx2). */
int orb_search_by_bow(int device_id, int n1, const float* angle1, const uint8_t* desc1,
                      const uint8_t* ok1, int n2, const float* angle2, const uint8_t* desc2,
                      const uint8_t* ok2, int n_nodes1, const int* node1, const int* start1,
                      const int* items1, int n_nodes2, const int* node2, const int* start2,
                      const int* items2, float nnratio, int max_dist, int check_orientation,
                      int* matches12, int* n_matches);

/* Stereo association, coarse stage, for the 752x480 stereo configuration.  No counterpart
 * in this fork of the reference (its Frame has no ComputeStereoMatches); stated the way
 * upstream ORB-SLAM2 states it, with this tree's ORBmatcher::DescriptorDistance and
 * TH_HIGH / TH_LOW (src/ORBmatcher.cc:41-43, 1991-2011): a right keypoint is a candidate of
 * the left keypoint iL if row (int)yL lies in floor(yR - r)..ceil(yR + r),
 * r = 2 * scale_factors[octaveR], its octave is within one of octaveL, and
 * xL - max_d <= xR <= xL - min_d; the first minimum of the distance in ascending right
 * index wins and is a match if below th_dist (upstream: (TH_HIGH + TH_LOW) / 2 = 75).
 * best_idx[iL] = right index or -1, best_dist[iL] = min(distance, TH_HIGH).
 * Sub-pixel refinement is not included. */
int orb_stereo_match(int device_id, int rows, int n_left, const float* l_x, const float* l_y,
                     const int* l_octave, const uint8_t* l_desc, int n_right, const float* r_x,
                     const float* r_y, const int* r_octave, const uint8_t* r_desc,
                     const float* scale_factors, int nlevels, float min_d, float max_d,
                     int th_dist, int* best_idx, int* best_dist, int* n_matches);

/* ORBmatcher::SearchForTriangulation(pKF1, pKF2, F12, vMatchedPairs, bOnlyStereo)
 * (src/ORBmatcher.cc:853-1048).  The two DBoW2::FeatureVector maps arrive as sorted
 * node ids with CSR item lists; (ex, ey) is the epipole of KF1's centre in KF2
 * (:862-871); has_mp = GetMapPoint(idx) != null.  matches12[idx1] = idx2 or -1. */
int orb_search_for_triangulation(
    int device_id, int n1, const float* x1, const float* y1, const int* octave1,
    const float* angle1, const uint8_t* desc1, const uint8_t* has_mp1, int n2, const float* x2,
    const float* y2, const int* octave2, const float* angle2, const uint8_t* desc2,
    const uint8_t* has_mp2, int n_nodes1, const int* node1, const int* start1, const int* items1,
    int n_nodes2, const int* node2, const int* start2, const int* items2, const float* f12,
    float ex, float ey, const float* scale_factors2, const float* level_sigma2_2, int nlevels,
    int check_orientation, int* matches12, int* n_matches);

/* Brute-force Hamming 2-nearest-neighbour search with the ratio test of the
 * reference's BoW / initialisation matchers ((float)best < ratio*(float)second,
 * src/ORBmatcher.cc:312,616): for each of n_q query descriptors the smallest and
 * second-smallest DescriptorDistance over n_db database rows (first minimum wins
 * ties, as in the reference's sequential scans).  All pointers are DEVICE pointers;
 * enqueued on `stream` and not waited for.
 *   db_index_base  global index of d_db row 0 (database shards of a multi-GPU job)
 *   d_best, d_second  [n_q] distances (256 = none);  d_idx [n_q] global row or -1
 *   d_accept [n_q] ratio-test decision, may be null. */
int orb_knn2_device(int device_id, const uint8_t* d_q, int n_q, const uint8_t* d_db, int n_db,
                    int db_index_base, float ratio, int* d_best, int* d_second, int* d_idx,
                    uint8_t* d_accept, void* stream);
/* The same search with the kernel chosen explicitly: 0 = as orb_knn2_device chooses,
 * 1 = popcount on the vector units, 2 = matrix cores (bits as +1 / -1 int8, distance =
 * (256 - dot product) / 2).  Both give identical results; the choice exists for tests and
 * measurements.  No counterpart in the reference. */
int orb_knn2_device_algo(int device_id, const uint8_t* d_q, int n_q, const uint8_t* d_db, int n_db,
                         int db_index_base, float ratio, int* d_best, int* d_second, int* d_idx,
                         uint8_t* d_accept, void* stream, int algo);
/* Merge of per-shard kNN-2 results (shards in ascending database order):
 * in arrays are [n_shards][n_q]; the merged result equals a single scan of the
 * concatenated database.  Device pointers. */
int orb_knn2_merge_device(int device_id, int n_shards, int n_q, const int* d_best,
                          const int* d_second, const int* d_idx, float ratio, int* d_out_best,
                          int* d_out_second, int* d_out_idx, uint8_t* d_out_accept, void* stream);
/* The per-query result of a shard as one 8-byte record, the form in which ranks exchange it:
 * word 0 = best | second << 16 (distances are at most 256), word 1 = global row (0xFFFFFFFF =
 * none).  orb_knn2_pack_device writes d_packed[n_q] records from the three arrays of
 * orb_knn2_device; orb_knn2_merge_packed_device is orb_knn2_merge_device on records laid out
 * [n_shards][n_q].  Device pointers, 8-byte aligned; enqueued on `stream`.  No counterpart in the
 * reference. */
int orb_knn2_pack_device(int device_id, int n_q, const int* d_best, const int* d_second,
                         const int* d_idx, void* d_packed, void* stream);
int orb_knn2_merge_packed_device(int device_id, int n_shards, int n_q, const void* d_packed,
                                 float ratio, int* d_out_best, int* d_out_second, int* d_out_idx,
                                 uint8_t* d_out_accept, void* stream);
/* The same search with the database sharded over the GPUs of a node, one process (rank) per GPU.
 * No counterpart in the reference, which has no multi-device code.  Rank r holds the database
 * rows [db_index_base, db_index_base + n_db_shard): the shards are contiguous and ascend with
 * the rank; the queries are replicated, and n_q must be the same number on every rank.
 * orb_knn2_sharded scans the local shard, exchanges the per-query records (above) with ONE
 * ncclAllGather (8 bytes x n_q per rank, RCCL over xGMI) and merges them in rank order, so every
 * rank ends up with the result of a single scan of the whole database, bit for bit (the earlier
 * shard keeps ties).  Everything is enqueued on `stream`; the first call, and a call with more
 * queries than any before, allocates the exchange buffers (n_q * (20 + 8 * world) bytes) and
 * waits for the stream first.
 * The call is collective: arguments are checked before any work, so a call that returns
 * ORBHIP_ERR_ARG has not entered the collective, and callers that pass the same kind of arguments
 * on every rank fail on every rank alike.  A failure after the checks (allocation, kernel launch)
 * on one rank cannot be reported to the others; a communicator the library created is then
 * aborted (ncclCommAbort), so that the other ranks' pending all-gather ends with an error instead
 * of waiting for ever, and refuses further calls.  A wrapped communicator is left to its owner.
 *   orb_knn2_comm_unique_id   rank 0 makes the 128-byte id (ncclGetUniqueId) and hands it to the
 *                             other ranks by its own means (MPI, a file, torch.distributed ...)
 *   orb_knn2_comm_create      every rank, collectively (ncclCommInitRank) on GPU device_id
 *   orb_knn2_comm_wrap        instead: use a communicator (ncclComm_t) the host already has
 * RCCL is loaded at run time (librccl.so.1); ORBHIP_ERR_UNSUPPORTED if it is not there. */
typedef struct OrbKnnComm OrbKnnComm;
#define ORBHIP_KNN_UNIQUE_ID_BYTES 128
int orb_knn2_comm_unique_id(uint8_t* id /* ORBHIP_KNN_UNIQUE_ID_BYTES */);
int orb_knn2_comm_create(const uint8_t* id, int rank, int world, int device_id, OrbKnnComm** out);
int orb_knn2_comm_wrap(void* nccl_comm, int rank, int world, int device_id, OrbKnnComm** out);
void orb_knn2_comm_destroy(OrbKnnComm* c);
/* This handle's rank and world size as given at creation, and the number of ranks the
 * communicator itself reports (ncclCommCount; -1 if RCCL does not offer it).  Null pointers are
 * skipped. */
int orb_knn2_comm_info(OrbKnnComm* c, int* rank, int* world, int* ranks_in_collective);
/* Milliseconds the last orb_knn2_sharded call on this handle spent, by HIP events on its stream,
 * in the shard scan (with packing) and in the exchange (all-gather + merge).  Waits for that
 * call.  Null pointers are skipped. */
int orb_knn2_sharded_times(OrbKnnComm* c, float* scan_ms, float* exchange_ms);
int orb_knn2_sharded(OrbKnnComm* c, const uint8_t* d_q, int n_q, const uint8_t* d_db_shard,
                     int n_db_shard, int db_index_base, float ratio, int* d_best, int* d_second,
                     int* d_idx, uint8_t* d_accept, void* stream);

/* Host-buffer convenience wrapper around orb_knn2_device on one GPU. */
int orb_knn2_bruteforce(int device_id, const uint8_t* q, int n_q, const uint8_t* db, int n_db,
                        float ratio, int* best, int* second, int* idx, uint8_t* accept);

/* MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:386-469) for n_points map points
 * at once.  desc holds the descriptors that observe each point (the rows the reference
 * collects at :407-416, in that order), point p's in rows start[p] .. start[p+1]) (start has
 * n_points + 1 entries, start[0] = 0).  best_idx[p] = position within the point's rows of the
 * descriptor with the least median distance to all of them (median = sorted row element
 * (int)(0.5 * (N - 1)), the first such descriptor on ties), -1 for a point without rows;
 * best_median[p] (may be null) = that median. */
int orb_distinctive_descriptors(int device_id, int n_points, const int* start, const uint8_t* desc,
                                int* best_idx, int* best_median);

/* ---- bag-of-words quantisation (DBoW2 vocabulary tree) ---- */

typedef struct OrbVocabulary OrbVocabulary;

/* Uploads a vocabulary tree as DBoW2::TemplatedVocabulary holds it after loadFromTextFile
 * (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1352-1438): node 0 is the root, node i has
 * the children children[child_start[i] .. child_start[i] + child_count[i]) in the order the
 * file lists them (ascending id, every child id above its parent's), node_desc[i] is its
 * 32-byte FORB descriptor, node_word[i] its word id if it is a leaf (else ignored);
 * levels = m_L.  Node weights stay with the caller.  The children array has n_nodes - 1
 * entries. */
int orb_vocabulary_create(int device_id, int n_nodes, int levels, const int* child_start,
                          const int* child_count, const int* children, const uint8_t* node_desc,
                          const int* node_word, OrbVocabulary** out);
void orb_vocabulary_destroy(OrbVocabulary* v);

/* TemplatedVocabulary::transform(feature, word_id, weight, nid, levelsup)
 * (TemplatedVocabulary.h:1232-1273) with FORB::distance (Thirdparty/DBoW2/DBoW2/FORB.cpp)
 * for n features at once: from the root, step to the child with the smallest Hamming
 * distance (the first one on ties) until a leaf; word_id[i] = that leaf's word,
 * node_id[i] = the node passed at depth levels - levelsup (0 if that depth is <= 0 or the
 * leaf is shallower).  The ordered BowVector / FeatureVector that
 * transform(features, v, fv, levelsup) (:1141-1208, as Frame::ComputeBoW calls it with
 * levelsup = 4) builds from these pairs are assembled by the caller.
 * The _device form takes DEVICE pointers (descriptors 16-byte aligned), is enqueued on
 * `stream` and not waited for. */
int orb_bow_transform_device(OrbVocabulary* v, int n, const uint8_t* d_desc, int levelsup,
                             int* d_word_id, int* d_node_id, void* stream);
int orb_bow_transform(OrbVocabulary* v, int n, const uint8_t* desc, int levelsup, int* word_id,
                      int* node_id);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* ORBHIP_H_ */
