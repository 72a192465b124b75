// Frame::UndistortKeyPoints and Frame::ComputeImageBounds (reference src/Frame.cc:502-596) on
// gfx950: cv::undistortPoints(src, dst, K, distCoef, Mat(), K) for the keypoints of a batch of
// frames, so that mvKeysUn exists on the GPU next to the extractor's output.
//
// cv::undistortPoints with its default criteria runs exactly five fixed-point iterations in
// double per point and rounds the result to float.  The kernel keeps OpenCV's order of
// operations (the file is built without multiply-add contraction), so its doubles are the ones
// a C restatement produces.
#include "orb_common.h"
#include "orb_hostbuf.h"

#include <vector>

namespace orbhip {
namespace {

struct CamD {
  double fx, fy, cx, cy, ifx, ify, k1, k2, p1, p2, k3;
};

__host__ __device__ inline void undistort_point(const CamD& c, float u_in, float v_in, float& u_out, float& v_out) {
  const double u = (double)u_in, v = (double)v_in;
  double x = (u - c.cx) * c.ifx, y = (v - c.cy) * c.ify;
  const double x0 = x, y0 = y;
  for (int j = 0; j < 5; j++) {
    const double r2 = x * x + y * y;
    // the rational terms k4..k6 and the prism / tilt terms are zero for a 4- or 5-coefficient
    // model: they contribute exact zeros (numerator exactly 1), so they are left out
    const double icdist = 1. / (1 + ((c.k3 * r2 + c.k2) * r2 + c.k1) * r2);
    if (icdist < 0) { x = (u - c.cx) * c.ifx; y = (v - c.cy) * c.ify; break; }
    const double dx = 2 * c.p1 * x * y + c.p2 * (r2 + 2 * x * x);
    const double dy = c.p1 * (r2 + 2 * y * y) + 2 * c.p2 * x * y;
    x = (x0 - dx) * icdist;
    y = (y0 - dy) * icdist;
  }
  // new camera matrix P = K, R = identity: RR = K, the homogeneous divisor is exactly 1
  u_out = (float)(c.fx * x + c.cx);
  v_out = (float)(c.fy * y + c.cy);
}

CamD make_cam(const OrbCamera& k) {
  CamD c;
  c.fx = k.fx; c.fy = k.fy; c.cx = k.cx; c.cy = k.cy;
  c.ifx = 1. / c.fx; c.ify = 1. / c.fy;
  c.k1 = k.k1; c.k2 = k.k2; c.p1 = k.p1; c.p2 = k.p2; c.k3 = k.k3;
  return c;
}

__global__ __launch_bounds__(256) void undistort_kps_kernel(int max_kp, const OrbKeyPoint* __restrict__ in,
                                                            const int* __restrict__ n, CamD c, int identity,
                                                            OrbKeyPoint* __restrict__ out) {
  const int f = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= min(n[f], max_kp)) return;
  OrbKeyPoint kp = in[(size_t)f * max_kp + i];
  if (!identity) undistort_point(c, kp.x, kp.y, kp.x, kp.y);
  out[(size_t)f * max_kp + i] = kp;
}

__global__ __launch_bounds__(256) void undistort_xy_kernel(int n, const float2* __restrict__ in, CamD c,
                                                           float2* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float2 p = in[i];
  undistort_point(c, p.x, p.y, p.x, p.y);
  out[i] = p;
}

bool cam_ok(const OrbCamera& k) { return k.fx != 0.f && k.fy != 0.f; }

}  // namespace
}  // namespace orbhip

using namespace orbhip;

extern "C" {

int orb_undistort_keypoints_device(int device_id, int n_frames, int max_kp, const OrbKeyPoint* d_kps, const int* d_n,
                                   OrbCamera cam, OrbKeyPoint* d_kps_un, void* stream) {
  if (n_frames <= 0 || max_kp <= 0) return ORBHIP_OK;
  if (!d_kps || !d_n || !d_kps_un) { set_error("null argument"); return ORBHIP_ERR_ARG; }
  if (!cam_ok(cam)) { set_error("focal length is zero"); return ORBHIP_ERR_ARG; }
  int prev = -1;
  hipGetDevice(&prev);
  if (prev != device_id) ORB_HIP_CHECK(hipSetDevice(device_id));
  // Frame::UndistortKeyPoints copies the keypoints when the first coefficient is zero (src/Frame.cc:506)
  hipLaunchKernelGGL(undistort_kps_kernel, dim3((max_kp + 255) / 256, n_frames), dim3(256), 0, (hipStream_t)stream, max_kp,
                     d_kps, d_n, make_cam(cam), cam.k1 == 0.0f ? 1 : 0, d_kps_un);
  hipError_t e = hipGetLastError();
  if (prev >= 0 && prev != device_id) hipSetDevice(prev);
  if (e != hipSuccess) return hip_fail(e, "undistort_kps_kernel");
  return ORBHIP_OK;
}

int orb_undistort_points(int device_id, int n, const float* xy, OrbCamera cam, float* xy_un) {
  if (n <= 0) return ORBHIP_OK;
  if (!xy || !xy_un) { set_error("null argument"); return ORBHIP_ERR_ARG; }
  if (!cam_ok(cam)) { set_error("focal length is zero"); return ORBHIP_ERR_ARG; }
  DevSel g(device_id);
  DevBuf buf;  // a stream and arena of the call's own
  float2 *d_in = nullptr, *d_out = nullptr;
  UP(buf, reinterpret_cast<const float2*>(xy), (size_t)n, d_in);
  UP(buf, (const float2*)nullptr, (size_t)n, d_out);
  hipLaunchKernelGGL(undistort_xy_kernel, dim3((n + 255) / 256), dim3(256), 0, buf.stream(), n, d_in, make_cam(cam), d_out);
  ORB_HIP_CHECK(hipGetLastError());
  DOWN(buf, reinterpret_cast<float2*>(xy_un), d_out, (size_t)n);
  SYNC(buf);
  return ORBHIP_OK;
}

int orb_compute_image_bounds(int rows, int cols, OrbCamera cam, OrbFrameBounds* out) {
  if (!out) { set_error("null argument"); return ORBHIP_ERR_ARG; }
  if (cam.k1 == 0.0f) {  // src/Frame.cc:588-595
    out->min_x = 0.f; out->max_x = (float)cols; out->min_y = 0.f; out->max_y = (float)rows;
    return ORBHIP_OK;
  }
  if (!cam_ok(cam)) { set_error("focal length is zero"); return ORBHIP_ERR_ARG; }
  // four corners through the same arithmetic on the host (src/Frame.cc:563-586)
  const CamD c = make_cam(cam);
  const float cx[4] = {0.f, (float)cols, 0.f, (float)cols}, cy[4] = {0.f, 0.f, (float)rows, (float)rows};
  float ux[4], uy[4];
  for (int i = 0; i < 4; i++) undistort_point(c, cx[i], cy[i], ux[i], uy[i]);
  out->min_x = std::min(ux[0], ux[2]);
  out->max_x = std::max(ux[1], ux[3]);
  out->min_y = std::min(uy[0], uy[1]);
  out->max_y = std::max(uy[2], uy[3]);
  return ORBHIP_OK;
}

}  // extern "C"
