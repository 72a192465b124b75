// Bag-of-words quantisation of ORB descriptors on gfx950: the descent of
// DBoW2::TemplatedVocabulary::transform(feature, word_id, weight, nid, levelsup)
// (reference Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1232-1273) with FORB::distance
// (Thirdparty/DBoW2/DBoW2/FORB.cpp) for every feature of a frame, one thread per feature.
// The ordered maps that transform(features, v, fv, levelsup) (:1141-1208) builds from the
// (word, node) pairs stay with the caller.
#include "orb_common.h"
#include "orb_hostbuf.h"

#include <string>
#include <vector>

struct OrbVocabulary {
  int device = 0;
  int n_nodes = 0, levels = 0;
  int* d_child_start = nullptr;
  int* d_child_count = nullptr;
  int* d_children = nullptr;
  int* d_word = nullptr;
  uint8_t* d_desc = nullptr;
  // staging for one call (grow-only)
};

namespace orbhip {
namespace {

__device__ __forceinline__ int bow_hamming(const uint32_t* q, const uint4 a, const uint4 b) {
  return __popc(q[0] ^ a.x) + __popc(q[1] ^ a.y) + __popc(q[2] ^ a.z) + __popc(q[3] ^ a.w) + __popc(q[4] ^ b.x) +
         __popc(q[5] ^ b.y) + __popc(q[6] ^ b.z) + __popc(q[7] ^ b.w);
}

__global__ __launch_bounds__(256) void bow_transform_kernel(int n, const uint4* __restrict__ feat,
                                                            const int* __restrict__ child_start,
                                                            const int* __restrict__ child_count,
                                                            const int* __restrict__ children,
                                                            const uint4* __restrict__ node_desc,
                                                            const int* __restrict__ node_word, int nid_level,
                                                            int max_depth, int* __restrict__ out_word,
                                                            int* __restrict__ out_node) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint4 f0 = feat[2 * i], f1 = feat[2 * i + 1];
  const uint32_t q[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
  int node = 0, nid = 0, level = 0;
  // a node's children are tried in order and the first minimum wins ('d < best_d', :1258);
  // children have larger ids than their parent, so the walk ends at a leaf
  for (int guard = 0; guard < max_depth; guard++) {
    const int cn = child_count[node];
    if (cn == 0) break;
    const int cs = child_start[node];
    level++;
    int best = 0x7FFFFFFF, bid = node;
    for (int c0 = 0; c0 < cn; c0 += 4) {
      int id[4];
      uint4 a[4], b[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {  // the loads of four children are in flight together
        id[u] = children[cs + min(c0 + u, cn - 1)];
        a[u] = node_desc[2 * id[u]];
        b[u] = node_desc[2 * id[u] + 1];
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int d = bow_hamming(q, a[u], b[u]);
        if (c0 + u < cn && d < best) { best = d; bid = id[u]; }
      }
    }
    node = bid;
    if (level == nid_level) nid = node;
  }
  out_word[i] = node_word[node];
  out_node[i] = nid;
}

}  // namespace
}  // namespace orbhip

using namespace orbhip;

extern "C" {

int orb_vocabulary_create(int device_id, int n_nodes, int levels, const int* child_start, const int* child_count,
                          const int* children, const uint8_t* node_desc, const int* node_word, OrbVocabulary** out) {
  if (!out) { set_error("null argument"); return ORBHIP_ERR_ARG; }
  *out = nullptr;
  if (n_nodes < 2 || levels < 1 || !child_start || !child_count || !children || !node_desc || !node_word) {
    set_error("empty vocabulary or null argument"); return ORBHIP_ERR_ARG;
  }
  // every node but the root is the child of exactly one node with a smaller id
  long long total = 0;
  for (int i = 0; i < n_nodes; i++) {
    if (child_count[i] < 0 || child_start[i] < 0 || (long long)child_start[i] + child_count[i] > n_nodes - 1) {
      set_error("child range out of bounds"); return ORBHIP_ERR_ARG;
    }
    for (int c = 0; c < child_count[i]; c++) {
      const int id = children[child_start[i] + c];
      if (id <= i || id >= n_nodes) { set_error("children must have larger ids than their parent"); return ORBHIP_ERR_ARG; }
    }
    if (child_count[i] == 0 && node_word[i] < 0 && i > 0) { set_error("leaf without a word id"); return ORBHIP_ERR_ARG; }
    total += child_count[i];
  }
  if (total != n_nodes - 1) { set_error("child lists do not cover the nodes"); return ORBHIP_ERR_ARG; }
  int prev = -1;
  hipGetDevice(&prev);
  if (prev != device_id) ORB_HIP_CHECK(hipSetDevice(device_id));
  OrbVocabulary* v = new OrbVocabulary;
  v->device = device_id; v->n_nodes = n_nodes; v->levels = levels;
  hipError_t e = hipMalloc(&v->d_child_start, sizeof(int) * n_nodes);
  if (e == hipSuccess) e = hipMalloc(&v->d_child_count, sizeof(int) * n_nodes);
  if (e == hipSuccess) e = hipMalloc(&v->d_children, sizeof(int) * n_nodes);
  if (e == hipSuccess) e = hipMalloc(&v->d_word, sizeof(int) * n_nodes);
  if (e == hipSuccess) e = hipMalloc(&v->d_desc, (size_t)32 * n_nodes);
  if (e == hipSuccess) e = hipMemcpy(v->d_child_start, child_start, sizeof(int) * n_nodes, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(v->d_child_count, child_count, sizeof(int) * n_nodes, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(v->d_children, children, sizeof(int) * (n_nodes - 1), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(v->d_word, node_word, sizeof(int) * n_nodes, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(v->d_desc, node_desc, (size_t)32 * n_nodes, hipMemcpyHostToDevice);
  if (prev >= 0 && prev != device_id) hipSetDevice(prev);
  if (e != hipSuccess) {
    hipFree(v->d_child_start); hipFree(v->d_child_count); hipFree(v->d_children); hipFree(v->d_word); hipFree(v->d_desc);
    delete v;
    return hip_fail(e, "vocabulary upload");
  }
  *out = v;
  return ORBHIP_OK;
}

void orb_vocabulary_destroy(OrbVocabulary* v) {
  if (!v) return;
  int prev = -1;
  hipGetDevice(&prev);
  if (prev != v->device) hipSetDevice(v->device);
  hipDeviceSynchronize();
  hipFree(v->d_child_start); hipFree(v->d_child_count); hipFree(v->d_children); hipFree(v->d_word); hipFree(v->d_desc);
  if (prev >= 0 && prev != v->device) hipSetDevice(prev);
  delete v;
}

int orb_bow_transform_device(OrbVocabulary* v, int n, const uint8_t* d_desc, int levelsup, int* d_word_id,
                             int* d_node_id, void* stream) {
  if (!v) { set_error("null vocabulary"); return ORBHIP_ERR_ARG; }
  if (n <= 0) return ORBHIP_OK;
  if (!d_desc || !d_word_id || !d_node_id) { set_error("null argument"); return ORBHIP_ERR_ARG; }
  if ((uintptr_t)d_desc & 15) { set_error("descriptors must be 16-byte aligned"); return ORBHIP_ERR_ARG; }
  int prev = -1;
  hipGetDevice(&prev);
  if (prev != v->device) ORB_HIP_CHECK(hipSetDevice(v->device));
  hipLaunchKernelGGL(bow_transform_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, n,
                     reinterpret_cast<const uint4*>(d_desc), v->d_child_start, v->d_child_count, v->d_children,
                     reinterpret_cast<const uint4*>(v->d_desc), v->d_word, v->levels - levelsup, v->n_nodes, d_word_id,
                     d_node_id);
  hipError_t e = hipGetLastError();
  if (prev >= 0 && prev != v->device) hipSetDevice(prev);
  if (e != hipSuccess) return hip_fail(e, "bow_transform_kernel");
  return ORBHIP_OK;
}

int orb_bow_transform(OrbVocabulary* v, int n, const uint8_t* desc, int levelsup, int* word_id, int* node_id) {
  if (!v) { set_error("null vocabulary"); return ORBHIP_ERR_ARG; }
  if (n <= 0) return ORBHIP_OK;
  if (!desc || !word_id || !node_id) { set_error("null argument"); return ORBHIP_ERR_ARG; }
  // a stream and arena of the call's own: Frame::ComputeBoW and KeyFrame::ComputeBoW run on the
  // tracking, mapping and loop-closing threads with one shared vocabulary
  DevSel g(v->device);
  DevBuf buf;
  uint8_t* d_q = nullptr;
  int* d_out = nullptr;
  UP(buf, desc, (size_t)32 * n, d_q);
  UP(buf, (const int*)nullptr, 2 * (size_t)n, d_out);
  const int rc = orb_bow_transform_device(v, n, d_q, levelsup, d_out, d_out + n, buf.stream());
  if (rc != ORBHIP_OK) return rc;
  DOWN(buf, word_id, d_out, (size_t)n);
  DOWN(buf, node_id, d_out + n, (size_t)n);
  SYNC(buf);
  return ORBHIP_OK;
}

}  // extern "C"
