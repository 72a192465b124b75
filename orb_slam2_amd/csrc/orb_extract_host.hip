// Host side of the extractor: parameter tables, per-image-size plan, device buffers,
// launch sequence and the extern "C" entry points declared in include/orbhip.h.
#include <math.h>
#include <string.h>

#include <algorithm>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "orb_common.h"

namespace orbhip {

// launchers and limits defined next to the kernels
void launch_pyr_level0(const uint8_t*, size_t, size_t, uint8_t*, const Plan&, int*, int, hipStream_t);
void launch_pyr_level0_color(const uint8_t*, size_t, size_t, int, bool, const int*, uint8_t*, const Plan&, int*, int, hipStream_t);
void launch_pyr_resize(uint8_t*, const Plan&, int, const XTab*, const YTab*, int, hipStream_t);
void launch_fast(const uint8_t*, const Plan&, const FastWork*, int, int, uint32_t*, int*, int, int, int, hipStream_t);
int fast_rows_dense();
int fast_use_width_narrow();
void launch_pyr_resize_dual(uint8_t*, const Plan&, int, const uint32_t*, const uint32_t*, const YTab*, const int2*, int,
                            const int2*, int, const int2*, int, int, int, int, int, int, int, int, hipStream_t);
hipError_t launch_octree(const uint32_t*, const int*, uint32_t*, uint32_t*, int*, const Plan&, int, int, hipStream_t);
void launch_blur(const uint8_t*, uint8_t*, const Plan&, const BlurItem*, int, const int*, int, hipStream_t);
hipError_t upload_orient_tables();
void launch_orient_desc(const uint8_t*, const uint8_t*, const uint32_t*, const int*, const Plan&, OrbKeyPoint*,
                        uint8_t*, int*, int, int, const int*, hipStream_t);
void launch_pyr_resize_tile(uint8_t*, const Plan&, int, const uint32_t*, const YTab*, const int2*, int, const int2*, int,
                            int, int, int, int, hipStream_t);
void launch_pyr_level01(const uint8_t*, size_t, size_t, uint8_t*, const Plan&, const uint32_t*, const YTab*, const int2*,
                        int, const int2*, int, const int2*, int, int, int, int*, int, hipStream_t);
int pyr_tile_cols();
int pyr_tile_rows();
int pyr_tile_rows_small();
int pyr_tile_max_src_rows();
int pyr_tile_max_src_bytes();
int fast_use_width();
int fast_max_cells();
int fast_max_view_h();
int octree_max_slots();
int blur_tile_w();
int blur_tile_h();

static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
int hip_fail(hipError_t e, const char* what) {
  g_err = std::string(what) + ": " + hipGetErrorString(e);
  return ORBHIP_ERR_HIP;
}

static inline int cv_round(float v) { return (int)lrintf(v); }
static inline int cv_round(double v) { return (int)lrint(v); }

}  // namespace orbhip

using namespace orbhip;

// A few host threads that share the packing of a batch into / out of the pinned staging
// buffers: one thread copies at about 10 GB/s, far less than the DMA engines and the kernels
// take.  submit(n, f) queues f(0) .. f(n - 1) and returns at once; the jobs are worked on in the
// order they were queued.  wait(job) returns when that job's items are done, wait_all() when
// every queued item is; both lend the calling thread to the queue while they wait.
// parallel_for(n, f) is submit + wait.  One thread at a time may submit and wait.
class HostCopyPool {
 public:
  struct Job {
    std::function<void(int)> f;
    int count = 0, next = 0, left = 0;
  };
  using JobRef = std::shared_ptr<Job>;

  explicit HostCopyPool(int workers) {
    for (int i = 0; i < workers; i++) threads_.emplace_back([this] { run(); });
  }
  ~HostCopyPool() {
    wait_all();
    { std::lock_guard<std::mutex> l(m_); stop_ = true; }
    cv_.notify_all();
    for (auto& t : threads_) t.join();
  }
  JobRef submit(int n, std::function<void(int)> f) {
    JobRef j = std::make_shared<Job>();
    if (n <= 0) return j;
    j->f = std::move(f); j->count = n; j->left = n;
    { std::lock_guard<std::mutex> l(m_); queue_.push_back(j); pending_ += n; }
    cv_.notify_all();
    return j;
  }
  void wait(const JobRef& j) { help_until([&] { return j->left == 0; }); }
  void wait_all() { help_until([&] { return pending_ == 0; }); }
  void parallel_for(int n, const std::function<void(int)>& f) { wait(submit(n, f)); }

 private:
  // (under the lock) the next item of the oldest job that still has items to hand out
  bool take(JobRef& j, int& i) {
    while (!queue_.empty() && queue_.front()->next >= queue_.front()->count) queue_.pop_front();
    if (queue_.empty()) return false;
    j = queue_.front();
    i = j->next++;
    return true;
  }
  // (under the lock) one item has been worked off
  void finished(const JobRef& j) {
    --pending_;
    if (--j->left == 0 || pending_ == 0) done_.notify_all();
  }
  template <typename Pred>
  void help_until(Pred ready) {
    std::unique_lock<std::mutex> l(m_);
    while (!ready()) {
      JobRef j; int i;
      if (take(j, i)) {
        l.unlock();
        j->f(i);
        l.lock();
        finished(j);
      } else {
        done_.wait(l);
      }
    }
  }
  void run() {
    std::unique_lock<std::mutex> l(m_);
    for (;;) {
      JobRef j; int i = 0;
      cv_.wait(l, [&] { return stop_ || take(j, i); });
      if (!j) return;  // stop_, and nothing taken
      l.unlock();
      j->f(i);
      l.lock();
      finished(j);
    }
  }
  std::vector<std::thread> threads_;
  std::mutex m_;
  std::condition_variable cv_, done_;
  std::deque<JobRef> queue_;
  long pending_ = 0;  // items queued or being worked on, over all jobs
  bool stop_ = false;
};

// waits for everything a call has queued before the call's locals go away, whichever way it returns
struct PoolDrain {
  HostCopyPool* pool;
  ~PoolDrain() { if (pool) pool->wait_all(); }
};

struct OrbExtractor {
  int nfeatures, nlevels, iniTh, minTh, device;
  double scaleFactor;
  float sf[kMaxLevels], inv_sf[kMaxLevels], sigma2[kMaxLevels], inv_sigma2[kMaxLevels];
  int quota[kMaxLevels];
  int umax[16];
  int taps[4];  // first four taps of the 7-tap blur kernel (it is symmetric), 8.8 fixed point
  int gray[4];  // cvtColor weights R, G, B and their shift
  hipStream_t stream = nullptr;
  // host-buffer batches in pieces: uploads and downloads on streams of their own, beside the kernels
  hipStream_t up = nullptr, down = nullptr;
  std::vector<hipEvent_t> pipe_ev;

  // plan for the current image size
  Plan plan{};
  bool have_plan = false;
  int cap_frames = 0;  // frames the buffers can hold
  int last_frames = 0;
  int ot_slots = 0;    // quad-tree node slots the largest level needs
  // tiled resize: per level, packed column table (one entry per slab column) and the
  // source ranges of every tile column / tile row; tile_ok[l] = the tiled kernel applies
  // yr_off / ngy / src_rows exist twice: [0] for tiles of pyr_tile_rows() rows (batches), [1] for the
  // short tiles of pyr_tile_rows_small() rows that a few frames are cut into, so that a level of one
  // frame is spread over many workgroups instead of waiting for a handful of long ones
  struct TileInfo { bool ok = false; int xtp_off = 0, xr_off = 0, ngx = 0, src_pitch = 0; int yr_off[2] = {0, 0}, ngy[2] = {0, 0}, src_rows[2] = {0, 0}; };
  // levels 0 and 1 in one launch (pyr_level01_kernel): usable for this plan, and where the table of
  // the first level-0 row each tile row writes starts in d_ranges
  bool fuse01 = false;
  int own_off = 0;
  TileInfo tile[kMaxLevels];
  // launches of a few frames: level l and level l + 1 in one launch (pair[l].ok), see pyr_resize_dual_kernel
  struct PairInfo { bool ok = false; int spitch0 = 0, pitchB = 0, offB = 0, lds = 0; };
  PairInfo pair[kMaxLevels];
  uint32_t* d_xtp = nullptr;
  int2* d_ranges = nullptr;
  std::vector<FastItem> fast_items;
  int fast_items_tall = 0;   // the first so many items have views of more than fast_rows_dense() rows
  int fast_items_few = 0;    // items [size - fast_items_few, size) are the same cells cut for launches of a few frames
  std::vector<BlurItem> blur_items;
  // device buffers
  uint8_t *d_pyr = nullptr, *d_blur = nullptr;
  // fused_blur: the descriptor kernel blurs each keypoint's patch itself and d_blur is only filled
  // when orb_get_blurred_level asks for it (blur_stale: not yet, for the last call's frames)
  bool fused_blur = true, blur_stale = false;
  XTab* d_xtab = nullptr;
  YTab* d_ytab = nullptr;
  FastWork* d_fast_items = nullptr;
  BlurItem* d_blur_items = nullptr;
  uint32_t *d_cand = nullptr, *d_pnode = nullptr, *d_sel = nullptr;
  int *d_cand_cnt = nullptr, *d_sel_cnt = nullptr;
  // staging for the host-buffer entry points
  uint8_t* d_img = nullptr;
  size_t d_img_bytes = 0;
  uint8_t* h_img = nullptr;  // pinned staging for host images
  size_t h_img_bytes = 0;
  uint8_t* h_out = nullptr;  // pinned staging for keypoints, descriptors and counts
  size_t h_out_bytes = 0;
  OrbKeyPoint* d_kps = nullptr;
  uint8_t* d_desc = nullptr;
  int* d_nout = nullptr;
  size_t out_cap = 0;  // frames*max_kp the output staging holds
  int out_frames = 0;
  HostCopyPool* pool = nullptr;  // created by the first call that packs more than a few MB
  // stage timing
  bool profiling = false;
  int channels = 1;       // interleaved bytes per input pixel: 1 grey, 3, 4
  bool rgb_order = true;  // first colour byte is red (the reference's Camera.RGB setting)
  std::vector<hipEvent_t> ev;  // ORBHIP_NUM_STAGES + 1 events per profiled call
  int prof_calls = 0;          // profiled calls since the times were last read
  // The host-buffer call for a few frames is bound by launch overhead (a memset, 13 dependent
  // kernels, four copies): its device work is captured once into a graph and replayed while
  // nothing the graph refers to (sizes, buffers, tables) has changed.
  hipGraphExec_t graph_exec = nullptr;
  // Last device-buffer call: an event on the caller's stream, so that a later call that has to
  // reallocate, and the accessors, wait for that call's work instead of for the whole device.
  hipEvent_t done_ev = nullptr;
  bool done_pending = false;
  // Device-buffer batches of kSplitMinFrames frames and more go through as two halves, the second on
  // this stream and one stage behind the first: the pyramid of one half (bound by memory bandwidth)
  // then runs beside FAST and the descriptor kernel of the other (bound by instruction issue).
  hipStream_t half = nullptr;
  int half_prio = 0;
  hipEvent_t ev_fork = nullptr, ev_stage = nullptr, ev_join = nullptr;
  bool split_batches = true;
  bool async_alloc = false;  // the device has stream-ordered memory pools: buffers come from hipMallocAsync
  struct GraphKey {
    int n_frames = 0, rows = 0, cols = 0, max_kp = 0, channels = 0, rgb = 0, fused = 0;
    int taps[4] = {0, 0, 0, 0}, gray[4] = {0, 0, 0, 0};
    const void *d_img = nullptr, *h_img = nullptr, *h_out = nullptr, *d_kps = nullptr, *d_pyr = nullptr;
    bool operator==(const GraphKey& o) const {
      return n_frames == o.n_frames && rows == o.rows && cols == o.cols && max_kp == o.max_kp && channels == o.channels &&
             rgb == o.rgb && fused == o.fused && memcmp(taps, o.taps, sizeof(taps)) == 0 &&
             memcmp(gray, o.gray, sizeof(gray)) == 0 && d_img == o.d_img && h_img == o.h_img && h_out == o.h_out &&
             d_kps == o.d_kps && d_pyr == o.d_pyr;
    }
  } graph_key;
};

namespace {

constexpr int kSmallBatchFrames = 8;               // up to this many frames a launch is cut into more, smaller workgroups
constexpr int kFewFrames = 4;                       // up to this many frames FAST takes shorter runs of cells
constexpr int kPipeMinFrames = 64;                  // host-buffer batches from this size on go through in pieces
constexpr int kPipePieces = 8;                      // ... at most this many
constexpr int kSplitMinFrames = 64;                 // device-buffer batches from this size on run as two staggered halves
constexpr int kGraphMaxFrames = 8;                 // batches up to this size replay a captured graph
constexpr int kPoolThreads = 12;                   // host threads that pack a large batch (the caller is one)
// ORBHIP_HOST_THREADS in the environment overrides that count (1 .. 64) for pools made afterwards
static int pool_threads() {
  const char* e = getenv("ORBHIP_HOST_THREADS");
  const int n = e ? atoi(e) : 0;
  return n >= 1 && n <= 64 ? n : kPoolThreads;
}
constexpr size_t kPoolMinBytes = (size_t)4 << 20;  // smaller batches are packed by the caller alone

// Device buffers of a handle come from the device's stream-ordered pool, allocated and freed on the
// handle's own stream: hipMalloc / hipFree wait for every stream of the device, which would make a
// tracking thread that changes its image size wait for the mapping and loop-closing threads' work.
// (Devices without memory pools fall back to hipMalloc / hipFree.)
template <typename T>
hipError_t dev_alloc(OrbExtractor* h, T** p, size_t bytes) {
  bytes = std::max<size_t>(bytes, 16);
  if (h->async_alloc) return hipMallocAsync(reinterpret_cast<void**>(p), bytes, h->stream);
  return hipMalloc(reinterpret_cast<void**>(p), bytes);
}
void dev_free(OrbExtractor* h, void* p) {
  if (!p) return;
  if (h->async_alloc) hipFreeAsync(p, h->stream);
  else hipFree(p);
}

// The captured graph refers to the handle's buffers by address: it goes whenever one of them does.
void drop_graph(OrbExtractor* h) {
  if (h->graph_exec) { hipGraphExecDestroy(h->graph_exec); h->graph_exec = nullptr; }
}

// Waits for everything this handle has queued, on its own streams and on the stream of the last
// device-buffer call; other handles' and other threads' streams are not waited for.
hipError_t wait_own(OrbExtractor* h) {
  hipError_t e = hipStreamSynchronize(h->stream);
  if (e != hipSuccess) return e;
  if (h->half && (e = hipStreamSynchronize(h->half)) != hipSuccess) return e;
  if (h->up && (e = hipStreamSynchronize(h->up)) != hipSuccess) return e;
  if (h->down && (e = hipStreamSynchronize(h->down)) != hipSuccess) return e;
  if (h->done_pending) {
    if ((e = hipEventSynchronize(h->done_ev)) != hipSuccess) return e;
    h->done_pending = false;
  }
  return hipSuccess;
}

void free_plan_buffers(OrbExtractor* h) {
  drop_graph(h);
  dev_free(h, h->d_pyr); dev_free(h, h->d_blur); dev_free(h, h->d_xtab); dev_free(h, h->d_ytab);
  dev_free(h, h->d_fast_items); dev_free(h, h->d_blur_items); dev_free(h, h->d_cand); dev_free(h, h->d_pnode);
  dev_free(h, h->d_sel); dev_free(h, h->d_cand_cnt); dev_free(h, h->d_sel_cnt);
  dev_free(h, h->d_xtp); dev_free(h, h->d_ranges);
  h->d_xtp = nullptr; h->d_ranges = nullptr;
  h->d_pyr = h->d_blur = nullptr; h->d_xtab = nullptr; h->d_ytab = nullptr;
  h->d_fast_items = nullptr; h->d_blur_items = nullptr;
  h->d_cand = h->d_pnode = h->d_sel = nullptr; h->d_cand_cnt = h->d_sel_cnt = nullptr;
  h->have_plan = false; h->cap_frames = 0;
}

// cv::resize(INTER_LINEAR) coefficient tables for one axis: source index and the two
// 11-bit fixed-point weights of every destination index.
void resize_axis(int dn, int sn, std::vector<int>& ofs, std::vector<int16_t>& w0, std::vector<int16_t>& w1) {
  const double scale = 1.0 / ((double)dn / sn);
  ofs.resize(dn); w0.resize(dn); w1.resize(dn);
  for (int d = 0; d < dn; d++) {
    float fx = (float)((d + 0.5) * scale - 0.5);
    int s = (int)floorf(fx);
    fx -= s;
    ofs[d] = s;
    auto sat = [](int v) { return (int16_t)std::min(std::max(v, -32768), 32767); };
    // the caller decides how out-of-range indices are clamped (x and y differ in OpenCV)
    w0[d] = sat(cv_round((1.f - fx) * 2048));
    w1[d] = sat(cv_round(fx * 2048));
  }
}

int build_plan(OrbExtractor* h, int rows, int cols, Plan& P, std::vector<XTab>& xtab, std::vector<YTab>& ytab,
               std::vector<FastItem>& fitems, std::vector<FastItem>& fitems_few, std::vector<BlurItem>& bitems) {
  memset(&P, 0, sizeof(P));
  P.nlevels = h->nlevels; P.rows = rows; P.cols = cols;
  size_t off = 0;
  int cand_off = 0, sel_off = 0;
  xtab.clear(); ytab.clear(); fitems.clear(); fitems_few.clear(); bitems.clear();
  for (int l = 0; l < h->nlevels; l++) {
    LevelGeom& L = P.lv[l];
    const float scale = h->inv_sf[l];
    L.w = cv_round((float)cols * scale);
    L.h = cv_round((float)rows * scale);
    if (L.w - 2 * kBorder > kCoordMask || L.h - 2 * kBorder > kCoordMask || L.w > 65000 || L.h > 65000) {
      set_error("image too large: level coordinates must fit 12 bits (at most 4127 px per side)");
      return ORBHIP_ERR_UNSUPPORTED;
    }
    L.pitch = (L.w + kXOff + kEdge + 63) / 64 * 64;
    L.off = (int)off;
    off += (size_t)L.pitch * (L.h + 2 * kEdge);
    off = (off + 255) / 256 * 256;
    if (off > 0x7fffffffull) { set_error("frame slab exceeds 2 GiB"); return ORBHIP_ERR_UNSUPPORTED; }
    L.maxBX = L.w - kBorder; L.maxBY = L.h - kBorder;
    const float width = (float)(L.maxBX - kBorder), height = (float)(L.maxBY - kBorder);
    L.nCols = (int)(width / 30.f); L.nRows = (int)(height / 30.f);
    if (L.nCols < 1 || L.nRows < 1) {
      set_error("pyramid level " + std::to_string(l) + " is smaller than 62 px: the reference divides by zero here");
      return ORBHIP_ERR_UNSUPPORTED;
    }
    L.wCell = (int)ceilf(width / L.nCols); L.hCell = (int)ceilf(height / L.nRows);
    L.wCellM = 65536 / std::max(1, L.wCell) + 1;
    if (L.hCell + 6 > fast_max_view_h() || L.wCell > fast_use_width()) {
      set_error("cell size out of range"); return ORBHIP_ERR_UNSUPPORTED;
    }
    L.nValidCols = 0; L.nValidRows = 0;
    for (int j = 0; j < L.nCols; j++) if (kBorder + j * L.wCell < L.maxBX - 6) L.nValidCols++;
    for (int i = 0; i < L.nRows; i++) if (kBorder + i * L.hCell < L.maxBY - 3) L.nValidRows++;
    L.quota = h->quota[l];
    L.nIni = (int)roundf(width / (float)(L.maxBY - kBorder));
    if (L.nIni < 1) {
      set_error("level aspect ratio below 0.5: the reference's quad-tree has no root here");
      return ORBHIP_ERR_UNSUPPORTED;
    }
    L.hX = width / L.nIni;
    const int max_nodes = std::max(L.quota + 3, 4 * L.nIni);
    if (max_nodes + L.nIni + 8 > octree_max_slots()) {
      set_error("nfeatures too large for this build: a level needs " + std::to_string(max_nodes) +
                " quad-tree nodes, limit " + std::to_string(octree_max_slots() - L.nIni - 8));
      return ORBHIP_ERR_UNSUPPORTED;
    }
    L.cand_off = cand_off;
    L.cand_cap = L.nValidCols * L.nValidRows * ((L.wCell + 1) / 2) * ((L.hCell + 1) / 2);
    cand_off += (L.cand_cap + 63) / 64 * 64;
    L.sel_off = sel_off;
    L.sel_cap = max_nodes + 1;
    sel_off += L.sel_cap;
    L.scale = h->sf[l];
    L.kp_size = (float)(int)(31 * h->sf[l]);
    // resize tables (level l from level l-1)
    L.xtab_off = (int)xtab.size(); L.ytab_off = (int)ytab.size();
    if (l > 0) {
      const LevelGeom& S = P.lv[l - 1];
      std::vector<int> ofs; std::vector<int16_t> w0, w1;
      resize_axis(L.w, S.w, ofs, w0, w1);
      for (int d = 0; d < L.w; d++) {
        XTab t{};
        int s = ofs[d]; int16_t a0 = w0[d], a1 = w1[d];
        if (s < 0) { s = 0; a0 = 2048; a1 = 0; }
        if (s >= S.w - 1) { s = S.w - 1; a0 = 2048; a1 = 0; }
        t.sx = (uint16_t)s; t.a0 = a0; t.a1 = a1;
        xtab.push_back(t);
      }
      resize_axis(L.h, S.h, ofs, w0, w1);
      for (int d = 0; d < L.h; d++) {
        YTab t{};
        auto clip = [&](int v) { return v < 0 ? 0 : (v < S.h ? v : S.h - 1); };
        t.sy0 = (uint16_t)clip(ofs[d]); t.sy1 = (uint16_t)clip(ofs[d] + 1);
        t.b0 = w0[d]; t.b1 = w1[d];
        ytab.push_back(t);
      }
    }
    // FAST work items: runs of cells per cell row
    // (levels whose cell views are taller than fast_rows_dense() are staged at the narrow pitch: shorter runs)
    const int usew = L.hCell + 6 > fast_rows_dense() ? fast_use_width_narrow() : fast_use_width();
    const int G = std::max(1, std::min(fast_max_cells(), usew / L.wCell));
    for (int i = 0; i < L.nValidRows; i++)
      for (int j0 = 0; j0 < L.nValidCols; j0 += G) {
        FastItem it; it.level = (int16_t)l; it.row = (int16_t)i; it.j0 = (int16_t)j0;
        it.ncells = (int16_t)std::min(G, L.nValidCols - j0);
        fitems.push_back(it);
      }
    // the same cells for launches of up to kFewFrames frames: one more run per cell row, of even
    // length (a frame alone has fewer workgroups than the GPU has CUs, and the longest sets the time)
    const int runs = (L.nValidCols + G - 1) / G + 1;
    const int Gf = std::max(1, (L.nValidCols + runs - 1) / runs);
    for (int i = 0; i < L.nValidRows; i++)
      for (int j0 = 0; j0 < L.nValidCols; j0 += Gf) {
        FastItem it; it.level = (int16_t)l; it.row = (int16_t)i; it.j0 = (int16_t)j0;
        it.ncells = (int16_t)std::min(Gf, L.nValidCols - j0);
        fitems_few.push_back(it);
      }
    for (int ty = 0; ty * blur_tile_h() < L.h; ty++)
      for (int tx = 0; tx * blur_tile_w() < L.w; tx++) {
        BlurItem it; it.level = (int16_t)l; it.tx = (int16_t)tx; it.ty = (int16_t)ty; it.pad = 0;
        bitems.push_back(it);
      }
  }
  P.slab = off;
  P.cand_total = cand_off;
  P.sel_total = sel_off;
  return ORBHIP_OK;
}

int ensure_plan(OrbExtractor* h, int rows, int cols, int nframes) {
  if (h->have_plan && h->plan.rows == rows && h->plan.cols == cols && nframes <= h->cap_frames) return ORBHIP_OK;
  Plan P; std::vector<XTab> xtab; std::vector<YTab> ytab; std::vector<FastItem> fi, fi_few; std::vector<BlurItem> bi;
  int rc = build_plan(h, rows, cols, P, xtab, ytab, fi, fi_few, bi);
  if (rc != ORBHIP_OK) return rc;
  // FAST items of levels whose cell views are too tall for six workgroups per CU at the wide LDS
  // pitch go first (they are staged at the narrow pitch, or launched apart)
  const int dense_rows = fast_rows_dense();
  const size_t n_tall = std::stable_partition(fi.begin(), fi.end(), [&](const FastItem& it) {
                          return P.lv[it.level].hCell + 6 > dense_rows; }) - fi.begin();
  fi.insert(fi.end(), fi_few.begin(), fi_few.end());  // one device array: the items for batches, then those for a few frames
  ORB_HIP_CHECK(wait_own(h));
  const int cap = std::max(nframes, (h->have_plan && h->plan.rows == rows && h->plan.cols == cols) ? h->cap_frames : 0);
  free_plan_buffers(h);
  ORB_HIP_CHECK(dev_alloc(h, &h->d_pyr, P.slab * cap));
  ORB_HIP_CHECK(dev_alloc(h, &h->d_blur, P.slab * cap));
  ORB_HIP_CHECK(hipMemsetAsync(h->d_pyr, 0, P.slab * cap, h->stream));
  ORB_HIP_CHECK(hipMemsetAsync(h->d_blur, 0, P.slab * cap, h->stream));
  ORB_HIP_CHECK(dev_alloc(h, &h->d_xtab, sizeof(XTab) * std::max<size_t>(xtab.size(), 1)));
  ORB_HIP_CHECK(dev_alloc(h, &h->d_ytab, sizeof(YTab) * std::max<size_t>(ytab.size(), 1)));
  // the items as the kernel reads them (see FastWork; the same arithmetic the kernel used to do per workgroup)
  std::vector<FastWork> fw(fi.size());
  for (size_t i = 0; i < fi.size(); i++) {
    const FastItem& it = fi[i];
    const LevelGeom& L = P.lv[it.level];
    FastWork& w = fw[i];
    memset(&w, 0, sizeof(w));
    const int iniY = kBorder + it.row * L.hCell, maxY = std::min(iniY + L.hCell + 6, L.maxBY);
    const int xb = kBorder + it.j0 * L.wCell, xe = std::min(kBorder + (it.j0 + it.ncells) * L.wCell + 6, L.maxBX);
    const int a0 = (xb + kXOff) & ~3, xoff = (xb + kXOff) - a0 + 3, wi = xe - xb - 6;
    w.base_off = L.off + (iniY + kEdge) * L.pitch + a0;
    w.pitch = L.pitch; w.cand_off = L.cand_off; w.cand_cap = L.cand_cap; w.wCellM = L.wCellM;
    w.level = it.level; w.ncells = it.ncells; w.wCell = (int16_t)L.wCell; w.hCell = (int16_t)L.hCell;
    w.tw = (int16_t)(xe - xb); w.th = (int16_t)(maxY - iniY); w.xb = (int16_t)xb; w.iniY = (int16_t)iniY;
    w.xoff = (int16_t)xoff; w.ndw = (int16_t)((xoff + 3 + wi + 3) >> 2);
  }
  ORB_HIP_CHECK(dev_alloc(h, &h->d_fast_items, sizeof(FastWork) * fw.size()));
  ORB_HIP_CHECK(dev_alloc(h, &h->d_blur_items, sizeof(BlurItem) * bi.size()));
  ORB_HIP_CHECK(hipMemcpyAsync(h->d_xtab, xtab.data(), sizeof(XTab) * xtab.size(), hipMemcpyHostToDevice, h->stream));
  ORB_HIP_CHECK(hipMemcpyAsync(h->d_ytab, ytab.data(), sizeof(YTab) * ytab.size(), hipMemcpyHostToDevice, h->stream));
  ORB_HIP_CHECK(hipMemcpyAsync(h->d_fast_items, fw.data(), sizeof(FastWork) * fw.size(), hipMemcpyHostToDevice, h->stream));
  ORB_HIP_CHECK(hipMemcpyAsync(h->d_blur_items, bi.data(), sizeof(BlurItem) * bi.size(), hipMemcpyHostToDevice, h->stream));
  {
    // tables of the tiled resize kernel
    std::vector<uint32_t> xtp;
    std::vector<int2> ranges;
    h->fuse01 = false;
    for (int l = 1; l < P.nlevels; l++) {
      const LevelGeom& L = P.lv[l];
      const LevelGeom& S = P.lv[l - 1];
      OrbExtractor::TileInfo ti;
      ti.xtp_off = (int)xtp.size();
      const int TC = pyr_tile_cols();
      const int TRv[2] = {pyr_tile_rows(), pyr_tile_rows_small()};
      ti.ngx = (L.pitch + TC - 1) / TC;
      auto refl = [](int i, int n) { i = i < 0 ? -i : i; return i >= n ? 2 * n - 2 - i : i; };
      const int ncol = ti.ngx * TC;
      std::vector<int2> xr(ti.ngx, make_int2(1 << 30, -1));
      bool weights_ok = true;
      for (int c = 0; c < ncol; c++) {
        const int px = c - kXOff;
        uint32_t e = 0xFFFF0000u;
        if (c < L.pitch && px >= -kEdge && px < L.w + kEdge) {
          const XTab& t = xtab[L.xtab_off + refl(px, L.w)];
          // the tiled kernel rebuilds the first weight as 2048 - a1.  cvRound((1.f - fx) * 2048)
          // can differ from that only where 1.f - fx is inexact in float, i.e. in column 0 of a
          // level, and then only if fx * 2048 lies within 2^-14 of a half-integer, which a
          // fraction with a denominator below 8192 cannot do; a level where it happens anyway
          // takes the generic kernel, which reads both weights
          if ((int)t.a0 + (int)t.a1 != 2048) weights_ok = false;
          e = (uint32_t)t.sx | ((uint32_t)(uint16_t)t.a1 << 16);
          int2& r = xr[c / TC];
          r.x = std::min(r.x, (int)t.sx); r.y = std::max(r.y, (int)t.sx + 1);
        }
        xtp.push_back(e);
      }
      ti.ok = weights_ok;
      for (int2& r : xr) {
        if (r.y < 0) { r.x = 0; r.y = 0; }
        const int a0 = (r.x + kXOff) & ~15;
        const int bytes = (r.y + kXOff - a0 + 16) & ~15;  // staged bytes per row (16-byte pieces)
        ti.src_pitch = std::max(ti.src_pitch, bytes);
        if (bytes > pyr_tile_max_src_bytes() || r.y + kXOff + 16 > S.pitch) ti.ok = false;
      }
      ti.xr_off = (int)ranges.size(); ranges.insert(ranges.end(), xr.begin(), xr.end());
      for (int v = 0; v < 2; v++) {
        const int TR = TRv[v];
        // tiles of the level's own rows (the frame rows are stored as copies of their mirror images)
        ti.ngy[v] = (L.h + TR - 1) / TR;
        std::vector<int2> yr(ti.ngy[v], make_int2(1 << 30, -1));
        for (int r = 0; r < L.h; r++) {
          const YTab& t = ytab[L.ytab_off + r];
          int2& q = yr[r / TR];
          q.x = std::min(q.x, (int)std::min(t.sy0, t.sy1)); q.y = std::max(q.y, (int)std::max(t.sy0, t.sy1));
        }
        for (int2& r : yr) {
          ti.src_rows[v] = std::max(ti.src_rows[v], r.y - r.x + 1);
          if (r.y - r.x + 1 > pyr_tile_max_src_rows()) ti.ok = false;
        }
        ti.yr_off[v] = (int)ranges.size(); ranges.insert(ranges.end(), yr.begin(), yr.end());
        if (l == 1 && v == 0) {
          // level 0 written by the tiles of level 1: every image row must belong to exactly one tile
          // row (the first that reads it), every 16-byte piece of a row to one tile column
          bool fuse = ti.ok && yr[0].x == 0;
          for (size_t g = 0; g + 1 < yr.size(); g++) fuse = fuse && yr[g + 1].x >= yr[g].x && yr[g + 1].x <= yr[g].y + 1;
          for (size_t g = 0; g + 1 < xr.size(); g++) fuse = fuse && xr[g + 1].x >= xr[g].x && xr[g].y >= xr[g].x;
          h->fuse01 = fuse;
          h->own_off = (int)ranges.size();
          for (const int2& r : yr) ranges.push_back(make_int2(r.x, 0));
          ranges.push_back(make_int2(S.h, 0));
        }
      }
      h->tile[l] = ti;
    }
    // pairs of levels for launches of a few frames (small tiles): what a tile of level l + 1 needs of
    // level l, and that of level l - 1
    for (int l = 1; l < kMaxLevels; l++) h->pair[l] = OrbExtractor::PairInfo();
    for (int l = 1; l + 1 < P.nlevels; l += 2) {
      const OrbExtractor::TileInfo &t1 = h->tile[l], &t2 = h->tile[l + 1];
      OrbExtractor::PairInfo pi;
      if (!t1.ok || !t2.ok || t2.ngx != 1) continue;
      const LevelGeom &L0 = P.lv[l - 1], &L1 = P.lv[l];
      pi.spitch0 = (L0.w + 1 + 15) & ~15;
      pi.pitchB = (L1.w + 4 + 15) & ~15;
      int rows0 = 0, rows1 = 0;
      for (int g = 0; g < t2.ngy[1]; g++) {
        const int2 y2 = ranges[t2.yr_off[1] + g];
        const YTab &ta = ytab[L1.ytab_off + y2.x], &tb = ytab[L1.ytab_off + y2.y];
        rows1 = std::max(rows1, y2.y - y2.x + 1);
        rows0 = std::max(rows0, (int)std::max(tb.sy0, tb.sy1) - (int)std::min(ta.sy0, ta.sy1) + 1);
      }
      pi.offB = rows0 * pi.spitch0;
      pi.lds = std::max(pi.offB + rows1 * pi.pitchB, t1.src_rows[1] * t1.src_pitch);
      pi.ok = rows1 <= 64 && pi.lds <= 60 * 1024;
      h->pair[l] = pi;
    }
    ORB_HIP_CHECK(dev_alloc(h, &h->d_xtp, sizeof(uint32_t) * std::max<size_t>(xtp.size(), 4)));
    ORB_HIP_CHECK(dev_alloc(h, &h->d_ranges, sizeof(int2) * std::max<size_t>(ranges.size(), 1)));
    ORB_HIP_CHECK(hipMemcpyAsync(h->d_xtp, xtp.data(), sizeof(uint32_t) * xtp.size(), hipMemcpyHostToDevice, h->stream));
    ORB_HIP_CHECK(hipMemcpyAsync(h->d_ranges, ranges.data(), sizeof(int2) * ranges.size(), hipMemcpyHostToDevice, h->stream));
    ORB_HIP_CHECK(hipStreamSynchronize(h->stream));  // xtp and ranges go out of scope
  }
  ORB_HIP_CHECK(dev_alloc(h, &h->d_cand, sizeof(uint32_t) * (size_t)P.cand_total * cap));
  ORB_HIP_CHECK(dev_alloc(h, &h->d_pnode, sizeof(uint32_t) * (size_t)P.cand_total * cap));
  ORB_HIP_CHECK(dev_alloc(h, &h->d_sel, sizeof(uint32_t) * (size_t)P.sel_total * cap));
  ORB_HIP_CHECK(dev_alloc(h, &h->d_cand_cnt, sizeof(int) * (size_t)P.nlevels * cap));
  ORB_HIP_CHECK(dev_alloc(h, &h->d_sel_cnt, sizeof(int) * (size_t)P.nlevels * cap));
  ORB_HIP_CHECK(hipMemsetAsync(h->d_sel_cnt, 0, sizeof(int) * (size_t)P.nlevels * cap, h->stream));
  ORB_HIP_CHECK(hipStreamSynchronize(h->stream));  // the host vectors above go out of scope
  h->plan = P; h->fast_items = fi; h->blur_items = bi;
  h->fast_items_tall = (int)n_tall;
  h->fast_items_few = (int)fi_few.size();
  h->ot_slots = 0;
  for (int l = 0; l < P.nlevels; l++)
    h->ot_slots = std::max(h->ot_slots, std::max(P.lv[l].quota + 3, 4 * P.lv[l].nIni) + P.lv[l].nIni + 8);
  h->have_plan = true; h->cap_frames = cap; h->last_frames = 0;
  return ORBHIP_OK;
}

// frame0: the batch's first frame takes slot frame0 of the handle's internal buffers (a host-buffer
// batch that is processed in pieces keeps every frame's intermediate results)
int enqueue(OrbExtractor* h, int nframes, const uint8_t* d_images, size_t stride, size_t frame_stride, int max_kp,
            OrbKeyPoint* d_kps, uint8_t* d_desc, int* d_nout, hipStream_t st, int frame0 = 0,
            hipEvent_t after_pyramid = nullptr) {
  const Plan& P = h->plan;
  uint8_t* const b_pyr = h->d_pyr + (size_t)frame0 * P.slab;
  uint8_t* const b_blur = h->d_blur + (size_t)frame0 * P.slab;
  uint32_t* const b_cand = h->d_cand + (size_t)frame0 * P.cand_total;
  uint32_t* const b_pnode = h->d_pnode + (size_t)frame0 * P.cand_total;
  uint32_t* const b_sel = h->d_sel + (size_t)frame0 * P.sel_total;
  int* const b_cand_cnt = h->d_cand_cnt + (size_t)frame0 * P.nlevels;
  int* const b_sel_cnt = h->d_sel_cnt + (size_t)frame0 * P.nlevels;
  const bool prof = h->profiling;
  hipEvent_t* ev = nullptr;
  if (prof) {
    const size_t need = (size_t)(h->prof_calls + 1) * (ORBHIP_NUM_STAGES + 1);
    while (h->ev.size() < need) {
      hipEvent_t e;
      ORB_HIP_CHECK(hipEventCreate(&e));
      h->ev.push_back(e);
    }
    ev = h->ev.data() + (size_t)h->prof_calls * (ORBHIP_NUM_STAGES + 1);
  }
  if (prof) ORB_HIP_CHECK(hipEventRecord(ev[0], st));
  const int cn = nframes;
  uint8_t* const c_pyr = b_pyr;
  const uint8_t* const c_img = d_images;
  int* const c_cnt = b_cand_cnt;
  // (the level-0 kernel also clears the frames' candidate counters)
  // (ORBHIP_PYR_FUSE01=0 in the environment keeps levels 0 and 1 in launches of their own: timing comparisons)
  static const bool fuse_off = getenv("ORBHIP_PYR_FUSE01") && getenv("ORBHIP_PYR_FUSE01")[0] == '0';
  const bool fused01 = h->fuse01 && h->tile[1].ok && !fuse_off && h->channels == 1 && nframes > kSmallBatchFrames && P.nlevels > 1;
  if (fused01) {
    const OrbExtractor::TileInfo& ti = h->tile[1];
    launch_pyr_level01(c_img, stride, frame_stride, c_pyr, P, h->d_xtp + ti.xtp_off, h->d_ytab, h->d_ranges + ti.xr_off,
                       ti.ngx, h->d_ranges + ti.yr_off[0], ti.ngy[0], h->d_ranges + h->own_off, pyr_tile_rows(),
                       ti.src_rows[0], ti.src_pitch, c_cnt, cn, st);
  }
  else if (h->channels == 1) launch_pyr_level0(c_img, stride, frame_stride, c_pyr, P, c_cnt, cn, st);
  else launch_pyr_level0_color(c_img, stride, frame_stride, h->channels, h->rgb_order, h->gray, c_pyr, P, c_cnt, cn, st);
  for (int l = fused01 ? 2 : 1; l < P.nlevels; l++) {
    const OrbExtractor::TileInfo& ti = h->tile[l];
    // (batches: levels 5 + 6, or 3 + 4 and 5 + 6, in one launch each make a 256-frame step 0.7 % and 1.5 % slower)
    if (nframes <= kSmallBatchFrames && h->pair[l].ok) {
      const OrbExtractor::TileInfo& t2 = h->tile[l + 1];
      const OrbExtractor::PairInfo& pi = h->pair[l];
      launch_pyr_resize_dual(c_pyr, P, l, h->d_xtp + ti.xtp_off, h->d_xtp + t2.xtp_off, h->d_ytab,
                             h->d_ranges + ti.xr_off, ti.ngx, h->d_ranges + ti.yr_off[1], ti.ngy[1],
                             h->d_ranges + t2.yr_off[1], t2.ngy[1], pyr_tile_rows_small(), ti.src_pitch, pi.spitch0,
                             pi.pitchB, pi.offB, pi.lds, cn, st);
      l++;  // level l + 1 is done as well
      continue;
    }
    if (ti.ok) {
      const int v = nframes <= kSmallBatchFrames ? 1 : 0;
      launch_pyr_resize_tile(c_pyr, P, l, h->d_xtp + ti.xtp_off, h->d_ytab, h->d_ranges + ti.xr_off, ti.ngx,
                             h->d_ranges + ti.yr_off[v], ti.ngy[v], v ? pyr_tile_rows_small() : pyr_tile_rows(),
                             ti.src_rows[v], ti.src_pitch, cn, st);
    }
    else
      launch_pyr_resize(c_pyr, P, l, h->d_xtab, h->d_ytab, cn, st);
  }
  if (prof) ORB_HIP_CHECK(hipEventRecord(ev[1], st));
  if (after_pyramid) ORB_HIP_CHECK(hipEventRecord(after_pyramid, st));
  {
    const int n_all = (int)h->fast_items.size(), n_batch = n_all - h->fast_items_few;
    if (nframes <= kFewFrames)
      launch_fast(b_pyr, P, h->d_fast_items + n_batch, h->fast_items_few, -1, b_cand, b_cand_cnt, h->iniTh, h->minTh,
                  nframes, st);
    else
      launch_fast(b_pyr, P, h->d_fast_items, n_batch, nframes <= kSmallBatchFrames ? -1 : h->fast_items_tall, b_cand,
                  b_cand_cnt, h->iniTh, h->minTh, nframes, st);
  }
  if (prof) ORB_HIP_CHECK(hipEventRecord(ev[2], st));
  ORB_HIP_CHECK(launch_octree(b_cand, b_cand_cnt, b_pnode, b_sel, b_sel_cnt, P, h->ot_slots, nframes, st));
  if (prof) ORB_HIP_CHECK(hipEventRecord(ev[3], st));
  if (!h->fused_blur) launch_blur(b_pyr, b_blur, P, h->d_blur_items, (int)h->blur_items.size(), h->taps, nframes, st);
  if (prof) ORB_HIP_CHECK(hipEventRecord(ev[4], st));
  launch_orient_desc(b_pyr, b_blur, b_sel, b_sel_cnt, P, d_kps, d_desc, d_nout, max_kp, nframes,
                     h->fused_blur ? h->taps : nullptr, st);
  h->blur_stale = h->fused_blur;
  if (prof) { ORB_HIP_CHECK(hipEventRecord(ev[5], st)); h->prof_calls++; }
  ORB_HIP_CHECK(hipGetLastError());
  h->last_frames = frame0 + nframes;
  return ORBHIP_OK;
}

// The two pieces of OpenCV arithmetic on this path that changed between releases.
//  * cv::GaussianBlur, 8-bit fixed-point path: the 7-tap sigma-2 kernel in 8.8 fixed point.  From
//    4.5 on the rounding error of a tap is carried to the next one and the centre tap takes the
//    remainder, so the taps sum to 256: 18 34 48 56 48 34 18.  4.0 - 4.4 (and 3.4 since the
//    fixed-point path appeared) round every tap on its own: 18 34 49 55 49 34 18, sum 257,
//    results saturated.
//  * cv::cvtColor to grey, 8-bit: 15-bit weights 9798 / 19235 / 3735 from 4.0 on, 14-bit weights
//    4899 / 9617 / 1868 before.
void set_opencv_tables(OrbExtractor* h, int major, int minor) {
  double k[7], s = 0;
  for (int i = 0; i < 7; i++) { const double x = i - 3.0; k[i] = exp(-0.125 * x * x); s += k[i]; }
  s = 1.0 / s;
  if (major > 4 || (major == 4 && minor >= 5)) {
    double err = 0; int acc = 0;
    for (int i = 0; i < 3; i++) {
      const double adj = k[i] * s * 256.0 + err;
      const int v = cv_round(adj);
      err = adj - v; h->taps[i] = v; acc += v;
    }
    h->taps[3] = 256 - 2 * acc;
  } else {
    for (int i = 0; i < 4; i++) h->taps[i] = cv_round(k[i] * s * 256.0);
  }
  if (major >= 4) { h->gray[0] = 9798; h->gray[1] = 19235; h->gray[2] = 3735; h->gray[3] = 15; }
  else { h->gray[0] = 4899; h->gray[1] = 9617; h->gray[2] = 1868; h->gray[3] = 14; }
}

struct DeviceGuard {
  int prev = -1;
  explicit DeviceGuard(int dev) { hipGetDevice(&prev); if (prev != dev) hipSetDevice(dev); else prev = -1; }
  ~DeviceGuard() { if (prev >= 0) hipSetDevice(prev); }
};

}  // namespace

extern "C" {

const char* orb_last_error(void) { return g_err.c_str(); }

int orb_device_count(void) {
  int n = 0;
  ORB_HIP_CHECK(hipGetDeviceCount(&n));
  return n;
}

int orb_extractor_create(int nfeatures, float scale_factor, int nlevels, int ini_th_fast, int min_th_fast,
                         int device_id, OrbExtractor** out) {
  if (!out) { set_error("out is null"); return ORBHIP_ERR_ARG; }
  *out = nullptr;
  if (nlevels < 1 || nlevels > kMaxLevels) { set_error("nlevels must be in 1.." + std::to_string(kMaxLevels)); return ORBHIP_ERR_ARG; }
  if (nfeatures < 1 || !(scale_factor > 1.0f)) { set_error("nfeatures must be positive and scaleFactor > 1"); return ORBHIP_ERR_ARG; }
  int ndev = 0;
  ORB_HIP_CHECK(hipGetDeviceCount(&ndev));
  if (device_id < 0 || device_id >= ndev) { set_error("no such device"); return ORBHIP_ERR_ARG; }
  OrbExtractor* h = new OrbExtractor();
  h->nfeatures = nfeatures; h->nlevels = nlevels; h->device = device_id;
  h->iniTh = std::min(std::max(ini_th_fast, 0), 255);  // cv::FAST clamps its threshold
  h->minTh = std::min(std::max(min_th_fast, 0), 255);
  h->scaleFactor = scale_factor;
  // scale tables and per-level quotas, reference src/ORBextractor.cc:480-528
  h->sf[0] = 1.f; h->sigma2[0] = 1.f;
  for (int i = 1; i < nlevels; i++) {
    h->sf[i] = (float)(h->sf[i - 1] * h->scaleFactor);
    h->sigma2[i] = h->sf[i] * h->sf[i];
  }
  for (int i = 0; i < nlevels; i++) { h->inv_sf[i] = 1.f / h->sf[i]; h->inv_sigma2[i] = 1.f / h->sigma2[i]; }
  const float factor = (float)(1.0 / h->scaleFactor);
  float want = nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nlevels));
  int sum = 0;
  for (int l = 0; l < nlevels - 1; l++) { h->quota[l] = cv_round(want); sum += h->quota[l]; want *= factor; }
  h->quota[nlevels - 1] = std::max(nfeatures - sum, 0);
  // row half-widths of the orientation disc, :544-572
  {
    const int R = 15;
    const int vmax = (int)floor(R * sqrt(2.f) / 2 + 1), vmin = (int)ceil(R * sqrt(2.f) / 2);
    for (int v = 0; v <= vmax; ++v) h->umax[v] = cv_round(sqrt((double)R * R - v * v));
    for (int v = R, v0 = 0; v >= vmin; --v) {
      while (h->umax[v0] == h->umax[v0 + 1]) ++v0;
      h->umax[v] = v0; ++v0;
    }
    static const int expect[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};
    if (memcmp(expect, h->umax, sizeof(expect)) != 0) { delete h; set_error("umax table mismatch"); return ORBHIP_ERR_UNSUPPORTED; }
  }
  set_opencv_tables(h, 4, 5);
  DeviceGuard g(device_id);
  hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
  if (e != hipSuccess) { delete h; return hip_fail(e, "hipStreamCreate"); }
  e = hipEventCreateWithFlags(&h->done_ev, hipEventDisableTiming);
  if (e != hipSuccess) { hipStreamDestroy(h->stream); delete h; return hip_fail(e, "hipEventCreate"); }
  {
    int pools = 0;
    const char* env = getenv("ORBHIP_ASYNC_ALLOC");  // 0: plain hipMalloc / hipFree (for comparisons)
    if (hipDeviceGetAttribute(&pools, hipDeviceAttributeMemoryPoolsSupported, device_id) != hipSuccess) pools = 0;
    h->async_alloc = pools != 0 && !(env && env[0] == '0');
  }
  e = upload_orient_tables();
  if (e != hipSuccess) { hipEventDestroy(h->done_ev); hipStreamDestroy(h->stream); delete h; return hip_fail(e, "hipMemcpyToSymbol"); }
  *out = h;
  return ORBHIP_OK;
}

void orb_extractor_destroy(OrbExtractor* h) {
  if (!h) return;
  DeviceGuard g(h->device);
  wait_own(h);
  free_plan_buffers(h);
  dev_free(h, h->d_img); dev_free(h, h->d_desc);  // (keypoints and counts live in the descriptors' allocation)
  hipStreamSynchronize(h->stream);  // the frees above are ordered on the handle's stream
  if (h->h_img) hipHostFree(h->h_img);
  { drop_graph(h); if (h->h_out) hipHostFree(h->h_out); }
  delete h->pool;
  for (hipEvent_t e : h->ev) if (e) hipEventDestroy(e);
  for (hipEvent_t e : h->pipe_ev) if (e) hipEventDestroy(e);
  if (h->done_ev) hipEventDestroy(h->done_ev);
  for (hipEvent_t e : {h->ev_fork, h->ev_stage, h->ev_join}) if (e) hipEventDestroy(e);
  if (h->half) hipStreamDestroy(h->half);
  if (h->up) hipStreamDestroy(h->up);
  if (h->down) hipStreamDestroy(h->down);
  hipStreamDestroy(h->stream);
  delete h;
}

int orb_extractor_get_scale_tables(const OrbExtractor* h, float* scale, float* inv_scale, float* sigma2,
                                   float* inv_sigma2) {
  if (!h) { set_error("null handle"); return ORBHIP_ERR_ARG; }
  for (int i = 0; i < h->nlevels; i++) {
    if (scale) scale[i] = h->sf[i];
    if (inv_scale) inv_scale[i] = h->inv_sf[i];
    if (sigma2) sigma2[i] = h->sigma2[i];
    if (inv_sigma2) inv_sigma2[i] = h->inv_sigma2[i];
  }
  return h->nlevels;
}

int orb_extractor_get_level_quotas(const OrbExtractor* h, int* quotas, int* umax) {
  if (!h) { set_error("null handle"); return ORBHIP_ERR_ARG; }
  if (quotas) for (int i = 0; i < h->nlevels; i++) quotas[i] = h->quota[i];
  if (umax) for (int i = 0; i < 16; i++) umax[i] = h->umax[i];
  return h->nlevels;
}

int orb_extractor_max_keypoints(OrbExtractor* h, int rows, int cols) {
  if (!h) { set_error("null handle"); return ORBHIP_ERR_ARG; }
  Plan P; std::vector<XTab> xt; std::vector<YTab> yt; std::vector<FastItem> fi, fi_few; std::vector<BlurItem> bi;
  int rc = build_plan(h, rows, cols, P, xt, yt, fi, fi_few, bi);
  if (rc != ORBHIP_OK) return rc;
  return P.sel_total;
}

int orb_extract_batch_device(OrbExtractor* h, int n_frames, const uint8_t* d_images, int rows, int cols,
                             size_t stride, size_t frame_stride, int max_kp, OrbKeyPoint* d_keypoints,
                             uint8_t* d_descriptors, int* d_n_out, void* stream) {
  if (!h || !d_n_out) { set_error("null argument"); return ORBHIP_ERR_ARG; }
  if (n_frames <= 0) return ORBHIP_OK;
  if (rows <= 0 || cols <= 0) {  // empty image: the reference returns without touching its outputs
    DeviceGuard g(h->device);
    ORB_HIP_CHECK(hipMemsetAsync(d_n_out, 0, sizeof(int) * n_frames, (hipStream_t)stream));
    return ORBHIP_OK;
  }
  if (!d_images || !d_keypoints || !d_descriptors || max_kp <= 0 || stride < (size_t)cols * h->channels) {
    set_error("null buffer, max_kp <= 0 or stride < cols * channels"); return ORBHIP_ERR_ARG;
  }
  DeviceGuard g(h->device);
  int rc = ensure_plan(h, rows, cols, n_frames);
  if (rc != ORBHIP_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
  // Stage times are taken with the whole batch on one stream (kernels that share the GPU with another
  // stage's would each read longer); a stream under capture is left to its owner.
  const bool split = h->split_batches && n_frames >= kSplitMinFrames && !h->profiling && cap == hipStreamCaptureStatusNone;
  if (split) {
    // the second stream runs at the priority of the caller's: a caller that puts the extraction above
    // other work on the GPU means both halves
    int prio = 0;
    if (hipStreamGetPriority(st, &prio) != hipSuccess) { (void)hipGetLastError(); prio = 0; }
    if (h->half && prio != h->half_prio) {
      ORB_HIP_CHECK(hipStreamSynchronize(h->half));
      ORB_HIP_CHECK(hipStreamDestroy(h->half));
      h->half = nullptr;
    }
    if (!h->half) {
      ORB_HIP_CHECK(hipStreamCreateWithPriority(&h->half, hipStreamNonBlocking, prio));
      h->half_prio = prio;
    }
    if (!h->ev_fork) {
      ORB_HIP_CHECK(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
      ORB_HIP_CHECK(hipEventCreateWithFlags(&h->ev_stage, hipEventDisableTiming));
      ORB_HIP_CHECK(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
    }
    const int n1 = n_frames / 2, n2 = n_frames - n1;  // (a quarter, three eighths, five eighths first: none is faster)
    // the second half starts after whatever precedes this call on the caller's stream ...
    ORB_HIP_CHECK(hipEventRecord(h->ev_fork, st));
    ORB_HIP_CHECK(hipStreamWaitEvent(h->half, h->ev_fork, 0));
    rc = enqueue(h, n1, d_images, stride, frame_stride, max_kp, d_keypoints, d_descriptors, d_n_out, st, 0, h->ev_stage);
    if (rc != ORBHIP_OK) return rc;
    // ... and after the first half's pyramid: from then on the halves are a stage apart (held back
    // until the first half's FAST has ended, or not at all, the pair is slower: 1.28 / 1.22 against
    // 1.21 ms per 256 frames)
    ORB_HIP_CHECK(hipStreamWaitEvent(h->half, h->ev_stage, 0));
    rc = enqueue(h, n2, d_images + (size_t)n1 * frame_stride, stride, frame_stride, max_kp, d_keypoints + (size_t)n1 * max_kp,
                 d_descriptors + (size_t)n1 * max_kp * 32, d_n_out + n1, h->half, n1);
    if (rc != ORBHIP_OK) return rc;
    ORB_HIP_CHECK(hipEventRecord(h->ev_join, h->half));
    ORB_HIP_CHECK(hipStreamWaitEvent(st, h->ev_join, 0));
  } else {
    rc = enqueue(h, n_frames, d_images, stride, frame_stride, max_kp, d_keypoints, d_descriptors, d_n_out, st);
    if (rc != ORBHIP_OK) return rc;
  }
  // where this call's work ends on the caller's stream (a stream that is being captured into a graph
  // takes no such event: the capture's owner orders the replays)
  if (cap == hipStreamCaptureStatusNone) {
    ORB_HIP_CHECK(hipEventRecord(h->done_ev, (hipStream_t)stream));
    h->done_pending = true;
  }
  return ORBHIP_OK;
}

int orb_extractor_set_input_format(OrbExtractor* h, int channels, int rgb_order) {
  if (!h) { set_error("null handle"); return ORBHIP_ERR_ARG; }
  if (channels != 1 && channels != 3 && channels != 4) { set_error("channels must be 1, 3 or 4"); return ORBHIP_ERR_ARG; }
  h->channels = channels;
  h->rgb_order = rgb_order != 0;
  return ORBHIP_OK;
}

int orb_extractor_set_opencv_compat(OrbExtractor* h, int major, int minor) {
  if (!h) { set_error("null handle"); return ORBHIP_ERR_ARG; }
  if (major < 3 || (major == 3 && minor < 4) || minor < 0) {
    set_error("OpenCV before 3.4 blurs 8-bit images in floating point; that path is not restated here");
    return ORBHIP_ERR_UNSUPPORTED;
  }
  set_opencv_tables(h, major, minor);
  return ORBHIP_OK;
}

int orb_extractor_set_blur_fusion(OrbExtractor* h, int fused) {
  if (!h) { set_error("null handle"); return ORBHIP_ERR_ARG; }
  h->fused_blur = fused != 0;
  return ORBHIP_OK;
}

int orb_extractor_set_batch_split(OrbExtractor* h, int enable) {
  if (!h) { set_error("null handle"); return ORBHIP_ERR_ARG; }
  h->split_batches = enable != 0;
  return ORBHIP_OK;
}

int orb_extractor_sync(OrbExtractor* h) {
  if (!h) { set_error("null handle"); return ORBHIP_ERR_ARG; }
  DeviceGuard g(h->device);
  ORB_HIP_CHECK(wait_own(h));
  return ORBHIP_OK;
}

int orb_extractor_set_profiling(OrbExtractor* h, int enable) {
  if (!h) { set_error("null handle"); return ORBHIP_ERR_ARG; }
  h->profiling = enable != 0;
  h->prof_calls = 0;
  return ORBHIP_OK;
}

int orb_extractor_stage_times(OrbExtractor* h, float* ms) {
  if (!h || !ms) { set_error("null argument"); return ORBHIP_ERR_ARG; }
  if (h->prof_calls == 0) { set_error("no profiled call has run"); return ORBHIP_ERR_ARG; }
  DeviceGuard g(h->device);
  for (int i = 0; i < ORBHIP_NUM_STAGES; i++) ms[i] = 0.f;
  for (int c = 0; c < h->prof_calls; c++) {
    hipEvent_t* ev = h->ev.data() + (size_t)c * (ORBHIP_NUM_STAGES + 1);
    ORB_HIP_CHECK(hipEventSynchronize(ev[ORBHIP_NUM_STAGES]));
    for (int i = 0; i < ORBHIP_NUM_STAGES; i++) {
      float t = 0.f;
      ORB_HIP_CHECK(hipEventElapsedTime(&t, ev[i], ev[i + 1]));
      ms[i] += t;
    }
  }
  for (int i = 0; i < ORBHIP_NUM_STAGES; i++) ms[i] /= (float)h->prof_calls;
  h->prof_calls = 0;
  return ORBHIP_NUM_STAGES;
}

int orb_extract_batch(OrbExtractor* h, int n_frames, const uint8_t* images, int rows, int cols, size_t stride,
                      size_t frame_stride, int max_kp, OrbKeyPoint* keypoints, uint8_t* descriptors, int* n_out) {
  if (!h || !n_out) { set_error("null argument"); return ORBHIP_ERR_ARG; }
  if (n_frames <= 0) return ORBHIP_OK;
  if (rows <= 0 || cols <= 0) { for (int k = 0; k < n_frames; k++) n_out[k] = 0; return ORBHIP_OK; }
  const size_t row_bytes = (size_t)cols * h->channels;
  if (!images || !keypoints || !descriptors || max_kp <= 0 || stride < row_bytes) {
    set_error("null buffer, max_kp <= 0 or stride < cols * channels"); return ORBHIP_ERR_ARG;
  }
  DeviceGuard g(h->device);
  int rc = ensure_plan(h, rows, cols, n_frames);
  if (rc != ORBHIP_OK) return rc;
  // a device-buffer call on another stream may still be using the handle's buffers
  if (h->done_pending) ORB_HIP_CHECK(hipStreamWaitEvent(h->stream, h->done_ev, 0));
  const size_t img_bytes = (size_t)rows * row_bytes * n_frames;
  if (img_bytes > h->d_img_bytes) {
    ORB_HIP_CHECK(wait_own(h));
    drop_graph(h);
    dev_free(h, h->d_img); h->d_img = nullptr; h->d_img_bytes = 0;
    ORB_HIP_CHECK(dev_alloc(h, &h->d_img, img_bytes));
    h->d_img_bytes = img_bytes;
  }
  const size_t need = (size_t)n_frames * max_kp;
  if (need > h->out_cap || n_frames > h->out_frames) {
    ORB_HIP_CHECK(wait_own(h));
    drop_graph(h);
    // one allocation, laid out like the pinned staging (descriptors, keypoints, counts), so that the
    // results come back in one copy
    const size_t cap = std::max(need, h->out_cap), capf = (size_t)std::max(n_frames, h->out_frames);
    dev_free(h, h->d_desc);
    h->d_desc = nullptr; h->out_cap = 0; h->out_frames = 0;
    ORB_HIP_CHECK(dev_alloc(h, &h->d_desc, cap * (32 + sizeof(OrbKeyPoint)) + sizeof(int) * capf));
    h->out_cap = cap; h->out_frames = (int)capf;
  }
  // (the layout follows this call's sizes, whatever the allocation's capacity)
  h->d_kps = reinterpret_cast<OrbKeyPoint*>(h->d_desc + need * 32);
  h->d_nout = reinterpret_cast<int*>(h->d_desc + need * (32 + sizeof(OrbKeyPoint)));
  hipStream_t st = h->stream;
  // Host images go through a pinned staging buffer, packed to rows of `row_bytes` bytes: one
  // DMA for the batch (a strided copy from pageable memory is far slower than the kernels).
  if (img_bytes > h->h_img_bytes) {
    ORB_HIP_CHECK(hipStreamSynchronize(h->stream));
    drop_graph(h);
    if (h->h_img) hipHostFree(h->h_img);
    h->h_img = nullptr; h->h_img_bytes = 0;
    ORB_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&h->h_img), img_bytes, hipHostMallocDefault));
    h->h_img_bytes = img_bytes;
  }
  const size_t frame_bytes = (size_t)rows * row_bytes;
  auto pack = [&](int k) {
    const uint8_t* src = images + (size_t)k * frame_stride;
    uint8_t* dst = h->h_img + (size_t)k * frame_bytes;
    if (stride == row_bytes) memcpy(dst, src, frame_bytes);
    else for (int r = 0; r < rows; r++) memcpy(dst + (size_t)r * row_bytes, src + (size_t)r * stride, row_bytes);
  };
  const bool small = n_frames <= kGraphMaxFrames && img_bytes < kPoolMinBytes;
  // Large batches go through in pieces: while the kernels work on one piece, the next is packed
  // and uploaded and the results of the one before come back, each on a stream of its own, so
  // that the call takes about as long as the slowest of the three instead of their sum.
  const bool pieces = n_frames >= kPipeMinFrames && !h->profiling;
  if (pieces) {
    const size_t out_bytes_p = need * (sizeof(OrbKeyPoint) + 32) + sizeof(int) * (size_t)n_frames;
    if (out_bytes_p > h->h_out_bytes) {
      ORB_HIP_CHECK(hipStreamSynchronize(h->stream));
      { drop_graph(h); if (h->h_out) hipHostFree(h->h_out); }
      h->h_out = nullptr; h->h_out_bytes = 0;
      ORB_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&h->h_out), out_bytes_p, hipHostMallocDefault));
      h->h_out_bytes = out_bytes_p;
    }
    if (!h->up) ORB_HIP_CHECK(hipStreamCreateWithFlags(&h->up, hipStreamNonBlocking));
    if (!h->down) ORB_HIP_CHECK(hipStreamCreateWithFlags(&h->down, hipStreamNonBlocking));
    const int npieces = std::min(kPipePieces, n_frames / 16);
    while ((int)h->pipe_ev.size() < 3 * npieces) {
      hipEvent_t e;
      ORB_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      h->pipe_ev.push_back(e);
    }
    if (!h->pool) h->pool = new HostCopyPool(pool_threads() - 1);
    uint8_t* const sp_desc = h->h_out;
    OrbKeyPoint* const sp_kps = reinterpret_cast<OrbKeyPoint*>(h->h_out + need * 32);
    int* const sp_n = reinterpret_cast<int*>(h->h_out + need * (32 + sizeof(OrbKeyPoint)));
    // the streams of this call start after whatever the handle's stream still has to do
    ORB_HIP_CHECK(hipStreamSynchronize(st));
    int status = ORBHIP_OK;
    auto piece = [&](int p, int& f0, int& fn) {
      f0 = (int)((long long)n_frames * p / npieces);
      fn = (int)((long long)n_frames * (p + 1) / npieces) - f0;
    };
    // The host threads pack the pieces one after the other from the start of the call; this
    // thread only waits for a piece, queues its upload, kernels and download, and hands the
    // unpacking of a finished piece back to the host threads.
    PoolDrain drain{h->pool};
    std::vector<HostCopyPool::JobRef> packed(npieces);
    for (int p = 0; p < npieces; p++) {
      int f0, fn;
      piece(p, f0, fn);
      packed[p] = h->pool->submit(fn, [&pack, f0](int i) { pack(f0 + i); });
    }
    // waits for piece p's download and queues the copy of the valid part of its frames to the
    // caller's arrays
    auto deliver = [&](int p) -> int {
      int f0, fn;
      piece(p, f0, fn);
      hipError_t e = hipEventSynchronize(h->pipe_ev[2 * npieces + p]);
      if (e != hipSuccess) return hip_fail(e, "hipEventSynchronize");
      int r = ORBHIP_OK;
      for (int k = f0; k < f0 + fn; k++) {
        n_out[k] = sp_n[k];
        if (n_out[k] > max_kp) { set_error("max_kp too small: a frame produced " + std::to_string(n_out[k]) + " keypoints"); r = ORBHIP_ERR_CAPACITY; }
      }
      h->pool->submit(fn, [=](int i) {
        const int k = f0 + i, n = std::min(sp_n[k], max_kp);
        if (n > 0) {
          memcpy(keypoints + (size_t)k * max_kp, sp_kps + (size_t)k * max_kp, sizeof(OrbKeyPoint) * n);
          memcpy(descriptors + (size_t)k * max_kp * 32, sp_desc + (size_t)k * max_kp * 32, 32 * (size_t)n);
        }
      });
      return r;
    };
    for (int p = 0; p < npieces; p++) {
      int f0, fn;
      piece(p, f0, fn);
      h->pool->wait(packed[p]);
      ORB_HIP_CHECK(hipMemcpyAsync(h->d_img + (size_t)f0 * frame_bytes, h->h_img + (size_t)f0 * frame_bytes,
                                   (size_t)fn * frame_bytes, hipMemcpyHostToDevice, h->up));
      ORB_HIP_CHECK(hipEventRecord(h->pipe_ev[2 * p], h->up));
      ORB_HIP_CHECK(hipStreamWaitEvent(st, h->pipe_ev[2 * p], 0));
      rc = enqueue(h, fn, h->d_img + (size_t)f0 * frame_bytes, row_bytes, frame_bytes, max_kp, h->d_kps + (size_t)f0 * max_kp,
                   h->d_desc + (size_t)f0 * max_kp * 32, h->d_nout + f0, st, f0);
      if (rc != ORBHIP_OK) { wait_own(h); return rc; }
      ORB_HIP_CHECK(hipEventRecord(h->pipe_ev[2 * p + 1], st));
      ORB_HIP_CHECK(hipStreamWaitEvent(h->down, h->pipe_ev[2 * p + 1], 0));
      const size_t k0 = (size_t)f0 * max_kp, kn = (size_t)fn * max_kp;
      ORB_HIP_CHECK(hipMemcpyAsync(sp_desc + k0 * 32, h->d_desc + k0 * 32, kn * 32, hipMemcpyDeviceToHost, h->down));
      ORB_HIP_CHECK(hipMemcpyAsync(sp_kps + k0, h->d_kps + k0, kn * sizeof(OrbKeyPoint), hipMemcpyDeviceToHost, h->down));
      ORB_HIP_CHECK(hipMemcpyAsync(sp_n + f0, h->d_nout + f0, sizeof(int) * fn, hipMemcpyDeviceToHost, h->down));
      ORB_HIP_CHECK(hipEventRecord(h->pipe_ev[2 * npieces + p], h->down));
      // the results of the piece before the last one enqueued are back by now, or soon
      if (p >= 2) { const int rc2 = deliver(p - 2); if (rc2 != ORBHIP_OK) status = rc2; }
    }
    for (int p = std::max(0, npieces - 2); p < npieces; p++) { const int rc2 = deliver(p); if (rc2 != ORBHIP_OK) status = rc2; }
    h->pool->wait_all();
    if (status == ORBHIP_ERR_HIP) return status;
    ORB_HIP_CHECK(hipStreamSynchronize(st));
    return status;
  }
  if (small) {
    for (int k = 0; k < n_frames; k++) pack(k);  // the upload is part of the replayed graph
  } else if (n_frames < 4 || img_bytes < kPoolMinBytes) {
    for (int k = 0; k < n_frames; k++) pack(k);
    ORB_HIP_CHECK(hipMemcpyAsync(h->d_img, h->h_img, img_bytes, hipMemcpyHostToDevice, st));
  } else {
    // large batches: the packing is shared by a few threads, and the DMA of one chunk of frames
    // runs while the next chunk is packed
    if (!h->pool) h->pool = new HostCopyPool(pool_threads() - 1);
    const int chunk = std::max(kPoolThreads, (int)((size_t)(8u << 20) / frame_bytes));
    for (int k0 = 0; k0 < n_frames; k0 += chunk) {
      const int kn = std::min(chunk, n_frames - k0);
      h->pool->parallel_for(kn, [&](int i) { pack(k0 + i); });
      ORB_HIP_CHECK(hipMemcpyAsync(h->d_img + (size_t)k0 * frame_bytes, h->h_img + (size_t)k0 * frame_bytes,
                                   (size_t)kn * frame_bytes, hipMemcpyHostToDevice, st));
    }
  }
  // results come back through pinned staging as well: one DMA for the batch, then the valid
  // part of every frame is copied to the caller's arrays
  const size_t out_bytes = need * (sizeof(OrbKeyPoint) + 32) + sizeof(int) * (size_t)n_frames;
  if (out_bytes > h->h_out_bytes) {
    ORB_HIP_CHECK(hipStreamSynchronize(h->stream));
    { drop_graph(h); if (h->h_out) hipHostFree(h->h_out); }
    h->h_out = nullptr; h->h_out_bytes = 0;
    ORB_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&h->h_out), out_bytes, hipHostMallocDefault));
    h->h_out_bytes = out_bytes;
  }
  uint8_t* s_desc = h->h_out;
  OrbKeyPoint* s_kps = reinterpret_cast<OrbKeyPoint*>(h->h_out + need * 32);
  int* s_n = reinterpret_cast<int*>(h->h_out + need * (32 + sizeof(OrbKeyPoint)));
  auto device_work = [&](bool with_upload) -> int {
    if (with_upload) ORB_HIP_CHECK(hipMemcpyAsync(h->d_img, h->h_img, img_bytes, hipMemcpyHostToDevice, st));
    int r = enqueue(h, n_frames, h->d_img, row_bytes, (size_t)rows * row_bytes, max_kp, h->d_kps, h->d_desc, h->d_nout, st);
    if (r != ORBHIP_OK) return r;
    ORB_HIP_CHECK(hipMemcpyAsync(h->h_out, h->d_desc, out_bytes, hipMemcpyDeviceToHost, st));
    return ORBHIP_OK;
  };
  const bool use_graph = small && !h->profiling;
  if (use_graph) {
    OrbExtractor::GraphKey key;
    key.n_frames = n_frames; key.rows = rows; key.cols = cols; key.max_kp = max_kp; key.channels = h->channels;
    key.rgb = h->rgb_order; key.fused = h->fused_blur;
    for (int i = 0; i < 4; i++) { key.taps[i] = h->taps[i]; key.gray[i] = h->gray[i]; }
    key.d_img = h->d_img; key.h_img = h->h_img; key.h_out = h->h_out; key.d_kps = h->d_kps; key.d_pyr = h->d_pyr;
    if (!h->graph_exec || !(key == h->graph_key)) {
      if (h->graph_exec) { hipGraphExecDestroy(h->graph_exec); h->graph_exec = nullptr; }
      hipGraph_t graph = nullptr;
      ORB_HIP_CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
      rc = device_work(true);
      hipError_t ce = hipStreamEndCapture(st, &graph);
      if (rc != ORBHIP_OK) { if (graph) hipGraphDestroy(graph); return rc; }
      if (ce != hipSuccess) return hip_fail(ce, "hipStreamEndCapture");
      ce = hipGraphInstantiate(&h->graph_exec, graph, nullptr, nullptr, 0);
      hipGraphDestroy(graph);
      if (ce != hipSuccess) { h->graph_exec = nullptr; return hip_fail(ce, "hipGraphInstantiate"); }
      h->graph_key = key;
    }
    ORB_HIP_CHECK(hipGraphLaunch(h->graph_exec, st));
    // (enqueue() ran when the graph was captured, not now: what it notes about the call is noted here)
    h->last_frames = n_frames;
    h->blur_stale = h->fused_blur;
  } else {
    // a small batch has only been packed so far (its upload is a node of the graph that is not used here)
    rc = device_work(small);
    if (rc != ORBHIP_OK) return rc;
  }
  ORB_HIP_CHECK(hipStreamSynchronize(st));
  int status = ORBHIP_OK;
  for (int k = 0; k < n_frames; k++) {
    n_out[k] = s_n[k];
    if (n_out[k] > max_kp) { set_error("max_kp too small: a frame produced " + std::to_string(n_out[k]) + " keypoints"); status = ORBHIP_ERR_CAPACITY; }
  }
  auto unpack = [&](int k) {
    const int n = std::min(s_n[k], max_kp);
    if (n > 0) {
      memcpy(keypoints + (size_t)k * max_kp, s_kps + (size_t)k * max_kp, sizeof(OrbKeyPoint) * n);
      memcpy(descriptors + (size_t)k * max_kp * 32, s_desc + (size_t)k * max_kp * 32, 32 * (size_t)n);
    }
  };
  if (h->pool && n_frames >= 4) h->pool->parallel_for(n_frames, unpack);
  else for (int k = 0; k < n_frames; k++) unpack(k);
  return status;
}

int orb_extract(OrbExtractor* h, const uint8_t* image, int rows, int cols, size_t stride, int max_kp,
                OrbKeyPoint* keypoints, uint8_t* descriptors, int* n_out) {
  return orb_extract_batch(h, 1, image, rows, cols, stride, 0, max_kp, keypoints, descriptors, n_out);
}

static int check_frame_level(OrbExtractor* h, int frame, int level) {
  if (!h) { set_error("null handle"); return ORBHIP_ERR_ARG; }
  if (!h->have_plan || h->last_frames == 0) { set_error("no extraction has run yet"); return ORBHIP_ERR_ARG; }
  if (frame < 0 || frame >= h->last_frames || level < 0 || level >= h->nlevels) { set_error("frame or level out of range"); return ORBHIP_ERR_ARG; }
  return ORBHIP_OK;
}

static int copy_level(OrbExtractor* h, const uint8_t* buf, int frame, int level, int padded, uint8_t* dst,
                      size_t dst_stride, int* width, int* height) {
  int rc = check_frame_level(h, frame, level);
  if (rc != ORBHIP_OK) return rc;
  const LevelGeom& L = h->plan.lv[level];
  const int w = padded ? L.w + 2 * kEdge : L.w, hh = padded ? L.h + 2 * kEdge : L.h;
  if (width) *width = w;
  if (height) *height = hh;
  if (!dst) return ORBHIP_OK;
  if (dst_stride < (size_t)w) { set_error("dst_stride too small"); return ORBHIP_ERR_ARG; }
  DeviceGuard g(h->device);
  ORB_HIP_CHECK(wait_own(h));
  const uint8_t* src = buf + (size_t)frame * h->plan.slab + L.off +
                       (padded ? (size_t)(kXOff - kEdge) : (size_t)kEdge * L.pitch + kXOff);
  // one plain copy of the rows as they lie in the slab, on the handle's stream, then row by row into
  // the caller's array
  std::vector<uint8_t> rows((size_t)(hh - 1) * L.pitch + w);
  ORB_HIP_CHECK(hipMemcpyAsync(rows.data(), src, rows.size(), hipMemcpyDeviceToHost, h->stream));
  ORB_HIP_CHECK(hipStreamSynchronize(h->stream));
  for (int r = 0; r < hh; r++) memcpy(dst + (size_t)r * dst_stride, rows.data() + (size_t)r * L.pitch, (size_t)w);
  return ORBHIP_OK;
}

int orb_get_pyramid_level(OrbExtractor* h, int frame, int level, int padded, uint8_t* dst, size_t dst_stride,
                          int* width, int* height) {
  return copy_level(h, h ? h->d_pyr : nullptr, frame, level, padded, dst, dst_stride, width, height);
}

int orb_get_blurred_level(OrbExtractor* h, int frame, int level, uint8_t* dst, size_t dst_stride, int* width,
                          int* height) {
  if (h && h->have_plan && h->blur_stale && h->last_frames > 0) {
    // the last call blurred patches only: blur the levels of its frames now
    DeviceGuard g(h->device);
    ORB_HIP_CHECK(wait_own(h));
    launch_blur(h->d_pyr, h->d_blur, h->plan, h->d_blur_items, (int)h->blur_items.size(), h->taps, h->last_frames,
                h->stream);
    ORB_HIP_CHECK(hipGetLastError());
    ORB_HIP_CHECK(hipStreamSynchronize(h->stream));
    h->blur_stale = false;
  }
  return copy_level(h, h ? h->d_blur : nullptr, frame, level, 0, dst, dst_stride, width, height);
}

static int fetch_points(OrbExtractor* h, const uint32_t* d_arr, int per_frame, int off, const int* d_cnt, int frame,
                        int level, std::vector<uint32_t>& out) {
  int rc = check_frame_level(h, frame, level);
  if (rc != ORBHIP_OK) return rc;
  DeviceGuard g(h->device);
  ORB_HIP_CHECK(wait_own(h));
  int n = 0;
  ORB_HIP_CHECK(hipMemcpyAsync(&n, d_cnt + frame * h->nlevels + level, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  ORB_HIP_CHECK(hipStreamSynchronize(h->stream));
  out.resize(n);
  if (n) {
    ORB_HIP_CHECK(hipMemcpyAsync(out.data(), d_arr + (size_t)frame * per_frame + off, sizeof(uint32_t) * n, hipMemcpyDeviceToHost, h->stream));
    ORB_HIP_CHECK(hipStreamSynchronize(h->stream));
  }
  return ORBHIP_OK;
}

int orb_get_fast_candidates(OrbExtractor* h, int frame, int level, int max_n, float* xyr) {
  std::vector<uint32_t> v;
  int rc = check_frame_level(h, frame, level);
  if (rc != ORBHIP_OK) return rc;
  const LevelGeom& L = h->plan.lv[level];
  rc = fetch_points(h, h->d_cand, h->plan.cand_total, L.cand_off, h->d_cand_cnt, frame, level, v);
  if (rc != ORBHIP_OK) return rc;
  if ((int)v.size() > L.cand_cap) v.resize(L.cand_cap);
  // workgroups append in arbitrary order; restore the reference's order:
  // cell row, cell column, then raster order inside the cell
  auto key = [&](uint32_t p) {
    const int x = p & kCoordMask, y = (p >> kCoordBits) & kCoordMask;
    return ((uint64_t)((y - 3) / L.hCell) << 40) | ((uint64_t)((x - 3) / L.wCell) << 28) | ((uint64_t)y << 14) | (uint64_t)x;
  };
  std::sort(v.begin(), v.end(), [&](uint32_t a, uint32_t b) { return key(a) < key(b); });
  const int n = (int)v.size();
  for (int i = 0; i < std::min(n, max_n); i++) {
    xyr[3 * i] = (float)(v[i] & kCoordMask);
    xyr[3 * i + 1] = (float)((v[i] >> kCoordBits) & kCoordMask);
    xyr[3 * i + 2] = (float)(v[i] >> 24);
  }
  return n;
}

int orb_get_level_keypoints(OrbExtractor* h, int frame, int level, int max_n, float* xyr) {
  std::vector<uint32_t> v;
  int rc = check_frame_level(h, frame, level);
  if (rc != ORBHIP_OK) return rc;
  const LevelGeom& L = h->plan.lv[level];
  rc = fetch_points(h, h->d_sel, h->plan.sel_total, L.sel_off, h->d_sel_cnt, frame, level, v);
  if (rc != ORBHIP_OK) return rc;
  const int n = (int)v.size();
  for (int i = 0; i < std::min(n, max_n); i++) {
    xyr[3 * i] = (float)((v[i] & kCoordMask) + kBorder);
    xyr[3 * i + 1] = (float)(((v[i] >> kCoordBits) & kCoordMask) + kBorder);
    xyr[3 * i + 2] = (float)(v[i] >> 24);
  }
  return n;
}

}  // extern "C"
