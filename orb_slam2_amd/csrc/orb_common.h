// Shared host/device declarations of liborbhip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

#include "../../include/orbhip.h"

namespace orbhip {

constexpr int kMaxLevels = 12;
constexpr int kEdge = 19;    // EDGE_THRESHOLD: width of the reflected frame around every level
constexpr int kXOff = 32;    // slab column that holds level x = 0 (>= kEdge, keeps rows 32-B aligned)
constexpr int kBorder = 16;  // origin of the FAST domain (EDGE_THRESHOLD - 3)
constexpr int kCoordBits = 12;
constexpr int kCoordMask = (1 << kCoordBits) - 1;

// One pyramid level of one frame.  A frame's levels live in one "slab"; level pixel
// (x, y) of frame f is at  pyr + f*slab + off + (y + kEdge)*pitch + x + kXOff.
struct LevelGeom {
  int w, h;          // level size
  int pitch;         // bytes per slab row (multiple of 64)
  int off;           // byte offset of the level's first (padded) row in the slab
  int nCols, nRows;  // FAST cell grid
  int wCell, hCell;
  int wCellM;        // 65536 / wCell + 1: (n * wCellM) >> 16 == n / wCell for 0 <= n < 256
  int nValidCols, nValidRows;  // cells the reference does not skip
  int maxBX, maxBY;  // w - 16, h - 16
  int quota;         // keypoints wanted on this level
  int nIni;          // quad-tree root count
  float hX;          // root width
  int cand_off, cand_cap;  // slice of the per-frame candidate array
  int sel_off, sel_cap;    // slice of the per-frame selected-keypoint array
  int xtab_off, ytab_off;  // resize coefficient tables (entries)
  float scale;       // mvScaleFactor[level]
  float kp_size;     // (float)(int)(31 * scale)
};

struct Plan {
  int nlevels;
  int rows, cols;
  int cand_total;  // candidate entries per frame
  int sel_total;   // selected entries per frame
  int pad_;
  unsigned long long slab;  // bytes per frame
  LevelGeom lv[kMaxLevels];
};

struct XTab {  // horizontal bilinear taps of one destination column
  uint16_t sx;
  int16_t a0, a1;
  uint16_t pad;
};
struct YTab {  // vertical taps of one destination row
  uint16_t sy0, sy1;
  int16_t b0, b1;
};
struct FastItem {  // a run of cells in one cell row, handled by one workgroup
  int16_t level, row, j0, ncells;
};
// What the FAST kernel needs to know about an item, worked out on the host: a workgroup fetches
// this one record and nothing of the plan's level table (whose entry it could only ask for after
// the item had arrived: one memory round trip less at the start of every workgroup).
struct FastWork {
  int32_t base_off;            // byte offset, in a frame's slab, of the staged view's first dword
  int32_t pitch;               // slab row pitch of the level
  int32_t cand_off, cand_cap;  // the level's slice of the frame's candidate array
  int32_t wCellM;              // 65536 / wCell + 1
  int16_t level, ncells, wCell, hCell;
  int16_t tw, th;              // size of the view (cells' pixels + 6), clipped to the level
  int16_t xb, iniY;            // its origin in level coordinates
  int16_t xoff, ndw;           // LDS column of the first interior pixel; dwords staged per row
  int16_t pad[4];
};
static_assert(sizeof(FastWork) == 48, "three 16-byte words");
struct BlurItem {
  int16_t level, tx, ty, pad;
};

void set_error(const std::string& msg);
int hip_fail(hipError_t e, const char* what);

#define ORB_HIP_CHECK(expr)                                   \
  do {                                                        \
    hipError_t e__ = (expr);                                  \
    if (e__ != hipSuccess) return ::orbhip::hip_fail(e__, #expr); \
  } while (0)

}  // namespace orbhip
