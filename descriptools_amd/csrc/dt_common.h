// dt_common.h -- shared host/device helpers of libdescriptools_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/descriptools_hip.h"

#define DT_NODATA (-100.0f)

// ---- error plumbing ---------------------------------------------------------------------
void dt_set_error(const char *fmt, ...);

#define DT_HIP(call)                                                                   \
  do {                                                                                 \
    hipError_t e__ = (call);                                                           \
    if (e__ != hipSuccess) {                                                           \
      dt_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__,   \
                   __LINE__);                                                          \
      return e__ == hipErrorOutOfMemory ? DT_ENOMEM : DT_EHIP;                         \
    }                                                                                  \
  } while (0)

#define DT_REQUIRE(cond, msg)                      \
  do {                                             \
    if (!(cond)) {                                 \
      dt_set_error("invalid argument: %s", msg);   \
      return DT_EINVAL;                            \
    }                                              \
  } while (0)

#define DT_TRY(expr)          \
  do {                        \
    int rc__ = (expr);        \
    if (rc__ != DT_OK) return rc__; \
  } while (0)

// ---- test / experiment knobs (dt_debug_set in the C ABI; all 0 by default) -------------------------
enum { DT_DBG_TWI_FLAG_ALL = 0, DT_DBG_TWI_PLAIN = 1, DT_DBG_TWI_WX = 2, DT_DBG_TWI_MAP = 3, DT_DBG_DS_MARGIN = 4, DT_DBG_NO_FUSED_FA_FH = 5, DT_DBG_HY_FILL_SWEEPS = 6, DT_DBG_HY_FLAT_SWEEPS = 7, DT_DBG_HY_COLOUR_MIN = 8 /* tiles from which the conditioning rounds are coloured (0: the default; tests) */, DT_DBG_COUNT = 9 };
#define DT_TWI_WX_DEFAULT 1 /* tile geometry of the fused slope + TI + MTI stencil: see k_slope_twi */
int dt_debug_get(int key);

// ---- context ------------------------------------------------------------------------------
struct dt_ctx {
  int device;
  hipStream_t stream;
  bool own_stream;
  char *scratch;
  size_t scratch_bytes;
  size_t scratch_used;  // bump pointer, reset at the start of every entry point
  int scratch_owner;    // which two-phase op's state lives in `scratch` (0 none, 1 flow accumulation, 2 HAND):
  int64_t owner_h, owner_w;  // set by *_local_w, cleared by every dt_scratch_reset, required by *_finish_w
  char *owner_ptr;      // where in `scratch` that state starts (HAND's follows flow accumulation's when phase 2 of the
  char *owner_ptr2;     // one and phase 1 of the other are fused: owner_ptr2 = HAND's region reserved beside it)
  char *scratch2;       // rank-level solves (must not disturb the two-phase tile scratch)
  size_t scratch2_bytes;
  hipEvent_t ev;        // fork / join with another context's stream (created on first use)
  int *status;          // device word of sticky DT_STATUS_* bits raised by kernels (dt_ctx_status reads and clears)
  char *aux;            // workspace of the fused slope + TI + MTI stencil (it runs between the two phases of
  size_t aux_bytes;     // the tile kernels in a multi-GPU step, so it must not touch `scratch`)
  uint64_t ws_gen;      // bumped whenever scratch / scratch2 / aux is reallocated or freed: a captured graph holds
                        // their raw addresses and must not be replayed across such a change (dt_graph_launch checks)
};

// grow-only scratch, bump-allocated per entry point (256-B aligned)
int dt_scratch_reset(dt_ctx *ctx, size_t total_bytes);
void *dt_scratch_take(dt_ctx *ctx, size_t bytes);
static inline size_t dt_align256(size_t b) { return (b + 255) & ~(size_t)255; }

// ---- D8 decoding (flowhand.py:801-824 / downslope.py:490-513) -------------------------------
// code -> bit index: 1=E 2=SE 4=S 8=SW 16=W 32=NW 64=N 128=NE
__device__ __forceinline__ bool dt_d8_valid(uint32_t code) {
  return code != 0u && (code & (code - 1u)) == 0u;
}
// dx/dy for bit index i, packed 2 bits each as (d + 1)
//   i :  0  1  2  3  4  5  6  7
//   dx:  1  1  0 -1 -1 -1  0  1
//   dy:  0  1  1  1  0 -1 -1 -1
#define DT_PK8(a, b, c, d, e, f, g, h)                                                 \
  ((uint32_t)((a) + 1) | (uint32_t)((b) + 1) << 2 | (uint32_t)((c) + 1) << 4 |          \
   (uint32_t)((d) + 1) << 6 | (uint32_t)((e) + 1) << 8 | (uint32_t)((f) + 1) << 10 |    \
   (uint32_t)((g) + 1) << 12 | (uint32_t)((h) + 1) << 14)
#define DT_DX_PACK DT_PK8(1, 1, 0, -1, -1, -1, 0, 1)
#define DT_DY_PACK DT_PK8(0, 1, 1, 1, 0, -1, -1, -1)
__device__ __forceinline__ void dt_d8_delta(uint32_t code, int &dy, int &dx) {
  int i = __ffs((int)code) - 1;
  dx = (int)((DT_DX_PACK >> (2 * i)) & 3u) - 1;
  dy = (int)((DT_DY_PACK >> (2 * i)) & 3u) - 1;
}

// ---- raster window -------------------------------------------------------------------------------
// A core window of H x W cells inside rasters of row stride ld, placed at (gy0, gx0) of a global
// Hg x Wg raster.  Raster pointers are relative to the core origin: cell (y, x) is p[y * ld + x].
// `halo` cells beyond the core are present in memory on every side (0 for a single-GPU raster, where
// the core IS the global raster: ld = W, gy0 = gx0 = 0, Hg = H, Wg = W).
struct DtWin {
  int H, W;
  long long ld;
  int gy0, gx0, Hg, Wg;
  int halo;
};
__host__ __device__ __forceinline__ bool dt_in_core(const DtWin &w, int y, int x) {
  return y >= 0 && y < w.H && x >= 0 && x < w.W;
}
// inside the global raster AND present in memory
__host__ __device__ __forceinline__ bool dt_readable(const DtWin &w, int y, int x) {
  int gy = w.gy0 + y, gx = w.gx0 + x;
  return gy >= 0 && gy < w.Hg && gx >= 0 && gx < w.Wg && y >= -w.halo && y < w.H + w.halo && x >= -w.halo &&
         x < w.W + w.halo;
}
// readable AND its D8 code is in memory: a rank computes the codes of its core and of its halo minus the outermost
// ring (a code needs the cell's eight neighbours), so a cell on the ring of the rank's memory has a height but no code
// -- unless that ring is the edge of the global raster, where the border rule gave it one
__host__ __device__ __forceinline__ bool dt_has_code(const DtWin &w, int y, int x) {
  if (!dt_readable(w, y, x)) return false;
  const int gy = w.gy0 + y, gx = w.gx0 + x;
  return (y > -w.halo || gy == 0) && (y < w.H + w.halo - 1 || gy == w.Hg - 1) && (x > -w.halo || gx == 0) &&
         (x < w.W + w.halo - 1 || gx == w.Wg - 1);
}
__host__ __device__ __forceinline__ bool dt_in_global(const DtWin &w, int y, int x) {
  int gy = w.gy0 + y, gx = w.gx0 + x;
  return gy >= 0 && gy < w.Hg && gx >= 0 && gx < w.Wg;
}
// ring of the core window ("rank perimeter"), P cells: top row, bottom row, left column, right column
__host__ __device__ __forceinline__ long long dt_perim_count(int H, int W) {
  if (H <= 0 || W <= 0) return 0;
  if (H == 1) return W;
  if (W == 1) return H;
  return 2ll * W + 2ll * (H - 2);
}
__host__ __device__ __forceinline__ long long dt_perim_index(int H, int W, int y, int x) {
  if (H == 1) return x;
  if (W == 1) return y;
  if (y == 0) return x;
  if (y == H - 1) return (long long)W + x;
  if (x == 0) return 2ll * W + (y - 1);
  if (x == W - 1) return 2ll * W + (H - 2) + (y - 1);
  return -1;
}
__host__ __device__ __forceinline__ void dt_perim_cell(int H, int W, long long i, int &y, int &x) {
  if (H == 1) { y = 0; x = (int)i; return; }
  if (W == 1) { y = (int)i; x = 0; return; }
  if (i < W) { y = 0; x = (int)i; }
  else if (i < 2ll * W) { y = H - 1; x = (int)(i - W); }
  else if (i < 2ll * W + (H - 2)) { y = (int)(i - 2ll * W) + 1; x = 0; }
  else { y = (int)(i - 2ll * W - (H - 2)) + 1; x = W - 1; }
}
