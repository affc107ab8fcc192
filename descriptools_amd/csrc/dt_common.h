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

// ---- context ------------------------------------------------------------------------------
struct dt_ctx {
  int device;
  hipStream_t stream;
  bool own_stream;
  char *scratch;
  size_t scratch_bytes;
  size_t scratch_used;  // bump pointer, reset at the start of every entry point
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
