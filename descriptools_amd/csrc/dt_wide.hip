// dt_wide.hip -- the descriptors that read HEIGHTS, on a DEM (or HAND) that float32 cannot hold.
//
// The reference takes every height difference in the raster's OWN dtype: slope.py:244-258 under Numba typing
// (float64 - float64 for a float64 DEM, int64 - int64 for an integer one), flowhand.py:436-438 `dem - dem[indices]`,
// downslope.py:468 `dem[i] - dem[pos]`; and gfi.py:289-294 / :429-440 add 0.01 to whatever HAND they are given, in
// float64.  The resident chain and every hot kernel of this library keep heights as float32 -- the same arithmetic
// exactly when the heights ARE float32 values, which is what descriptools_amd/_lib.py checks at the boundary.  A
// raster that fails that check (a genuinely float64 DEM; integer heights beyond 2^24, which float64 holds exactly up
// to 2^53) takes the kernels below instead: one thread per cell on global memory, heights as float64, the literal
// expressions.  They are the capability, not the benchmark: a float64 raster moves twice the bytes and these kernels
// make no attempt at the LDS staging of the float32 path (slope: 9 L2-served reads per cell; downslope: the plain
// per-cell walk).  Pinned by tests/golden/f64.npz, the reference's own run on such a raster.
#include <math.h>

#include "dt_common.h"
#include "dt_kernels.h"

// S3 slope (%), slope.py:210-259 with float64 heights
__global__ __launch_bounds__(256) void k_slope_f64(const double *__restrict__ dem, int H, int W, double px,
                                                  float *__restrict__ slope) {
  const int x = (int)(blockIdx.x * 64u + (threadIdx.x & 63u));
  const int y = (int)(blockIdx.y * 4u + (threadIdx.x >> 6));
  if (x >= W || y >= H) return;
  const long long i = (long long)y * W + x;
  const double c = dem[i];
  if (c <= -100.0) {  // slope.py:231
    slope[i] = DT_NODATA;
    return;
  }
  const double dcard = px, ddiag = px * sqrt(2.0);
  double aux = 0.0;
  // scan order NW, N, NE, W, E, SW, S, SE with strict `<` (slope.py:244-258); only the maximum reaches the output
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const int dy = k < 3 ? -1 : (k < 5 ? 0 : 1);
    const int dx = (k == 0 || k == 3 || k == 5) ? -1 : ((k == 1 || k == 6) ? 0 : 1);
    const int yy = y + dy, xx = x + dx;
    if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;  // the -100 ring (slope.py:175-182)
    const double nb = dem[(long long)yy * W + xx];
    if (nb == -100.0) continue;  // slope.py:247
    const double v = (c - nb) / ((dy == 0 || dx == 0) ? dcard : ddiag);
    if (aux < v) aux = v;
  }
  slope[i] = (float)(aux * 100.0);
}

// F4 HAND, flowhand.py:414-442, in float64
__global__ __launch_bounds__(256) void k_hand_f64(const double *__restrict__ dem, const int64_t *__restrict__ idx,
                                                 int64_t n, double *__restrict__ hand) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double z = dem[i];
  double h = -100.0;
  int64_t k = idx[i];
  if (z != -100.0 && k != -100) {
    if (k < 0) k += n;  // numpy negative indexing, as the reference's dem[indices]
    if (k >= 0 && k < n) {
      h = z - dem[k];
      if (h < 0.0 && h != -100.0) h = 0.0;
    }
  }
  hand[i] = h;
}

// D2 + D3 downslope, downslope.py:435-532 + :161-314, heights in float64 (the walk of k_downslope in dt_kernels.hip)
__device__ __forceinline__ int64_t dw_step(int64_t pos, uint32_t code, int H, int W, bool &diag) {
  if (!dt_d8_valid(code)) return -1;
  int dy, dx;
  dt_d8_delta(code, dy, dx);
  const int y = (int)(pos / W), x = (int)(pos - (int64_t)y * W);
  const int ny = y + dy, nx = x + dx;
  if (ny < 0 || ny >= H || nx < 0 || nx >= W) return -2;
  diag = dy != 0 && dx != 0;
  return (int64_t)ny * W + nx;
}
__global__ __launch_bounds__(256) void k_downslope_f64(const double *__restrict__ dem, const uint8_t *__restrict__ fdr,
                                                      int H, int W, double px, double dz, int raw,
                                                      float *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)H * W) return;
  const double z0 = dem[i];
  if (z0 <= -100.0) {  // downslope.py:460
    out[i] = DT_NODATA;
    return;
  }
  const double dcard = px, ddiag = px * sqrt(2.0);
  int64_t pos = i;
  double dist = 0.0, drop = 0.0;
  int loop = 0;
  bool failed = false;
  while (drop < dz) {
    bool diag = false;
    const int64_t t = dw_step(pos, fdr[pos], H, W, diag);
    if (t == -2) { failed = true; break; }  // raster-edge exit: stop (downslope.py:209-228)
    if (t >= 0) {
      const double zt = dem[t];
      if (zt == -100.0) { failed = true; break; }  // nodata ahead: stop without moving (:231-281)
      pos = t;
      dist += diag ? ddiag : dcard;
      drop = z0 - zt;  // downslope.py:468, the DEM's own dtype
    }
    if (++loop == 5000) { failed = true; break; }  // :303-304 / :518-521
  }
  if (raw && failed) out[i] = -50.0f;
  else out[i] = dist == 0.0 ? 0.0f : (float)(drop / dist);
}

// G2 / G3 with a float64 HAND: ln(b * (A * size^2)^n / (hand + 0.01)), gfi.py:268-294 (A = fac[idx], no zero guard)
// and :404-440 (A = the cell's own fac, 0 -> 1)
__global__ __launch_bounds__(256) void k_gfi_f64h(const double *__restrict__ hand, const int64_t *__restrict__ fac,
                                                 const int64_t *__restrict__ idx, int64_t n, double expo, double b,
                                                 double size, int own_area, float *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double h = hand[i];
  if (h <= -100.0) {
    out[i] = DT_NODATA;
    return;
  }
  double a;
  if (own_area) {  // 1: the cell's own area with the zero guard of gfi.py:432; 2: an explicit area raster, no guard
    const int64_t f = fac[i];
    a = (f == 0 && own_area == 1) ? 1.0 * (size * size) : (double)f * (size * size);
  } else {
    int64_t k = idx[i];
    if (k == -100) k = 0;  // gfi.py:141-143: fac.flat[0]
    if (k < 0) k += n;     // numpy negative indexing
    a = (double)((k >= 0 && k < n) ? fac[k] : fac[0]) * (size * size);
  }
  out[i] = (float)log(b * pow(a, expo) / (h + 0.01));
}

int dt_launch_slope_f64(hipStream_t s, const double *dem, int64_t H, int64_t W, double px, float *slope) {
  if (H * W == 0) return DT_OK;
  hipLaunchKernelGGL(k_slope_f64, dim3((unsigned)((W + 63) / 64), (unsigned)((H + 3) / 4)), dim3(256), 0, s, dem, (int)H,
                     (int)W, px, slope);
  return DT_OK;
}
int dt_launch_hand_f64(hipStream_t s, const double *dem, const int64_t *idx, int64_t n, double *hand) {
  if (n) hipLaunchKernelGGL(k_hand_f64, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dem, idx, n, hand);
  return DT_OK;
}
int dt_launch_downslope_f64(hipStream_t s, const double *dem, const uint8_t *fdr, int64_t H, int64_t W, double px,
                            double dz, int raw, float *out) {
  const int64_t n = H * W;
  if (n) hipLaunchKernelGGL(k_downslope_f64, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dem, fdr, (int)H, (int)W,
                            px, dz, raw, out);
  return DT_OK;
}
int dt_launch_gfi_f64h(hipStream_t s, const double *hand, const int64_t *fac, const int64_t *idx, int64_t n,
                       double expo, double b, double size, int own_area, float *out) {
  if (n) hipLaunchKernelGGL(k_gfi_f64h, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, hand, fac, idx, n, expo, b,
                            size, own_area, out);
  return DT_OK;
}
