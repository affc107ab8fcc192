// dt_stencil.hip -- 3x3 stencil kernels: slope (S3), D8 (N1), radians, fused slope + TI + MTI (T2, T3).
// Reference citations are file:line relative to /root/reference/descriptools/.
#include <math.h>

#include "dt_common.h"
#include "dt_kernels.h"
#include "dt_math.h"

// ===========================================================================================
// 3x3 stencil: slope (S3, slope.py:210-259) + D8 (N1) + radians + optional fused TI/MTI.
//
// Tile = 256 columns x 16 rows per 256-thread workgroup, staged (with a 1-cell halo) through
// LDS by coalesced 16-byte row loads; thread (tx, ty) then owns a 4-wide x 4-tall patch and
// reads its 6 x 6 neighbourhood as one ds_read_b128 + two ds_read_b32 per row.  Stores are
// float4 / uchar4 per row (1 KiB / 256 B contiguous per wave).  Workgroup ids are remapped so
// that each XCD (ids congruent mod 8 share one) sweeps its own horizontal band of the raster
// top to bottom: the halo rows shared by vertically adjacent tiles are then re-read from that
// XCD's L2 instead of HBM.
//
// Exactness: the reference compares float64 quotients (z_c - z_nb)/d in scan order
// NW,N,NE,W,E,SW,S,SE with strict '<'.  Division by a positive constant is monotone and
// injective on float32 differences, so the maximum over the 4 cardinal (4 diagonal)
// neighbours is taken on the float32 differences and only the two class maxima are divided
// in float64 -- bit-identical results with 2 instead of 8 float64 divisions per cell.
// ===========================================================================================
#define SD_TX 256
#define SD_TY 16
#define SD_LDW (SD_TX + 8)  // LDS row stride in floats; interior starts at column 4

struct SlopeCell {
  float slope;
  uint8_t code;
};

// scan positions: NW0 N1 NE2 W3 E4 SW5 S6 SE7.  NEED_CODE = false (slope only): the D8 bookkeeping
// (which neighbour, scan position for ties) is skipped -- the slope value does not depend on it.
// scan position (NW 0, N 1, NE 2, W 3, E 4, SW 5, S 6, SE 7; 8 = none) of a D8 code
__device__ __forceinline__ int dt_scan_pos(uint32_t code) {
  if (code == 0u) return 8;
  // bit index of the code 0..7 = E SE S SW W NW N NE -> position 4 7 6 5 3 0 1 2
  return (int)((0x21035674u >> (4 * (__ffs((int)code) - 1))) & 0xFu);
}

// Neighbour heights arrive with nodata (== -100) replaced by +inf (done once per cell when the tile is
// staged): c - inf = -inf never beats a candidate, which is the reference's "neighbour == -100 skipped"
// (slope.py:247) without a test per neighbour.  `c` is the centre's original value.
template <bool NEED_CODE, bool NEED_SLOPE>
__device__ __forceinline__ SlopeCell dt_slope_cell(float c, float nw, float n, float ne, float w,
                                                  float e, float sw, float s, float se,
                                                  double inv_card, double inv_diag, double dcard,
                                                  double ddiag) {
  SlopeCell r;
  if (c <= DT_NODATA) {  // slope.py:231
    r.slope = DT_NODATA;
    r.code = 0;
    return r;
  }
  // cardinals in scan order N, W, E, S, then the diagonals NW, NE, SW, SE; strict > keeps the first maximum
  float cb = 0.0f, db = 0.0f;
  uint32_t ccode = 0, dcode = 0;
  if (NEED_CODE) {
#define DT_CAND(nb, best, bcode, code_) \
  {                                     \
    float d_ = c - (nb);                \
    if (d_ > best) {                    \
      best = d_;                        \
      bcode = code_;                    \
    }                                   \
  }
    DT_CAND(n, cb, ccode, 64u)
    DT_CAND(w, cb, ccode, 16u)
    DT_CAND(e, cb, ccode, 1u)
    DT_CAND(s, cb, ccode, 4u)
    DT_CAND(nw, db, dcode, 32u)
    DT_CAND(ne, db, dcode, 128u)
    DT_CAND(sw, db, dcode, 8u)
    DT_CAND(se, db, dcode, 2u)
#undef DT_CAND
  } else {
    cb = fmaxf(fmaxf(fmaxf(c - n, c - w), fmaxf(c - e, c - s)), 0.0f);
    db = fmaxf(fmaxf(fmaxf(c - nw, c - ne), fmaxf(c - sw, c - se)), 0.0f);
  }
  // Exact float64 divisions are ~15 instructions each.  Fast path: multiply by the (correctly rounded)
  // reciprocals -- within 3 float64 ulp of the reference's quotient -- and accept the result only if
  // neither the cardinal / diagonal comparison nor (when the slope is wanted) the final float32 rounding
  // can be affected by those ulps; otherwise divide.  Results are bit-identical either way.
  double vc = (double)cb * inv_card, vd = (double)db * inv_diag;
  const double EPS = 8.9e-16;  // 4 ulp, relative
  double vmax = vc > vd ? vc : vd;
  bool ambiguous = (vc != vd) && fabs(vc - vd) <= EPS * vmax;
  if (NEED_SLOPE) {
    float f_lo = (float)(vmax * (100.0 * (1.0 - EPS))), f_hi = (float)(vmax * (100.0 * (1.0 + EPS)));
    ambiguous = ambiguous || (f_lo != f_hi);
  }
  if (ambiguous) {
    vc = cb > 0.0f ? (double)cb / dcard : 0.0;
    vd = db > 0.0f ? (double)db / ddiag : 0.0;
  }
  double v;
  uint32_t code;
  // equal quotients: the candidate met first in scan order wins (positions looked up only then)
  if (vc > vd || (vc == vd && dt_scan_pos(ccode) < dt_scan_pos(dcode))) {
    v = vc;
    code = ccode;
  } else {
    v = vd;
    code = dcode;
  }
  r.slope = (float)(v * 100.0);  // slope.py:259
  r.code = (uint8_t)code;
  return r;
}

// XCD-aware tile mapping: workgroup id b runs on XCD group (b % 8); give each group a band of tile rows
// and walk it row-major -- each group starting an eighth of a band further in than the previous one (and
// wrapping), so that the eight write fronts are not exactly one band apart in memory: 1-1.5 % on the fused
// stencil on every placement tried (tools/placement_probe5.py; identity and row-interleaved maps: slower).
__device__ __forceinline__ int sd_tile_of_block(int b, int ntiles) {
  int xcd = b & 7, j = b >> 3;
  int q = ntiles >> 3, rem = ntiles & 7;
  int base = xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q;
  int n = q + (xcd < rem ? 1 : 0);  // tiles of this band (j < n)
  int k = j + xcd * (n >> 3);
  return base + (k >= n ? k - n : k);
}
// Experimental workgroup -> tile maps of the fused stencil (DT_DBG_TWI_MAP = mode | param << 8), for the placement
// study of DESIGN.md 6: the band map above makes the eight XCDs sweep eight 4-MiB windows per raster that stay a
// fixed distance apart; the others change which parts of the rasters are written at the same time.
//   1  bands without the stagger            2  tile rows dealt round-robin to the XCDs (one moving front per raster)
//   3  groups of `param` tile rows per XCD, round-robin        4  bands, stagger of xcd * param tiles
//   5  bands walked in chunks of 256 tiles in a scrambled order (chunk * 5 + 3 xcd mod chunks)
__device__ __forceinline__ int sd_tile_of_block_x(int b, int tiles_x, int tiles_y, int mode_param) {
  const int ntiles = tiles_x * tiles_y, mode = mode_param & 0xFF, param = mode_param >> 8;
  const int xcd = b & 7, j = b >> 3, q = ntiles >> 3;
  if ((ntiles & 7) != 0 || mode == 0) return sd_tile_of_block(b, ntiles);
  if (mode == 1) return xcd * q + j;
  if (mode == 2 || mode == 3) {
    const int G = mode == 2 ? 1 : (param > 0 ? param : 2);
    if (tiles_y % (8 * G) != 0) return sd_tile_of_block(b, ntiles);
    const int jr = j / tiles_x, jc = j - jr * tiles_x;
    const int row = ((jr / G) * 8 + xcd) * G + jr % G;
    return row * tiles_x + jc;
  }
  if (mode == 4) {
    int k = j + xcd * param;
    k %= q;
    return xcd * q + k;
  }
  if (mode == 5) {
    const int nch = q >> 8;
    if (nch < 2 || (q & 255) != 0) return sd_tile_of_block(b, ntiles);
    const int ch = j >> 8, in = j & 255;
    const int ch2 = (int)(((unsigned)ch * 5u + 3u * (unsigned)xcd) % (unsigned)nch);
    return xcd * q + ch2 * 256 + in;
  }
  return sd_tile_of_block(b, ntiles);
}

__device__ __forceinline__ void sd_tile_origin(int b, int tiles_x, int tiles_y, int &x0, int &y0) {
  int tile = sd_tile_of_block(b, tiles_x * tiles_y);
  int tyi = tile / tiles_x, txi = tile - tyi * tiles_x;
  x0 = txi * SD_TX;
  y0 = tyi * SD_TY;
}

// stage (SD_TY + 2) x (SD_TX + 2) cells; outside the GLOBAL raster = -100 ring (slope.py:175); cells outside
// the core but inside the global raster come from the halo of the window.  nodata (and everything outside
// the raster) is staged as +inf: see dt_slope_cell.  The caller synchronises.
template <int TX = SD_TX, int TY = SD_TY>
__device__ __forceinline__ void sd_stage(float *t, const float *__restrict__ dem, const DtWin &w, int x0, int y0,
                                         int vec_ok) {
  const int H = w.H, W = w.W;
  const int ylo = -(w.gy0 > 0 ? 1 : 0), yhi = H + (w.gy0 + H < w.Hg ? 1 : 0);  // readable rows [ylo, yhi)
  const int xlo = -(w.gx0 > 0 ? 1 : 0), xhi = W + (w.gx0 + W < w.Wg ? 1 : 0);
  const float pinf = __builtin_inff();
  // Block-uniform fast form for tiles whose whole 18 x 258 window is readable: every load of a thread is issued
  // before the first use (five 16-byte loads and one halo value in flight per thread).  The guarded loop below
  // waits for each load before the next: five dependent memory round trips per workgroup, which made every
  // stencil kernel latency-bound (the D8-only kernel took as long as the 8 B/cell slope kernel).
  if (vec_ok && y0 - 1 >= ylo && y0 + TY + 1 <= yhi && x0 - 1 >= xlo && x0 + TX + 1 <= xhi) {
    constexpr int NV = ((TY + 2) * (TX / 4) + 255) / 256;  // 5
    float4 v[NV];
    const float *base = dem + (long long)(y0 - 1) * w.ld + x0;
#pragma unroll
    for (int u = 0; u < NV; u++) {
      const int i = threadIdx.x + 256 * u;
      if (i < (TY + 2) * (TX / 4)) {
        const int r = i / (TX / 4), c4 = i - r * (TX / 4);
        v[u] = *reinterpret_cast<const float4 *>(base + (long long)r * w.ld + c4 * 4);
      }
    }
    float hv = 0.0f;
    const int hr = threadIdx.x >> 1, hside = threadIdx.x & 1;
    if (threadIdx.x < (TY + 2) * 2) hv = base[(long long)hr * w.ld + (hside ? TX : -1)];
#pragma unroll
    for (int u = 0; u < NV; u++) {
      const int i = threadIdx.x + 256 * u;
      if (i < (TY + 2) * (TX / 4)) {
        const int r = i / (TX / 4), c4 = i - r * (TX / 4);
        float4 q = v[u];
        q.x = q.x == DT_NODATA ? pinf : q.x;
        q.y = q.y == DT_NODATA ? pinf : q.y;
        q.z = q.z == DT_NODATA ? pinf : q.z;
        q.w = q.w == DT_NODATA ? pinf : q.w;
        *reinterpret_cast<float4 *>(&t[r * (TX + 8) + 4 + c4 * 4]) = q;
      }
    }
    if (threadIdx.x < (TY + 2) * 2) t[hr * (TX + 8) + (hside ? 4 + TX : 3)] = hv == DT_NODATA ? pinf : hv;
    return;
  }
  for (int i = threadIdx.x; i < (TY + 2) * (TX / 4); i += 256) {
    int r = i / (TX / 4), c4 = i - r * (TX / 4);
    int gy = y0 - 1 + r, gx = x0 + c4 * 4;
    float4 v = make_float4(DT_NODATA, DT_NODATA, DT_NODATA, DT_NODATA);
    if (gy >= ylo && gy < yhi) {
      const float *p = dem + (long long)gy * w.ld + gx;
      if (vec_ok && gx + 3 < xhi) {
        v = *reinterpret_cast<const float4 *>(p);
      } else {
        if (gx < xhi) v.x = p[0];
        if (gx + 1 < xhi) v.y = p[1];
        if (gx + 2 < xhi) v.z = p[2];
        if (gx + 3 < xhi) v.w = p[3];
      }
    }
    v.x = v.x == DT_NODATA ? pinf : v.x;
    v.y = v.y == DT_NODATA ? pinf : v.y;
    v.z = v.z == DT_NODATA ? pinf : v.z;
    v.w = v.w == DT_NODATA ? pinf : v.w;
    *reinterpret_cast<float4 *>(&t[r * (TX + 8) + 4 + c4 * 4]) = v;
  }
  for (int i = threadIdx.x; i < (TY + 2) * 2; i += 256) {
    int r = i >> 1, side = i & 1;
    int gy = y0 - 1 + r, gx = side ? x0 + TX : x0 - 1;
    float v = DT_NODATA;
    if (gy >= ylo && gy < yhi && gx >= xlo && gx < xhi) v = dem[(long long)gy * w.ld + gx];
    t[r * (TX + 8) + (side ? 4 + TX : 3)] = v == DT_NODATA ? pinf : v;
  }
}

template <bool W_SLOPE, bool W_FDR, bool W_RAD>
__global__ __launch_bounds__(256, 6) void k_stencil(const float *__restrict__ dem, DtWin w,
                                                double px, float *__restrict__ slope,
                                                uint8_t *__restrict__ fdr,
                                                float *__restrict__ slope_rad, int tiles_x, int tiles_y,
                                                int vec_ok) {
  __shared__ __attribute__((aligned(16))) float t[(SD_TY + 2) * SD_LDW];

  int x0, y0;
  sd_tile_origin(blockIdx.x, tiles_x, tiles_y, x0, y0);
  const int H = w.H, W = w.W;
  const float pinf = __builtin_inff();
  sd_stage(t, dem, w, x0, y0, vec_ok);
  __syncthreads();

  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int cx = tx * 4;  // tile column of the patch
  const int ry = ty * 4;  // tile row of the patch
  const int gx = x0 + cx;
  if (gx >= W) return;
  const double dcard = px, ddiag = px * sqrt(2.0);
  const double inv_card = 1.0 / dcard, inv_diag = 1.0 / ddiag;

  // rolling 3-row window of 6 values (cols cx-1 .. cx+4)
  float a[6], bb[6], cc[6];
  auto load_row = [&](int lr, float *dst) {
    const float *p = &t[lr * SD_LDW + 4 + cx];
    float4 m = *reinterpret_cast<const float4 *>(p);
    dst[0] = p[-1];
    dst[1] = m.x;
    dst[2] = m.y;
    dst[3] = m.z;
    dst[4] = m.w;
    dst[5] = p[4];
  };
  load_row(ry, a);       // row above the first output row (tile row ry == raster row y0-1+ry)
  load_row(ry + 1, bb);  // first output row
#pragma unroll
  for (int j = 0; j < 4; j++) {
    int gy = y0 + ry + j;
    load_row(ry + 2 + j, cc);
    if (gy < H) {
      float so[4], ro[4];
      uint32_t codes = 0;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const float cz = bb[k + 1] == pinf ? DT_NODATA : bb[k + 1];  // the centre's own value
        SlopeCell sc = dt_slope_cell<W_FDR, (W_SLOPE || W_RAD)>(cz, a[k], a[k + 1], a[k + 2], bb[k],
                                                                       bb[k + 2], cc[k], cc[k + 1], cc[k + 2],
                                                                       inv_card, inv_diag, dcard, ddiag);
        so[k] = sc.slope;
        uint32_t code = sc.code;
        if (W_FDR) {
          // N1 border rule: a border cell with no lower neighbour drains out of the raster
          int gyy = w.gy0 + gy, gxx = w.gx0 + gx + k;  // global position
          if (code == 0u && cz > DT_NODATA) {
            if (gyy == w.Hg - 1) code = 4u;
            else if (gyy == 0) code = 64u;
            else if (gxx == 0) code = 16u;
            else if (gxx == w.Wg - 1) code = 1u;
          }
          codes |= code << (8 * k);
        }
        if (W_RAD) ro[k] = dt_slope_rad(sc.slope, cz);
      }
      long long o = (long long)gy * w.ld + gx;
      bool full = vec_ok && gx + 3 < W;
      if (full) {
        if (W_SLOPE) *reinterpret_cast<float4 *>(slope + o) = make_float4(so[0], so[1], so[2], so[3]);
        if (W_RAD) *reinterpret_cast<float4 *>(slope_rad + o) = make_float4(ro[0], ro[1], ro[2], ro[3]);
        if (W_FDR) *reinterpret_cast<uint32_t *>(fdr + o) = codes;
      } else {
#pragma unroll
        for (int k = 0; k < 4; k++) {
          if (gx + k < W) {
            if (W_SLOPE) slope[o + k] = so[k];
            if (W_RAD) slope_rad[o + k] = ro[k];
            if (W_FDR) fdr[o + k] = (uint8_t)(codes >> (8 * k));
          }
        }
      }
    }
#pragma unroll
    for (int q = 0; q < 6; q++) {
      a[q] = bb[q];
      bb[q] = cc[q];
    }
  }
}

// ===========================================================================================
// Fused slope + TI + MTI (+ radians), hot / cold split -- the north_star's "slope+TWI stencil".
//
// k_slope_twi is branch-free: every cell takes the product form of the slope and the float32 fast path
// of the logarithms (dt_math.h), and a cell whose result cannot be PROVEN identical to the literal
// float64 expression is only flagged (one bit per cell, 16 per lane):
//   slope   q = max(cb * kc, db * kd), kc = 100 / px, kd = 100 / (px sqrt 2), is within 2^-51 (relative) of
//           the reference's fl(fl(d / dist) * 100) whichever of the two classes wins, so float32(q) is the
//           reference's float32 unless q lies within SD_MID ulps of a float32 rounding boundary (bits 0-28
//           of the mantissa = 2^28), is not a normal float32, or the quotient / arctangent leave their
//           fast domain;
//   TI/MTI  fast-path domain and |result| >= DT_FAST_MIN, exactly the conditions of dt_twi_cell.
// A workgroup with a flagged cell marks its tile and writes the lanes' masks; k_slope_twi_fix, a few
// workgroups striding over the tile marks afterwards, recomputes those cells with the exact per-cell
// functions (dt_slope_cell / dt_slope_rad / dt_twi_cell: true float64 divisions, table logarithms,
// library fall-backs) and overwrites them.  ~1e-7 of the cells of a terrain raster are flagged.  The hot
// kernel has no slow-path code, no scratch, <= 64 VGPRs (8 waves per SIMD) and 19 KiB of LDS.
//
// The columns left and right of a lane's 4-wide patch are its neighbours' own registers: a lane's row is
// one ds_read_b128 plus two DPP wave shifts (lane 0 / 63 take the tile's halo columns from a
// wave-uniform LDS address), instead of two bank-conflicting ds_read_b32.
// ===========================================================================================
#define SD_MID 16u /* flag |low 29 mantissa bits - 2^28| <= SD_MID */
// The fix-up kernels read the tile marks 256 at a time; SD_FIX_SPLIT workgroups share the marked tiles of one such
// chunk.  (With one workgroup per chunk a raster whose tiles are ALL marked -- slopes beyond 250 %, TI / MTI near 0 on
// many cells: 15 % of the cells of a rough synthetic DEM with 40 m pits at 10 m pixels -- ran on one workgroup per CU.)
#define SD_FIX_SPLIT 8

// value of `v` in the previous / next lane of the wave; lane 0 / lane 63 keep `edge`
__device__ __forceinline__ float sd_from_prev_lane(float edge, float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(edge), __float_as_int(v), 0x138 /* wave_shr:1 */,
                                                    0xF, 0xF, false));
}
__device__ __forceinline__ float sd_from_next_lane(float edge, float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(edge), __float_as_int(v), 0x130 /* wave_shl:1 */,
                                                    0xF, 0xF, false));
}

// slope.py:210-259 in product form; returns true when the cell must be redone exactly
__device__ __forceinline__ bool sd_slope_fast(float c, float nw, float n, float ne, float w, float e, float sw,
                                              float s, float se, double kc, double kd, float &slope) {
  float cb = fmaxf(fmaxf(fmaxf(c - n, c - w), fmaxf(c - e, c - s)), 0.0f);
  float db = fmaxf(fmaxf(fmaxf(c - nw, c - ne), fmaxf(c - sw, c - se)), 0.0f);
  double qc = (double)cb * kc, qd = (double)db * kd;
  double q = qc > qd ? qc : qd;
  unsigned long long bits = (unsigned long long)__double_as_longlong(q);
  uint32_t lo = (uint32_t)bits, hi = (uint32_t)(bits >> 32);
  bool near_mid = ((lo & 0x1FFFFFFFu) - (0x10000000u - SD_MID)) <= 2u * SD_MID;
  bool odd = ((hi >> 20) - 897u) > 253u && bits != 0ull;  // not a normal float32 (and not 0): inf, NaN, tiny
  slope = (float)q;
  return near_mid || odd;
}

// TI / MTI (topoindexes.py:234-295) straight from q = tan(slope angle) = slope % / 100, without the arctangent:
//   tan(atan q + 0.01) = (q + t) / (1 - q t), t = tan 0.01
//   TI = ln A - ln(q + t) + ln(1 - q t),  MTI = n ln A - ...,  A = max(fac, 1) px^2
// Three hardware log2 of mantissas in [0.5, 1) (absolute error <= 6e-8 each) and exact exponents, summed in
// float64 in the log2 domain: ~2e-7 absolute whatever the magnitudes, accepted only when |TI|, |MTI| >=
// DT_FAST_MIN (<= 1e-6 relative).  The reference goes through the float32 rounding of the angle
// (Example/example.py:63), a perturbation of <= 4e-7 absolute for angles <= 1.19 (q <= 2.5); steeper cells, a
// negative or non-finite q and fac < 0 are flagged.  Returns true when the cell must be redone exactly
// (fac <= -100, the nodata of topoindexes.py:252, is handled by the caller).
template <typename AccT>
__device__ __forceinline__ bool sd_twi_fast(AccT fac, float q, double n, double lnpx2, double nlnpx2, float &ti,
                                            float &mti) {
  const float TAN001 = 0.010000333346667207f;
  const float u = q + TAN001, v = fmaf(-q, TAN001, 1.0f);
  // fac == 0 -> 1 (topoindexes.py:256): ln 1 = 1 + log2(0.5) = 0.  (An accumulation above 2^24 is rounded to float32
  // here, 6e-8 relative: within the fast path's error budget for either width.)
  const float ff = (float)(fac > 1 ? fac : (AccT)1);
  const float lf = __log2f(__builtin_amdgcn_frexp_mantf(ff));
  const float l1 = __log2f(__builtin_amdgcn_frexp_mantf(u));
  const float l2 = __log2f(__builtin_amdgcn_frexp_mantf(v));
  const int ef = __builtin_amdgcn_frexp_expf(ff);
  const int e12 = __builtin_amdgcn_frexp_expf(u) - __builtin_amdgcn_frexp_expf(v);
  const double a2 = (double)ef + (double)lf;            // log2 fac
  const double s2 = (double)e12 + (double)(l1 - l2);    // log2 tan(angle + 0.01)
  const double LN2 = 0.6931471805599453;
  ti = (float)fma(a2 - s2, LN2, lnpx2);
  mti = (float)fma(fma(n, a2, -s2), LN2, nlnpx2);
  return !(fac >= 0 && q >= 0.0f && q <= 2.5f && fabsf(ti) >= (float)DT_FAST_MIN && fabsf(mti) >= (float)DT_FAST_MIN);
}

typedef float sd_v4f __attribute__((ext_vector_type(4)));
typedef int sd_v4i __attribute__((ext_vector_type(4)));
typedef long long sd_v2l __attribute__((ext_vector_type(2)));
// Cache policy of the fused stencil's streams (POL; DT_DBG_TWI_PLAIN selects it for A/B runs):
//   0 plain loads, plain stores      1 nt loads, nt stores (the default)      2 nt loads, plain stores
//   3 nt loads, sc1 stores (write-through: the bytes leave L2 in issue order instead of eviction order)
//   4 nt loads, sc0 sc1 stores       5 plain loads, nt stores
template <int POL>
__device__ __forceinline__ void sd_store4(float *p, float a, float b, float c, float d) {
  sd_v4f v = {a, b, c, d};
  if (POL == 1 || POL == 5) __builtin_nontemporal_store(v, reinterpret_cast<sd_v4f *>(p));
  else if (POL == 3) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  else if (POL == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
  else *reinterpret_cast<sd_v4f *>(p) = v;
}

// WX = waves of a workgroup side by side: the tile is 256 WX columns x 16 / WX rows (256 x 16, 512 x 8 or 1024 x 4;
// always 4096 cells, one 4 x 4 patch per lane, a wave = 256 columns x 4 rows).  AccT = width of the accumulation raster.
template <bool W_SLOPE, bool W_RAD, int POL, typename AccT, int WX>
__global__ __launch_bounds__(256, WX == 1 ? 8 : (WX == 2 ? 7 : 6)) void k_slope_twi(const float *__restrict__ dem, DtWin w, double kc, double kd,
                                                     float *__restrict__ slope, float *__restrict__ slope_rad,
                                                     const AccT *__restrict__ acc32, double n_top, double lnpx2,
                                                     float *__restrict__ ti, float *__restrict__ mti, int tiles_x,
                                                     int tiles_y, int vec_ok, uint8_t *__restrict__ tile_mark,
                                                     uint16_t *__restrict__ lane_mask, uint32_t flag_all,
                                                     int map_mode) {
  constexpr int TX = SD_TX * WX, TY = SD_TY / WX, LDW = TX + 8;
  constexpr bool NT = POL >= 1 && POL <= 4;  // non-temporal loads of the accumulation raster
  __shared__ __attribute__((aligned(16))) float t[(TY + 2) * LDW];
  const double nlnpx2 = n_top * lnpx2;
  // (tried: identity and row-interleaved block -> tile maps instead of one band per XCD: 2 % slower)
  const int tile = map_mode ? sd_tile_of_block_x(blockIdx.x, tiles_x, tiles_y, map_mode)
                            : sd_tile_of_block(blockIdx.x, tiles_x * tiles_y);
  const int tyi = tile / tiles_x, txi = tile - tyi * tiles_x;
  const int x0 = txi * TX, y0 = tyi * TY;
  const int H = w.H, W = w.W;
  const float pinf = __builtin_inff();
  sd_stage<TX, TY>(t, dem, w, x0, y0, vec_ok);
  __syncthreads();

  const int tx = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int wx0 = (wv % WX) * 256;              // tile column of the wave's first cell
  const int cx = wx0 + tx * 4, ry = (wv / WX) * 4;  // tile column / row of the lane's 4 x 4 patch
  const int gx = x0 + cx;
  // every lane stays active to the end (its neighbours' DPP reads need it); stores are guarded
  auto load_row = [&](int lr, float *dst) {
    const float *row = &t[lr * LDW];
    float4 m = *reinterpret_cast<const float4 *>(row + 4 + cx);
    float lh = row[3 + wx0], rh = row[4 + wx0 + 256];  // columns beside the wave: wave-uniform address (broadcast)
    dst[0] = sd_from_prev_lane(lh, m.w);
    dst[1] = m.x;
    dst[2] = m.y;
    dst[3] = m.z;
    dst[4] = m.w;
    dst[5] = sd_from_next_lane(rh, m.x);
  };
  const bool full = vec_ok && gx + 3 < W;
  float a[6], bb[6], cc[6];
  load_row(ry, a);
  load_row(ry + 1, bb);
  uint32_t mask = flag_all;  // test knob: 0xFFFF sends every cell through the exact path as well
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int gy = y0 + ry + j;
    const long long o = (long long)gy * w.ld + gx;
    AccT fv[4] = {(AccT)-100, (AccT)-100, (AccT)-100, (AccT)-100};
    if (gy < H) {
      const AccT *pf = acc32 + o;
      if (full && sizeof(AccT) == 8) {
        const sd_v2l *p2 = reinterpret_cast<const sd_v2l *>(pf);
        sd_v2l a2 = NT ? __builtin_nontemporal_load(p2) : p2[0], b2 = NT ? __builtin_nontemporal_load(p2 + 1) : p2[1];
        fv[0] = (AccT)a2.x; fv[1] = (AccT)a2.y; fv[2] = (AccT)b2.x; fv[3] = (AccT)b2.y;
      } else if (full) {
        sd_v4i f4 = NT ? __builtin_nontemporal_load(reinterpret_cast<const sd_v4i *>(pf))
                       : *reinterpret_cast<const sd_v4i *>(pf);
        fv[0] = (AccT)f4.x; fv[1] = (AccT)f4.y; fv[2] = (AccT)f4.z; fv[3] = (AccT)f4.w;
      } else {
        if (gx < W) fv[0] = pf[0];
        if (gx + 1 < W) fv[1] = pf[1];
        if (gx + 2 < W) fv[2] = pf[2];
        if (gx + 3 < W) fv[3] = pf[3];
      }
    }
    load_row(ry + 2 + j, cc);
    float so[4], ro[4], tio[4], mtio[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const float c = bb[k + 1];
      const float cz = c == pinf ? DT_NODATA : c;  // the centre's own value
      float sl, rad = 0.0f, tv, mv;
      bool flag = sd_slope_fast(c, a[k], a[k + 1], a[k + 2], bb[k], bb[k + 2], cc[k], cc[k + 1], cc[k + 2], kc, kd, sl);
      const bool snod = cz <= DT_NODATA;  // slope.py:231
      sl = snod ? DT_NODATA : sl;
      flag = flag && !snod;
      const float q = dt_pct_to_tan(sl);
      if (W_RAD) {  // dt_slope_rad: -100 where dem == -100, else the arctangent (q outside its domain: flagged)
        const bool rnod = cz == DT_NODATA;
        rad = rnod ? DT_NODATA : (float)dt_atanf_pos(q);
        flag = flag || (!rnod && !(q >= 0.0f && q < 1e30f));
      }
      const AccT f = fv[k];
      const bool tnod = f <= (AccT)-100;  // topoindexes.py:252
      flag = (sd_twi_fast(f, q, n_top, lnpx2, nlnpx2, tv, mv) && !tnod) || flag;
      so[k] = sl;
      ro[k] = rad;
      tio[k] = tnod ? DT_NODATA : tv;
      mtio[k] = tnod ? DT_NODATA : mv;
      mask |= (flag ? 1u : 0u) << (4 * j + k);
    }
    if (gy < H) {
      if (full) {
        if (W_SLOPE) sd_store4<POL>(slope + o, so[0], so[1], so[2], so[3]);
        if (W_RAD) sd_store4<POL>(slope_rad + o, ro[0], ro[1], ro[2], ro[3]);
        sd_store4<POL>(ti + o, tio[0], tio[1], tio[2], tio[3]);
        sd_store4<POL>(mti + o, mtio[0], mtio[1], mtio[2], mtio[3]);
      } else {
#pragma unroll
        for (int k = 0; k < 4; k++) {
          if (gx + k < W) {
            if (W_SLOPE) slope[o + k] = so[k];
            if (W_RAD) slope_rad[o + k] = ro[k];
            ti[o + k] = tio[k];
            mti[o + k] = mtio[k];
          } else {
            mask &= ~(1u << (4 * j + k));
          }
        }
      }
    } else {
      mask &= ~(0xFu << (4 * j));
    }
#pragma unroll
    for (int q = 0; q < 6; q++) {
      a[q] = bb[q];
      bb[q] = cc[q];
    }
  }
  const int any = __syncthreads_or(mask != 0u);
  if (threadIdx.x == 0) tile_mark[tile] = (uint8_t)(any != 0);
  if (any) lane_mask[(size_t)tile * 256 + threadIdx.x] = (uint16_t)mask;
}

// the cold half: exact recomputation of the flagged cells (a handful per raster)
template <typename AccT, int WX>
__global__ __launch_bounds__(256) void k_slope_twi_fix(const float *__restrict__ dem, DtWin w, double px,
                                                      float *__restrict__ slope, float *__restrict__ slope_rad,
                                                      const AccT *__restrict__ acc32, double n_top, double lnpx2,
                                                      float *__restrict__ ti, float *__restrict__ mti, int tiles_x,
                                                      int tiles_y, int vec_ok, const uint8_t *__restrict__ tile_mark,
                                                      const uint16_t *__restrict__ lane_mask,
                                                      const DtLogEntry *__restrict__ g_tab) {
  constexpr int TX = SD_TX * WX, TY = SD_TY / WX, LDW = TX + 8;
  __shared__ __attribute__((aligned(16))) float t[(TY + 2) * LDW];
  const int ntiles = tiles_x * tiles_y;
  const double dcard = px, ddiag = px * sqrt(2.0);
  const double inv_card = 1.0 / dcard, inv_diag = 1.0 / ddiag;
  const float pinf = __builtin_inff();
  // 256 tile marks per step, one per lane: an unmarked raster costs ceil(ntiles / 256) independent byte loads
  __shared__ uint8_t s_mark[256];
  for (int chunk = blockIdx.x / SD_FIX_SPLIT; chunk * 256 < ntiles; chunk += gridDim.x / SD_FIX_SPLIT) {
    const int mine = chunk * 256 + (int)threadIdx.x;
    const uint8_t m = mine < ntiles ? tile_mark[mine] : (uint8_t)0;
    __syncthreads();  // the previous chunk's readers are done with s_mark
    s_mark[threadIdx.x] = m;
    if (!__syncthreads_or(m)) continue;
    for (int i = blockIdx.x % SD_FIX_SPLIT; i < 256; i += SD_FIX_SPLIT) {
      if (!s_mark[i]) continue;  // block-uniform
      const int tile = chunk * 256 + i;
      const int tyi = tile / tiles_x, txi = tile - tyi * tiles_x;
      const int x0 = txi * TX, y0 = tyi * TY;
      __syncthreads();  // the previous tile's readers are done with t
      sd_stage<TX, TY>(t, dem, w, x0, y0, vec_ok);
      __syncthreads();
      uint32_t mask = lane_mask[(size_t)tile * 256 + threadIdx.x];
      const int wv = threadIdx.x >> 6;
      const int cx = (wv % WX) * 256 + (threadIdx.x & 63) * 4, ry = (wv / WX) * 4;
      while (mask) {
        const int bit = __ffs((int)mask) - 1;
        mask &= mask - 1u;
        const int j = bit >> 2, k = bit & 3;
        const int gy = y0 + ry + j, gx = x0 + cx + k;
        if (gy >= w.H || gx >= w.W) continue;
        const float *p = &t[(ry + j + 1) * LDW + 4 + cx + k];  // the centre in the staged tile
        const float cz = p[0] == pinf ? DT_NODATA : p[0];
        SlopeCell sc = dt_slope_cell<false, true>(cz, p[-LDW - 1], p[-LDW], p[-LDW + 1], p[-1], p[1],
                                                  p[LDW - 1], p[LDW], p[LDW + 1], inv_card, inv_diag, dcard,
                                                  ddiag);
        const long long o = (long long)gy * w.ld + gx;
        const float rad = dt_slope_rad(sc.slope, cz);
        float tv, mv;
        dt_twi_cell((int64_t)acc32[o], rad, lnpx2, n_top, tv, mv, g_tab);
        if (slope) slope[o] = sc.slope;
        if (slope_rad) slope_rad[o] = rad;
        ti[o] = tv;
        mti[o] = mv;
      }
    }
  }
}

// ===========================================================================================
// D8 alone (N1), hot / cold like k_slope_twi: the chain's first kernel writes only the direction codes.
// Within a class the first strict maximum of the float32 differences wins (scan order N, W, E, S / NW, NE, SW, SE);
// between the classes the reference compares cb / px with db / (px sqrt 2), i.e. cb with db / sqrt 2: decided in
// float32 whenever the two differ by more than 2^-21 relative (the float32 product is within 2^-23 of db / sqrt 2),
// flagged for the exact float64 path of dt_slope_cell otherwise (~1e-6 of the cells; equality is impossible for
// non-zero differences, the ratio being irrational).
// ===========================================================================================
__device__ __forceinline__ bool sd_d8_fast(float c, float nw, float n, float ne, float w, float e, float sw, float s,
                                           float se, uint32_t &code) {
  float cb = 0.0f, db = 0.0f;
  uint32_t ccode = 0, dcode = 0;
#define SD_CAND(nb, best, bcode, code_) \
  {                                     \
    const float d_ = c - (nb);          \
    const bool up_ = d_ > best;         \
    best = up_ ? d_ : best;             \
    bcode = up_ ? code_ : bcode;        \
  }
  SD_CAND(n, cb, ccode, 64u)
  SD_CAND(w, cb, ccode, 16u)
  SD_CAND(e, cb, ccode, 1u)
  SD_CAND(s, cb, ccode, 4u)
  SD_CAND(nw, db, dcode, 32u)
  SD_CAND(ne, db, dcode, 128u)
  SD_CAND(sw, db, dcode, 8u)
  SD_CAND(se, db, dcode, 2u)
#undef SD_CAND
  const float t = db * 0.70710678118654752f;
  const float hi = fmaxf(cb, t);
  code = cb > t ? ccode : dcode;  // both zero: dcode == 0
  // not finite / tiny differences go to the exact path as well (inf - finite = inf would compare equal)
  return hi != 0.0f && (fabsf(cb - t) <= hi * 4.76837158e-7f || !(hi < 3.0e38f) || hi < 1.0e-30f);
}

// nod4 (may be NULL): the nodata mask, one 16-bit word per 4 x 4 patch of cells -- bit 4 j + k = cell (4 r + j, 4 i + k)
// holds the nodata sentinel (z <= -100; a non-finite height counts as nodata here, as for the codes) -- what the flow-accumulation
// pass needs of the DEM (-100 on nodata cells), so that it reads 0.125 instead of 4 bytes per cell (dt_dev_slope_d8_m /
// dt_dev_flowacc_river_flowhand_local_m).  A patch is what one thread of this kernel owns: ONE store per thread (four
// byte stores, a row each, cost this issue-bound kernel 12 %).  ldm = words per row of patches.  Only for a single
// raster (window origin on the 4-cell grid).
template <bool NT>
__global__ __launch_bounds__(256, 8) void k_d8(const float *__restrict__ dem, DtWin w, uint8_t *__restrict__ fdr,
                                              int tiles_x, int tiles_y, int vec_ok, uint8_t *__restrict__ tile_mark,
                                              uint16_t *__restrict__ lane_mask, uint8_t *__restrict__ nod4, int ldm) {
  __shared__ __attribute__((aligned(16))) float t[(SD_TY + 2) * SD_LDW];
  const int tile = sd_tile_of_block(blockIdx.x, tiles_x * tiles_y);
  const int tyi = tile / tiles_x, txi = tile - tyi * tiles_x;
  const int x0 = txi * SD_TX, y0 = tyi * SD_TY;
  const int H = w.H, W = w.W;
  const float pinf = __builtin_inff();
  sd_stage(t, dem, w, x0, y0, vec_ok);
  __syncthreads();
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int cx = tx * 4, ry = ty * 4;
  const int gx = x0 + cx;
  auto load_row = [&](int lr, float *dst) {
    const float *row = &t[lr * SD_LDW];
    float4 m = *reinterpret_cast<const float4 *>(row + 4 + cx);
    float lh = row[3], rh = row[4 + SD_TX];
    dst[0] = sd_from_prev_lane(lh, m.w);
    dst[1] = m.x;
    dst[2] = m.y;
    dst[3] = m.z;
    dst[4] = m.w;
    dst[5] = sd_from_next_lane(rh, m.x);
  };
  const bool full = vec_ok && gx + 3 < W;
  // block-uniform: does the tile touch the border of the GLOBAL raster?  (only there does the border rule apply;
  // the kernel is limited by VALU issue and the rule is a sixth of a cell's instructions)
  const bool on_border = w.gy0 + y0 == 0 || w.gx0 + x0 == 0 || w.gy0 + y0 + SD_TY >= w.Hg || w.gx0 + x0 + SD_TX >= w.Wg;
  float a[6], bb[6], cc[6];
  load_row(ry, a);
  load_row(ry + 1, bb);
  uint32_t mask = 0, nodmask = 0;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int gy = y0 + ry + j;
    load_row(ry + 2 + j, cc);
    uint32_t codes = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const float c = bb[k + 1];
      const bool nod = !(c < pinf) || c <= DT_NODATA;  // staged nodata (+inf), NaN, or below the sentinel: code 0
      nodmask |= (nod ? 1u : 0u) << (4 * j + k);
      uint32_t code;
      bool flag = sd_d8_fast(c, a[k], a[k + 1], a[k + 2], bb[k], bb[k + 2], cc[k], cc[k + 1], cc[k + 2], code);
      if (on_border) {
        // N1 border rule: a border cell with no lower neighbour drains out of the raster
        const int gyy = w.gy0 + gy, gxx = w.gx0 + gx + k;
        const uint32_t out = gyy == w.Hg - 1 ? 4u : (gyy == 0 ? 64u : (gxx == 0 ? 16u : (gxx == w.Wg - 1 ? 1u : 0u)));
        code = code == 0u ? out : code;
      }
      codes |= (nod ? 0u : code) << (8 * k);
      mask |= ((flag && !nod) ? 1u : 0u) << (4 * j + k);
    }
    if (gy < H) {
      const long long o = (long long)gy * w.ld + gx;
      if (full) {
        if (NT) __builtin_nontemporal_store(codes, reinterpret_cast<uint32_t *>(fdr + o));
        else *reinterpret_cast<uint32_t *>(fdr + o) = codes;
      } else {
#pragma unroll
        for (int k = 0; k < 4; k++) {
          if (gx + k < W) fdr[o + k] = (uint8_t)(codes >> (8 * k));
          else mask &= ~(1u << (4 * j + k));
        }
      }
    } else {
      mask &= ~(0xFu << (4 * j));
    }
#pragma unroll
    for (int q = 0; q < 6; q++) {
      a[q] = bb[q];
      bb[q] = cc[q];
    }
  }
  if (nod4 && gx < W && y0 + ry < H)
    reinterpret_cast<uint16_t *>(nod4)[(long long)((y0 + ry) >> 2) * ldm + (gx >> 2)] = (uint16_t)nodmask;
  const int any = __syncthreads_or(mask != 0u);
  if (threadIdx.x == 0) tile_mark[tile] = (uint8_t)(any != 0);
  if (any) lane_mask[(size_t)tile * 256 + threadIdx.x] = (uint16_t)mask;
}

__global__ __launch_bounds__(256) void k_d8_fix(const float *__restrict__ dem, DtWin w, double px,
                                               uint8_t *__restrict__ fdr, int tiles_x, int tiles_y, int vec_ok,
                                               const uint8_t *__restrict__ tile_mark,
                                               const uint16_t *__restrict__ lane_mask) {
  __shared__ __attribute__((aligned(16))) float t[(SD_TY + 2) * SD_LDW];
  __shared__ uint8_t s_mark[256];
  const int ntiles = tiles_x * tiles_y;
  const double dcard = px, ddiag = px * sqrt(2.0);
  const double inv_card = 1.0 / dcard, inv_diag = 1.0 / ddiag;
  const float pinf = __builtin_inff();
  for (int chunk = blockIdx.x / SD_FIX_SPLIT; chunk * 256 < ntiles; chunk += gridDim.x / SD_FIX_SPLIT) {
    const int mine = chunk * 256 + (int)threadIdx.x;
    const uint8_t m = mine < ntiles ? tile_mark[mine] : (uint8_t)0;
    __syncthreads();
    s_mark[threadIdx.x] = m;
    if (!__syncthreads_or(m)) continue;
    for (int i = blockIdx.x % SD_FIX_SPLIT; i < 256; i += SD_FIX_SPLIT) {
      if (!s_mark[i]) continue;  // block-uniform
      const int tile = chunk * 256 + i;
      const int tyi = tile / tiles_x, txi = tile - tyi * tiles_x;
      const int x0 = txi * SD_TX, y0 = tyi * SD_TY;
      __syncthreads();
      sd_stage(t, dem, w, x0, y0, vec_ok);
      __syncthreads();
      uint32_t mask = lane_mask[(size_t)tile * 256 + threadIdx.x];
      const int cx = (threadIdx.x & 63) * 4, ry = (threadIdx.x >> 6) * 4;
      while (mask) {
        const int bit = __ffs((int)mask) - 1;
        mask &= mask - 1u;
        const int j = bit >> 2, k = bit & 3;
        const int gy = y0 + ry + j, gx = x0 + cx + k;
        if (gy >= w.H || gx >= w.W) continue;
        const float *p = &t[(ry + j + 1) * SD_LDW + 4 + cx + k];
        const float cz = p[0] == pinf ? DT_NODATA : p[0];
        SlopeCell sc = dt_slope_cell<true, false>(cz, p[-SD_LDW - 1], p[-SD_LDW], p[-SD_LDW + 1], p[-1], p[1],
                                                  p[SD_LDW - 1], p[SD_LDW], p[SD_LDW + 1], inv_card, inv_diag, dcard,
                                                  ddiag);
        uint32_t code = sc.code;
        const int gyy = w.gy0 + gy, gxx = w.gx0 + gx;
        if (code == 0u && cz > DT_NODATA) {
          if (gyy == w.Hg - 1) code = 4u;
          else if (gyy == 0) code = 64u;
          else if (gxx == 0) code = 16u;
          else if (gxx == w.Wg - 1) code = 1u;
        }
        fdr[(long long)gy * w.ld + gx] = (uint8_t)code;
      }
    }
  }
}

// bytes of the mark / mask workspace of the fused slope + TI + MTI launch for an H x W window
size_t dt_stencil_aux_bytes(int64_t H, int64_t W) {
  int64_t ntiles = 0;  // the largest tile count of the three tile geometries (they differ on ragged rasters)
  for (int wx = 1; wx <= 4; wx *= 2) {
    const int64_t tx = SD_TX * wx, ty = SD_TY / wx, n = ((W + tx - 1) / tx) * ((H + ty - 1) / ty);
    ntiles = n > ntiles ? n : ntiles;
  }
  return dt_align256((size_t)ntiles) + (size_t)ntiles * 512;
}

// the fused slope + TI + MTI pair (hot kernel + fix-up of the flagged cells) for one tile geometry / accumulation width
template <typename AccT, int WX>
static int launch_slope_twi(hipStream_t s, const DtWin &w, const float *dem, double px, float *slope, float *slope_rad,
                            const AccT *acc, double n_top, float *ti, float *mti, void *aux, int vec_ok) {
  constexpr int TX = SD_TX * WX, TY = SD_TY / WX;
  const int tiles_x = (int)((w.W + TX - 1) / TX), tiles_y = (int)((w.H + TY - 1) / TY);
  const int64_t ntiles = (int64_t)tiles_x * tiles_y;
  DT_REQUIRE(ntiles < (1ll << 31), "raster too large for one launch");
  DT_REQUIRE(aux != nullptr, "fused TWI needs its mark / mask workspace");
  dim3 g((unsigned)ntiles), b(256);
  const bool ws = slope != nullptr, wr = slope_rad != nullptr;
  uint8_t *mark = (uint8_t *)aux;
  uint16_t *lmask = (uint16_t *)((char *)aux + dt_align256((size_t)ntiles));
  const double kc = 100.0 / px, kd = 100.0 / (px * sqrt(2.0)), lnpx2 = log(px * px);
  const DtLogEntry *g_tab = dt_math_device_table(s);
#define DT_HOT(S, R, N)                                                                                             \
  hipLaunchKernelGGL((k_slope_twi<S, R, N, AccT, WX>), g, b, 0, s, dem, w, kc, kd, slope, slope_rad, acc, n_top, lnpx2, \
                     ti, mti, tiles_x, tiles_y, vec_ok, mark, lmask, dt_debug_get(DT_DBG_TWI_FLAG_ALL) ? 0xFFFFu : 0u, \
                     dt_debug_get(DT_DBG_TWI_MAP))
  // non-temporal loads of the accumulation raster and stores of the outputs (each byte is touched once):
  // 0.86 instead of 0.92 ms at 16384^2; the knob selects another cache policy (sd_store4) for A/B runs of the
  // benchmark's form of the kernel (slope + TI + MTI, int32 accumulation, 256 x 16 tiles)
  const int pol = dt_debug_get(DT_DBG_TWI_PLAIN);
  bool done = false;
  if constexpr (WX == 1 && sizeof(AccT) == 4) {
    if (ws && !wr && pol >= 1 && pol <= 5) {
      if (pol == 1) DT_HOT(true, false, 0);
      else if (pol == 2) DT_HOT(true, false, 2);
      else if (pol == 3) DT_HOT(true, false, 3);
      else if (pol == 4) DT_HOT(true, false, 4);
      else DT_HOT(true, false, 5);
      done = true;
    }
  }
  if (done) {
  } else if (ws && wr) DT_HOT(true, true, 1);
  else if (ws) DT_HOT(true, false, 1);
  else if (wr) DT_HOT(false, true, 1);
  else DT_HOT(false, false, 1);
#undef DT_HOT
  unsigned fix_blocks = SD_FIX_SPLIT * (unsigned)((ntiles + 255) / 256 < 1024 ? (ntiles + 255) / 256 : 1024);
  hipLaunchKernelGGL((k_slope_twi_fix<AccT, WX>), dim3(fix_blocks), b, 0, s, dem, w, px, slope, slope_rad, acc, n_top,
                     lnpx2, ti, mti, tiles_x, tiles_y, vec_ok, mark, lmask, g_tab);
  return DT_OK;
}

// `acc` (fused TI / MTI only): int32_t* raster, or int64_t* with acc64 != 0
int dt_launch_stencil(hipStream_t s, const DtWin &w, const float *dem, double px, float *slope,
                      uint8_t *fdr, float *slope_rad, const void *acc, int acc64, double n_top, float *ti,
                      float *mti, void *aux, uint8_t *nod4, int ldm) {
  const int64_t H = w.H, W = w.W;
  if (H == 0 || W == 0) return DT_OK;
  int tiles_x = (int)((W + SD_TX - 1) / SD_TX), tiles_y = (int)((H + SD_TY - 1) / SD_TY);
  int64_t ntiles = (int64_t)tiles_x * tiles_y;
  DT_REQUIRE(ntiles < (1ll << 31), "raster too large for one launch");
  // 16-byte vector path needs W % 4 == 0 and 16-byte aligned bases
  int vec_ok = (W % 4 == 0) && (w.ld % 4 == 0) && (((uintptr_t)dem & 15) == 0) && (!slope || ((uintptr_t)slope & 15) == 0) &&
               (!slope_rad || ((uintptr_t)slope_rad & 15) == 0) && (!ti || ((uintptr_t)ti & 15) == 0) &&
               (!mti || ((uintptr_t)mti & 15) == 0) && (!fdr || ((uintptr_t)fdr & 3) == 0) &&
               (!acc || ((uintptr_t)acc & 15) == 0);
  dim3 g((unsigned)ntiles), b(256);
  bool ws = slope != nullptr, wf = fdr != nullptr, wr = slope_rad != nullptr, wt = ti != nullptr;
  DT_REQUIRE(!nod4 || (wf && aux && !ws && !wr && !wt),
             "the nodata mask comes from the D8-only kernel (fdr and a workspace, no other output)");
#define DT_GO(S, F, R) \
  hipLaunchKernelGGL((k_stencil<S, F, R>), g, b, 0, s, dem, w, px, slope, fdr, slope_rad, tiles_x, tiles_y, vec_ok)
  if (wt) {
    DT_REQUIRE(acc && mti, "fused TWI needs the accumulation raster, ti and mti");
    // tile geometry (DT_DBG_TWI_WX: 1, 2 or 4 waves side by side; 0 = default), narrowed for narrow rasters
    int wx = dt_debug_get(DT_DBG_TWI_WX);
    if (wx != 1 && wx != 2 && wx != 4) wx = DT_TWI_WX_DEFAULT;
    while (wx > 1 && W < (int64_t)SD_TX * wx) wx >>= 1;
    if (acc64) {
      const long long *a = (const long long *)acc;
      if (wx == 4) return launch_slope_twi<long long, 4>(s, w, dem, px, slope, slope_rad, a, n_top, ti, mti, aux, vec_ok);
      if (wx == 2) return launch_slope_twi<long long, 2>(s, w, dem, px, slope, slope_rad, a, n_top, ti, mti, aux, vec_ok);
      return launch_slope_twi<long long, 1>(s, w, dem, px, slope, slope_rad, a, n_top, ti, mti, aux, vec_ok);
    }
    const int32_t *a = (const int32_t *)acc;
    if (wx == 4) return launch_slope_twi<int32_t, 4>(s, w, dem, px, slope, slope_rad, a, n_top, ti, mti, aux, vec_ok);
    if (wx == 2) return launch_slope_twi<int32_t, 2>(s, w, dem, px, slope, slope_rad, a, n_top, ti, mti, aux, vec_ok);
    return launch_slope_twi<int32_t, 1>(s, w, dem, px, slope, slope_rad, a, n_top, ti, mti, aux, vec_ok);
  } else if (ws && wf && wr) DT_GO(true, true, true);
  else if (ws && wf) DT_GO(true, true, false);
  else if (ws && wr) DT_GO(true, false, true);
  else if (wf && wr) DT_GO(false, true, true);
  else if (ws) DT_GO(true, false, false);
  else if (wf && aux) {  // D8 alone with a workspace: the hot / cold pair
    uint8_t *mark = (uint8_t *)aux;
    uint16_t *lmask = (uint16_t *)((char *)aux + dt_align256((size_t)ntiles));
    hipLaunchKernelGGL(k_d8<false>, g, b, 0, s, dem, w, fdr, tiles_x, tiles_y, vec_ok, mark, lmask, nod4, ldm);
    unsigned fix_blocks = SD_FIX_SPLIT * (unsigned)((ntiles + 255) / 256 < 1024 ? (ntiles + 255) / 256 : 1024);
    hipLaunchKernelGGL(k_d8_fix, dim3(fix_blocks), b, 0, s, dem, w, px, fdr, tiles_x, tiles_y, vec_ok, mark, lmask);
  } else if (wf) DT_GO(false, true, false);
  else if (wr) DT_GO(false, false, true);
#undef DT_GO
  return DT_OK;
}
