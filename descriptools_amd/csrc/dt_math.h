// dt_math.h -- float64 ln / ln(tan) / atan for the descriptor kernels, ~3-4x fewer instructions than the
// device library's log / tan / pow / atan while staying far inside what the float32 outputs can
// resolve (the reference rounds ONE float64 expression to float32; a float64 error of 1e-13 moves
// that rounding in ~1 cell per 10^6, and by one float32 ulp).
//
//   dt_fast_log(x)    x = m * 2^e;  128-entry table on the top 7 mantissa bits: ln m = ln c_i +
//                     ln(1 + u), u = m / c_i - 1, |u| <= 2^-8, degree-5 series.  rel. error ~2e-16.
//   dt_fast_lntan(y)  y in (0, pi/2): ln tan y = +-(ln z + g(z^2)), z = min(y, pi/2 - y),
//                     g = degree-10 polynomial (dt_math_coeffs.h), abs error ~6e-14.
//   dt_fast_atan(q)   q >= 0: one division into [0, tan(pi/8)], degree-7 polynomial, abs error ~1e-12.
// Arguments outside those domains (negative / NaN / inf / subnormal) take the device library's
// functions, so NaN and inf propagate exactly as before.
#pragma once
#include <hip/hip_runtime.h>

#include "dt_math_coeffs.h"

#define DT_LOGTAB_N 128
struct DtLogEntry {
  double rc;   // ~ 1 / c_i,  c_i = 1 + (i + 0.5) / 128
  double lnc;  // -ln(rc)
};

// fills `tab` (host) -- uploaded once per process by dt_capi.hip
void dt_math_host_table(DtLogEntry *tab);
// device copy of the table (global memory); kernels stage it into LDS with dt_math_stage()
const DtLogEntry *dt_math_device_table(hipStream_t s);

__device__ __forceinline__ void dt_math_stage(const DtLogEntry *__restrict__ g, DtLogEntry *s_tab) {
  for (int i = threadIdx.x; i < DT_LOGTAB_N; i += blockDim.x) s_tab[i] = g[i];
}

__device__ __forceinline__ double dt_fast_log(double x, const DtLogEntry *s_tab) {
  unsigned long long b = (unsigned long long)__double_as_longlong(x);
  unsigned hi = (unsigned)(b >> 32);
  unsigned ex = hi >> 20;  // sign + exponent
  if (ex - 1u >= 0x7FEu) return log(x);  // zero, subnormal, negative, inf, NaN: library semantics
  int e = (int)ex - 1023;
  unsigned idx = (hi >> 13) & 127u;
  double m = __longlong_as_double((long long)((b & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull));
  DtLogEntry t = s_tab[idx];
  double u = fma(m, t.rc, -1.0);
  double p = fma(u, 0.2, -0.25);
  p = fma(u, p, 1.0 / 3.0);
  p = fma(u, p, -0.5);
  p = fma(u, p, 1.0);
  p = u * p;
  return fma((double)e, 0.6931471805599453, t.lnc + p);
}

__device__ __forceinline__ double dt_fast_lntan(double y, const DtLogEntry *s_tab) {
  const double HALF_PI = 1.5707963267948966, QUARTER_PI = 0.7853981633974483;
  if (!(y > 0.0 && y < HALF_PI)) return log(tan(y));
  bool lo = y <= QUARTER_PI;
  double z = lo ? y : HALF_PI - y;
  if (!(z > 1e-300)) return log(tan(y));
  double t = fma(z * z, DT_LNTAN_SCALE, -1.0);
  double g = dt_lntan_c[DT_LNTAN_DEG];
#pragma unroll
  for (int k = DT_LNTAN_DEG - 1; k >= 0; k--) g = fma(g, t, dt_lntan_c[k]);
  double r = dt_fast_log(z, s_tab) + g;
  return lo ? r : -r;
}

// fast reciprocal (rel. error ~1e-16): hardware estimate + two Newton steps
__device__ __forceinline__ double dt_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}

__device__ __forceinline__ double dt_fast_atan(double q) {
  const double T8 = 0.41421356237309503, T38 = 2.414213562373095;  // tan(pi/8), tan(3 pi/8)
  if (!(q >= 0.0 && q < 1e300)) return atan(q);
  double x, base, sgn = 1.0;
  if (q <= T8) {
    x = q;
    base = 0.0;
  } else if (q < T38) {
    x = (q - 1.0) * dt_rcp(q + 1.0);
    base = 0.7853981633974483;
  } else {
    x = dt_rcp(q);
    base = 1.5707963267948966;
    sgn = -1.0;
  }
  double t = fma(x * x, DT_ATAN_SCALE, -1.0);
  double h = dt_atan_c[DT_ATAN_DEG];
#pragma unroll
  for (int k = DT_ATAN_DEG - 1; k >= 0; k--) h = fma(h, t, dt_atan_c[k]);
  return fma(sgn * x, h, base);
}

// ---------------------------------------------------------------------------------------------------
// float32 fast path.  The descriptors are rounded to float32 and must stay within 1e-5 relative of the
// reference: each logarithm below is  e * ln2  (exact exponent, float64)  +  log2(mantissa) * ln2
// (hardware v_log_f32 on [0.5, 1): absolute error <= 6e-8), so a sum of three of them is good to
// ~2e-7 ABSOLUTE whatever the magnitude of the arguments.  Callers use the result only when
// |result| >= DT_FAST_MIN (relative error <= 1e-6) and recompute with the float64 routines above
// otherwise (results near a zero crossing, out-of-domain arguments).  ~8 instructions per logarithm
// instead of ~20 float64 ones: this is what lets the fused slope+TI+MTI and GFI kernels run at the
// HBM rate instead of the float64 ALU rate.
// ---------------------------------------------------------------------------------------------------
#define DT_FAST_MIN 0.25

// ln(x) for a positive, finite, normal float x
__device__ __forceinline__ double dt_lnf(float x) {
  float m = __builtin_amdgcn_frexp_mantf(x);  // [0.5, 1)
  int e = __builtin_amdgcn_frexp_expf(x);
  return fma((double)e, 0.6931471805599453, (double)__log2f(m) * 0.6931471805599453);
}

// ln(tan(y)) for y in [0.005, 1.25] (y given in float64: theta + 0.01)
__device__ __forceinline__ double dt_lntanf(double y) {
  const double HALF_PI = 1.5707963267948966;
  bool lo = y <= 0.7853981633974483;
  float z = (float)(lo ? y : HALF_PI - y);
  float t = fmaf(z * z, (float)DT_LNTAN_SCALE, -1.0f);
  float g = dt_lntan_c32[6];
#pragma unroll
  for (int k = 5; k >= 0; k--) g = fmaf(g, t, dt_lntan_c32[k]);
  double r = dt_lnf(z) + (double)g;
  return lo ? r : -r;
}

// atan(q), q >= 0 float32, without branches: reduction as dt_fast_atan (q, (q - 1) / (q + 1) or 1 / q into
// [-tan(pi/8), tan(pi/8)]) with the quotient from the hardware reciprocal and one Newton step (<= 0.51 ulp),
// degree-5 polynomial; the last multiply-add is float64 so that the result rounds to float32 within ~1 ulp
__device__ __forceinline__ double dt_atanf_pos(float q) {
  const float T8 = 0.41421356f, T38 = 2.41421356f;
  const bool lowr = q <= T8, high = !(q < T38);
  const float num = lowr ? q : (high ? 1.0f : q - 1.0f);
  const float den = lowr ? 1.0f : (high ? q : q + 1.0f);
  const float r0 = __builtin_amdgcn_rcpf(den);
  float x = num * r0;
  x = fmaf(fmaf(-den, x, num), r0, x);
  x = lowr ? q : x;
  const double base = lowr ? 0.0 : (high ? 1.5707963267948966 : 0.7853981633974483);
  float t = fmaf(x * x, (float)DT_ATAN_SCALE, -1.0f);
  float h = dt_atan_c32[5];
#pragma unroll
  for (int k = 4; k >= 0; k--) h = fmaf(h, t, dt_atan_c32[k]);
  const double xs = high ? -(double)x : (double)x;
  return fma(xs, (double)h, base);
}

// slope % / 100 in float32 (Example/example.py:63 divides in float32): product with the rounded reciprocal
// and one residual correction, <= 0.51 ulp from the quotient, 3 instructions instead of an IEEE division
__device__ __forceinline__ float dt_pct_to_tan(float slope_pct) {
  const float q0 = slope_pct * 0.01f;
  return fmaf(fmaf(-q0, 100.0f, slope_pct), 0.01f, q0);
}

#ifndef DT_NODATA
#define DT_NODATA (-100.0f)
#endif

// ln(x) in float64 without branches, ~2e-13 absolute: dt_fast_log's table with a degree-4 series.  Finite
// normal x > 0 go through the table; x == 0 gives -inf, x < 0 or NaN gives NaN, +inf gives +inf -- what
// log / pow of the reference give (gfi.py:292-294: an area of 0 makes GFI -inf, a negative one NaN).
__device__ __forceinline__ double dt_log_sel(double x, const DtLogEntry *s_tab) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(x);
  const unsigned hi = (unsigned)(b >> 32);
  const int e = (int)((hi >> 20) & 0x7FFu) - 1023;
  const DtLogEntry t = s_tab[(hi >> 13) & 127u];
  const double m = __longlong_as_double((long long)((b & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull));
  const double u = fma(m, t.rc, -1.0);
  double p = fma(u, -0.25, 1.0 / 3.0);
  p = fma(u, p, -0.5);
  p = fma(u, p, 1.0);
  double r = fma((double)e, 0.6931471805599453, fma(u, p, t.lnc));
  r = x > 0.0 ? r : (x == 0.0 ? -__builtin_inf() : __builtin_nan(""));
  return x == __builtin_inf() ? x : r;
}

// GFI and ln(hl/H) in one pass (gfi.py:268-294, :404-440): hand is read once and ln(h + 0.01) evaluated once.
// The indices cross zero inside ordinary terrain (8 % of the cells of the benchmark DEM lie within 0.25 of a
// zero), where a relative tolerance needs float64 logarithms: all three are the table logarithm above on exact
// integer -> float64 conversions of the areas; no fast / slow split, no divergence.
template <typename IT>  // int32_t or long long areas: both convert to float64 exactly (< 2^53)
__device__ __forceinline__ void dt_gfi_both_cell(float h, IT ar, IT f, double expo, double c0,
                                                 const DtLogEntry *s_tab, float &g_out, float &l_out) {
  const double lh = c0 - dt_log_sel((double)h + 0.01, s_tab);
  const double g = lh + expo * dt_log_sel((double)ar, s_tab);
  const double l = lh + (f == 0 ? 0.0 : expo * dt_log_sel((double)f, s_tab));
  const bool nod = h <= DT_NODATA;
  g_out = nod ? DT_NODATA : (float)g;
  l_out = nod ? DT_NODATA : (float)l;
}

// TI / MTI of one cell (topoindexes.py:234-295); float64 inside, float32 out.
//   TI  = ln(A / t)   = ln A - ln t,      A = a * px^2 (a = fac, 0 -> 1), t = tan(slope + 0.01)
//   MTI = ln(A^n / t) = n ln A - ln t
// evaluated from two logarithms and one tangent instead of pow + 2 log + 2 divisions (the float64
// difference to the reference's literal expression is ~1e-16 relative, invisible after the float32
// rounding except at rounding ties; NaN / inf cases propagate identically: ln of a negative A or t
// is NaN like pow / log of it).  lnpx2 = ln(px^2) is computed once on the host.
__device__ __forceinline__ void dt_twi_cell(int64_t fac, float srad, double lnpx2, double n, float &ti,
                                            float &mti, const DtLogEntry *s_tab) {
  if (fac <= -100) {
    ti = DT_NODATA;
    mti = DT_NODATA;
    return;
  }
  // float32 fast path (dt_math.h): valid arguments and results away from zero
  if (fac >= 0 && fac < (1ll << 40) && srad >= 0.0f && srad <= 1.2f) {
    double la = (fac == 0 ? 0.0 : dt_lnf((float)fac)) + lnpx2;
    double lt = dt_lntanf((double)srad + 0.01);
    double a = la - lt, b = n * la - lt;
    if (fabs(a) >= DT_FAST_MIN && fabs(b) >= DT_FAST_MIN) {
      ti = (float)a;
      mti = (float)b;
      return;
    }
  }
  double la = (fac == 0 ? 0.0 : dt_fast_log((double)fac, s_tab)) + lnpx2;
  double lt = dt_fast_lntan((double)srad + 0.01, s_tab);
  ti = (float)(la - lt);
  mti = (float)(n * la - lt);
}

// slope % -> radians as Example/example.py:63-64 does on the host: float32 quotient, arctan,
// float32 result (within ~1 float32 ulp of numpy's float32 arctan); -100 where dem == -100.
__device__ __forceinline__ float dt_slope_rad(float slope_pct, float dem) {
  if (dem == DT_NODATA) return DT_NODATA;
  float q = dt_pct_to_tan(slope_pct);
  if (q >= 0.0f && q < 1e30f) return (float)dt_atanf_pos(q);
  return (float)dt_fast_atan((double)q);
}
