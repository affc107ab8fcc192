// dt_kernels.hip -- hand-written HIP kernels for the descriptools hot path (gfx950 / CDNA4).
//
// All kernels are HBM-bound integer / float stencil, streaming or graph-propagation work: no MFMA.
// Reference citations are file:line relative to /root/reference/descriptools/.
#include <math.h>

#include "dt_common.h"
#include "dt_kernels.h"
#include "dt_math.h"

// ---- float64 math tables (dt_math.h) -------------------------------------------------------------
void dt_math_host_table(DtLogEntry *tab) {
  for (int i = 0; i < DT_LOGTAB_N; i++) {
    double c = 1.0 + (i + 0.5) / DT_LOGTAB_N;
    double rc = 1.0 / c;
    tab[i].rc = rc;
    tab[i].lnc = (double)(-logl((long double)rc));
  }
}
__device__ DtLogEntry g_logtab[DT_LOGTAB_N];
const DtLogEntry *dt_math_device_table(hipStream_t s) {
  static thread_local int uploaded_dev = -1;
  int dev = 0;
  (void)hipGetDevice(&dev);
  DtLogEntry *p = nullptr;
  (void)hipGetSymbolAddress((void **)&p, HIP_SYMBOL(g_logtab));
  if (uploaded_dev != dev) {
    static DtLogEntry host[DT_LOGTAB_N];
    dt_math_host_table(host);
    (void)hipMemcpyAsync(p, host, sizeof(host), hipMemcpyHostToDevice, s);
    (void)hipStreamSynchronize(s);
    uploaded_dev = dev;
  }
  return p;
}

// ===========================================================================================
// Synthetic DEM ("tilted integer fBm") -- integer arithmetic identical to
// oracle/dt_oracle.c so that CPU and GPU rasters are bit-identical.
// ===========================================================================================
#define DT_SYNTH_TILT 64
#define DT_SYNTH_KMIN 5
#define DT_SYNTH_SX 4
__constant__ int32_t c_synth_amp[15] = {0,    0,    0,    0,    0,     256,   445,  776,
                                        1351, 2352, 4096, 7131, 12416, 21618, 37640};

__device__ __forceinline__ uint32_t dt_hash32(uint32_t seed, uint32_t o, uint32_t ix, uint32_t iy) {
  uint32_t h = seed * 0x9E3779B1u ^ (o + 1u) * 0x85EBCA77u;
  h ^= ix * 0xC2B2AE3Du;
  h = ((h << 13) | (h >> 19)) * 0x27D4EB2Fu;
  h ^= iy * 0x165667B1u;
  h = ((h << 13) | (h >> 19)) * 0x9E3779B1u;
  h ^= h >> 15;
  h *= 0x85EBCA6Bu;
  h ^= h >> 13;
  h *= 0xC2B2AE35u;
  h ^= h >> 16;
  return h;
}

__global__ __launch_bounds__(256) void k_synth_dem(uint32_t seed, int O, int64_t Hg, int64_t y0,
                                                  int64_t x0, int64_t h, int64_t w, int nodata_pct,
                                                  float *__restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= h * w) return;
  int64_t r = i / w, c = i - r * w;
  int64_t y = y0 + r, x = x0 + c;
  int64_t z = (int64_t)DT_SYNTH_TILT * (Hg - 1 - y);
  for (int o = 0; o < O; o++) {
    int sh = O - o;
    if (sh < DT_SYNTH_KMIN) break;
    int shx = sh - DT_SYNTH_SX;
    uint32_t lx = (uint32_t)(x >> shx), ly = (uint32_t)(y >> sh);
    uint64_t fx = (uint64_t)(x & ((1 << shx) - 1)) << (16 - shx);
    uint64_t fy = (uint64_t)(y & ((1 << sh) - 1)) << (16 - sh);
    uint64_t v00 = dt_hash32(seed, (uint32_t)o, lx, ly) >> 16;
    uint64_t v10 = dt_hash32(seed, (uint32_t)o, lx + 1, ly) >> 16;
    uint64_t v01 = dt_hash32(seed, (uint32_t)o, lx, ly + 1) >> 16;
    uint64_t v11 = dt_hash32(seed, (uint32_t)o, lx + 1, ly + 1) >> 16;
    uint64_t top = v00 * (65536 - fx) + v10 * fx;
    uint64_t bot = v01 * (65536 - fx) + v11 * fx;
    uint64_t val = (top * (65536 - fy) + bot * fy) >> 32;
    z += (int64_t)((val * (uint64_t)c_synth_amp[sh]) >> 16);
  }
  float zf = (float)z * (1.0f / 256.0f);
  if (nodata_pct > 0) {
    uint32_t hb = dt_hash32(seed ^ 0xA5A5A5A5u, 77u, (uint32_t)(x >> 5), (uint32_t)(y >> 5));
    if ((hb % 100u) < (uint32_t)nodata_pct) zf = DT_NODATA;
  }
  out[i] = zf;
}

int dt_launch_synth_dem(hipStream_t s, uint32_t seed, int O, int64_t Hg, int64_t y0, int64_t x0,
                        int64_t h, int64_t w, int nodata_pct, float *out) {
  int64_t n = h * w;
  if (n == 0) return DT_OK;
  hipLaunchKernelGGL(k_synth_dem, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, seed, O, Hg,
                     y0, x0, h, w, nodata_pct, out);
  return DT_OK;
}


// ===========================================================================================
// Flow accumulation (N2), v1: in-degree countdown with ONE packed 64-bit word per cell,
//   state = remaining_in_degree << 56 | accumulated_upstream_cells
// A contribution is a single atomicAdd of (carry + 1) - (1 << 56): it adds the upstream count
// and retires one pending donor in the same RMW, so the thread that observes "I was the last
// donor" also observes the complete sum -- no fence / ordering protocol between two words.
// Integer adds: the result is independent of arrival order (bit-exact, deterministic).
// ===========================================================================================
#define FA_CNT_SHIFT 56
#define FA_ACC_MASK ((1ull << FA_CNT_SHIFT) - 1ull)
#define FA_CNT(s) (((s) >> FA_CNT_SHIFT) & 0xFull)
#define FA_SRC_BIT (1ull << 62) /* set at init on cells with in-degree 0; never modified */

__device__ __forceinline__ int64_t dt_step(int64_t pos, uint32_t code, int H, int W, bool &diag) {
  // returns target cell, -1 for a non-D8 code (incl. 0), -2 when leaving the raster
  if (!dt_d8_valid(code)) return -1;
  int dy, dx;
  dt_d8_delta(code, dy, dx);
  int y = (int)(pos / W), x = (int)(pos - (int64_t)y * W);
  y += dy;
  x += dx;
  diag = (dy != 0) && (dx != 0);
  if (y < 0 || y >= H || x < 0 || x >= W) return -2;
  return (int64_t)y * W + x;
}

__global__ __launch_bounds__(256) void k_fa_indeg(const uint8_t *__restrict__ fdr, int H, int W,
                                                 unsigned long long *__restrict__ state) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)H * W) return;
  int y = (int)(i / W), x = (int)(i - (int64_t)y * W);
  // neighbour (y+dy, x+dx) drains into me iff its code points back: (-dy, -dx)
  // code for delta (dy,dx): E(0,1)=1 SE(1,1)=2 S(1,0)=4 SW(1,-1)=8 W(0,-1)=16 NW(-1,-1)=32 N(-1,0)=64 NE(-1,1)=128
  unsigned cnt = 0;
#define DT_IN(dy, dx, code)                                                            \
  {                                                                                    \
    int yy = y + (dy), xx = x + (dx);                                                  \
    if (yy >= 0 && yy < H && xx >= 0 && xx < W && fdr[(size_t)yy * W + xx] == (code)) cnt++; \
  }
  DT_IN(0, -1, 1)     // W neighbour flowing E
  DT_IN(-1, -1, 2)    // NW neighbour flowing SE
  DT_IN(-1, 0, 4)     // N neighbour flowing S
  DT_IN(-1, 1, 8)     // NE neighbour flowing SW
  DT_IN(0, 1, 16)     // E neighbour flowing W
  DT_IN(1, 1, 32)     // SE neighbour flowing NW
  DT_IN(1, 0, 64)     // S neighbour flowing N
  DT_IN(1, -1, 128)   // SW neighbour flowing NE
#undef DT_IN
  state[i] = ((unsigned long long)cnt << FA_CNT_SHIFT) | (cnt == 0u ? FA_SRC_BIT : 0ull);
}

__global__ __launch_bounds__(256) void k_fa_walk(const uint8_t *__restrict__ fdr, int H, int W,
                                                unsigned long long *__restrict__ state) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)H * W) return;
  // original sources only (a non-source whose count later reaches 0 is continued by its last
  // donor, never restarted).  A source's word is never written, so a plain read is safe.
  if (!(state[i] & FA_SRC_BIT)) return;
  unsigned long long carry = 0ull;  // accumulation of the cell I am leaving
  int64_t cur = i;
  const int64_t limit = (int64_t)H * W;  // a walk visits each cell at most once: hard bound
  for (int64_t it = 0; it < limit; it++) {
    bool diag;
    int64_t t = dt_step(cur, fdr[cur], H, W, diag);
    if (t < 0) break;
    unsigned long long add = (carry + 1ull) - (1ull << FA_CNT_SHIFT);
    unsigned long long old = atomicAdd(&state[t], add);
    if (FA_CNT(old) != 1ull) break;  // other donors still pending: the last one carries on
    carry = (old & FA_ACC_MASK) + carry + 1ull;
    cur = t;
  }
}

__global__ __launch_bounds__(256) void k_fa_final(const unsigned long long *__restrict__ state,
                                                 const float *__restrict__ dem, int64_t n,
                                                 int32_t *__restrict__ acc32) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  unsigned long long s = state[i];
  int32_t v = (int32_t)(s & FA_ACC_MASK);
  if (FA_CNT(s) != 0ull) v = -100;  // on a D8 cycle: undefined
  if (dem && dem[i] <= DT_NODATA) v = -100;
  acc32[i] = v;
}

int dt_launch_flowacc(hipStream_t s, const uint8_t *fdr, const float *dem, int64_t H, int64_t W,
                      unsigned long long *state, int32_t *acc32) {
  int64_t n = H * W;
  if (n == 0) return DT_OK;
  dim3 g((unsigned)((n + 255) / 256)), b(256);
  hipLaunchKernelGGL(k_fa_indeg, g, b, 0, s, fdr, (int)H, (int)W, state);
  hipLaunchKernelGGL(k_fa_walk, g, b, 0, s, fdr, (int)H, (int)W, state);
  hipLaunchKernelGGL(k_fa_final, g, b, 0, s, state, dem, n, acc32);
  return DT_OK;
}

__global__ __launch_bounds__(256) void k_river_mask(const int32_t *__restrict__ acc32, int64_t n,
                                                   int32_t thr, int8_t *__restrict__ river) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) river[i] = acc32[i] > thr ? 1 : 0;  // example.py:52
}
int dt_launch_river_mask(hipStream_t s, const int32_t *acc32, int64_t n, int64_t thr, int8_t *river) {
  if (n == 0) return DT_OK;
  int32_t t32 = thr > 2147483647ll ? 2147483647 : (thr < -2147483648ll ? (int32_t)-2147483648ll : (int32_t)thr);
  hipLaunchKernelGGL(k_river_mask, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, acc32, n, t32, river);
  return DT_OK;
}

// ===========================================================================================
// Flow distance / river index / HAND (F3, F4), v1: pointer doubling on ONE packed 64-bit word
// per cell,
//   state = ptr:32 | n_diag:16 | done:1 | n_card:15
// "The path from this cell to cell `ptr` takes n_card cardinal and n_diag diagonal moves."
// Any historical value of a cell's word is a true statement, so rounds update in place with
// relaxed 64-bit loads/stores and no inter-workgroup protocol; 15 rounds resolve every path of
// <= 20000 moves (2^15 > 20000), longer paths / cycles exceed the cap and die, exactly the
// reference's `loop > 20000 -> -100` (flowhand.py:834-837).  The distance is materialised once,
// float32(px*n_card + px*sqrt(2)*n_diag) in float64: association-free, <= 1 float32 ulp from the
// reference's sequential float64 sum.
// ===========================================================================================
#define FH_DEAD 0xFFFFFFFFu
#define FH_DONE 0x8000u
#define FH_CAP 20000u

__device__ __forceinline__ unsigned long long fh_pack(uint32_t ptr, uint32_t nd, uint32_t nc_flags) {
  return ((unsigned long long)ptr << 32) | ((unsigned long long)nd << 16) | (unsigned long long)nc_flags;
}

__global__ __launch_bounds__(256) void k_fh_init(const uint8_t *__restrict__ fdr,
                                                const int8_t *__restrict__ river, int H, int W,
                                                unsigned long long *__restrict__ state) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)H * W) return;
  uint32_t code = fdr[i];
  unsigned long long s;
  if (code == 0u) {
    s = fh_pack(FH_DEAD, 0, FH_DONE);  // flowhand.py:601
  } else if (river[i] == 1) {
    s = fh_pack((uint32_t)i, 0, FH_DONE);  // flowhand.py:609-612
  } else {
    bool diag = false;
    int64_t t = dt_step(i, code, H, W, diag);
    // leaving the raster (:623-628 ...), a non-D8 code (caught by the revisit test :830) and
    // arrival on fdr == 0 (:826-828) all end in -100
    if (t < 0 || fdr[t] == 0u) s = fh_pack(FH_DEAD, 0, FH_DONE);
    else s = fh_pack((uint32_t)t, diag ? 1u : 0u, diag ? 0u : 1u);
  }
  state[i] = s;
}

__global__ __launch_bounds__(256) void k_fh_jump(unsigned long long *__restrict__ state, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  unsigned long long s = __hip_atomic_load(&state[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  uint32_t ncf = (uint32_t)(s & 0xFFFFu);
  if (ncf & FH_DONE) return;
  uint32_t ptr = (uint32_t)(s >> 32), nd = (uint32_t)((s >> 16) & 0xFFFFu), nc = ncf;
  unsigned long long t = __hip_atomic_load(&state[ptr], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  uint32_t tptr = (uint32_t)(t >> 32), tnd = (uint32_t)((t >> 16) & 0xFFFFu);
  uint32_t tncf = (uint32_t)(t & 0xFFFFu);
  uint32_t tnc = tncf & 0x7FFFu;
  bool tdone = (tncf & FH_DONE) != 0u;
  unsigned long long o;
  uint32_t nnc = nc + tnc, nnd = nd + tnd;
  if (tptr == FH_DEAD || nnc + nnd > FH_CAP) o = fh_pack(FH_DEAD, 0, FH_DONE);
  else o = fh_pack(tptr, nnd, nnc | (tdone ? FH_DONE : 0u));
  __hip_atomic_store(&state[i], o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(256) void k_fh_final(const unsigned long long *__restrict__ state,
                                                 const float *__restrict__ dem,
                                                 const int32_t *__restrict__ acc32, int64_t n,
                                                 double px, float *__restrict__ fdist,
                                                 int32_t *__restrict__ idx32, float *__restrict__ hand,
                                                 int32_t *__restrict__ a_river) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  unsigned long long s = state[i];
  uint32_t ptr = (uint32_t)(s >> 32), nd = (uint32_t)((s >> 16) & 0xFFFFu);
  uint32_t ncf = (uint32_t)(s & 0xFFFFu), nc = ncf & 0x7FFFu;
  bool ok = (ncf & FH_DONE) && ptr != FH_DEAD;  // not DONE after all rounds == longer than the cap
  if (fdist) fdist[i] = ok ? (float)(px * (double)nc + (px * sqrt(2.0)) * (double)nd) : DT_NODATA;
  if (idx32) idx32[i] = ok ? (int32_t)ptr : -100;
  if (hand) {
    float h = DT_NODATA;
    float z = dem[i];
    if (z != DT_NODATA && ok) {  // flowhand.py:436
      h = z - dem[ptr];
      if (h < 0.0f && h != DT_NODATA) h = 0.0f;  // flowhand.py:438
    }
    hand[i] = h;
  }
  if (a_river) a_river[i] = ok ? acc32[ptr] : -100;  // payload for GFI; -100 where hand is -100
}

int dt_launch_flowhand(hipStream_t s, const float *dem, const uint8_t *fdr, const int8_t *river,
                       const int32_t *acc32, int64_t H, int64_t W, double px,
                       unsigned long long *state, float *fdist, int32_t *idx32, float *hand,
                       int32_t *a_river) {
  int64_t n = H * W;
  if (n == 0) return DT_OK;
  dim3 g((unsigned)((n + 255) / 256)), b(256);
  hipLaunchKernelGGL(k_fh_init, g, b, 0, s, fdr, river, (int)H, (int)W, state);
  for (int r = 0; r < 15; r++) hipLaunchKernelGGL(k_fh_jump, g, b, 0, s, state, n);
  hipLaunchKernelGGL(k_fh_final, g, b, 0, s, state, dem, acc32, n, px, fdist, idx32, hand, a_river);
  return DT_OK;
}

// ===========================================================================================
// Pointwise descriptors: TI/MTI (T2/T3), GFI (G2), ln(hl/H) (G3).  float64 math, one float32
// rounding, as Numba types the reference kernels.
// ===========================================================================================
template <typename IT>
__global__ __launch_bounds__(256) void k_twi(const IT *__restrict__ acc32,
                                            const float *__restrict__ srad, int64_t n, double px2,
                                            double n_top, float *__restrict__ ti,
                                            float *__restrict__ mti, const DtLogEntry *__restrict__ g_tab) {
  __shared__ DtLogEntry s_tab[DT_LOGTAB_N];
  dt_math_stage(g_tab, s_tab);
  __syncthreads();
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float a, b;
  dt_twi_cell((int64_t)acc32[i], srad[i], px2, n_top, a, b, s_tab);
  ti[i] = a;
  mti[i] = b;
}
int dt_launch_twi(hipStream_t s, const int32_t *acc32, const float *srad, int64_t n, double px,
                  double n_top, float *ti, float *mti) {
  if (n == 0) return DT_OK;
  hipLaunchKernelGGL(k_twi<int32_t>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, acc32, srad, n,
                     log(px * px), n_top, ti, mti, dt_math_device_table(s));
  return DT_OK;
}
int dt_launch_twi_i64(hipStream_t s, const int64_t *fac, const float *srad, int64_t n, double px,
                      double n_top, float *ti, float *mti) {
  if (n == 0) return DT_OK;
  hipLaunchKernelGGL(k_twi<int64_t>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, fac, srad, n,
                     log(px * px), n_top, ti, mti, dt_math_device_table(s));
  return DT_OK;
}

// gfi.py:268-294 (own_cell = false: A = a_river, no zero guard) and gfi.py:404-440 (own_cell =
// true: A = fac, fac == 0 -> 1):  ln(b * (A size^2)^n / (h + 0.01)) = c0 + n ln A - ln(h + 0.01)
// with c0 = ln b + n ln(size^2) from the host; two logarithms instead of pow + log + division
// (same remark on rounding and NaN / inf propagation as dt_twi_cell: A == 0 gives -inf, A < 0 NaN).
template <bool OWN_CELL, typename IT>
__global__ __launch_bounds__(256) void k_gfi(const float *__restrict__ hand,
                                            const IT *__restrict__ area, int64_t n, double expo,
                                            double c0, float *__restrict__ out,
                                            const DtLogEntry *__restrict__ g_tab) {
  __shared__ DtLogEntry s_tab[DT_LOGTAB_N];
  dt_math_stage(g_tab, s_tab);
  __syncthreads();
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float h = hand[i];
  if (h <= DT_NODATA) {
    out[i] = DT_NODATA;
    return;
  }
  IT ar = area[i];
  // float64 table logarithms on the exactly converted integer area (dt_math.h): the index crosses zero inside
  // ordinary terrain, where a relative tolerance leaves no room for a float32 fast path
  double la = (OWN_CELL && ar == 0) ? 0.0 : dt_log_sel((double)ar, s_tab);
  out[i] = (float)(c0 + expo * la - dt_log_sel((double)h + 0.01, s_tab));
}
// 4 cells per thread with 16-byte loads / stores, grid-stride (the table is staged once per workgroup)
template <typename AccT>
__device__ __forceinline__ void gfi_load4(const AccT *p, AccT (&v)[4]);
template <>
__device__ __forceinline__ void gfi_load4<int32_t>(const int32_t *p, int32_t (&v)[4]) {
  const int4 q = *reinterpret_cast<const int4 *>(p);
  v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
}
template <>
__device__ __forceinline__ void gfi_load4<long long>(const long long *p, long long (&v)[4]) {
  const longlong2 a = *reinterpret_cast<const longlong2 *>(p), b = *reinterpret_cast<const longlong2 *>(p + 2);
  v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
}
template <typename AccT>
__global__ __launch_bounds__(256) void k_gfi_both(const float *__restrict__ hand,
                                                 const AccT *__restrict__ a_river,
                                                 const AccT *__restrict__ fac, int64_t n, double expo,
                                                 double c0, float *__restrict__ gfi,
                                                 float *__restrict__ lnhlh,
                                                 const DtLogEntry *__restrict__ g_tab, int vec_ok) {
  __shared__ DtLogEntry s_tab[DT_LOGTAB_N];
  dt_math_stage(g_tab, s_tab);
  __syncthreads();
  const int64_t stride = (int64_t)gridDim.x * 1024;
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
    if (vec_ok && i + 3 < n) {
      float4 h = *reinterpret_cast<const float4 *>(hand + i);
      AccT ar[4], f[4];
      gfi_load4<AccT>(a_river + i, ar);
      gfi_load4<AccT>(fac + i, f);
      float4 g, l;
      dt_gfi_both_cell(h.x, ar[0], f[0], expo, c0, s_tab, g.x, l.x);
      dt_gfi_both_cell(h.y, ar[1], f[1], expo, c0, s_tab, g.y, l.y);
      dt_gfi_both_cell(h.z, ar[2], f[2], expo, c0, s_tab, g.z, l.z);
      dt_gfi_both_cell(h.w, ar[3], f[3], expo, c0, s_tab, g.w, l.w);
      *reinterpret_cast<float4 *>(gfi + i) = g;
      *reinterpret_cast<float4 *>(lnhlh + i) = l;
    } else {
      for (int k = 0; k < 4 && i + k < n; k++) {
        float g, l;
        dt_gfi_both_cell(hand[i + k], a_river[i + k], fac[i + k], expo, c0, s_tab, g, l);
        gfi[i + k] = g;
        lnhlh[i + k] = l;
      }
    }
  }
}
// a_river / fac: int32_t* rasters, or int64_t* with acc64 != 0
int dt_launch_gfi_both(hipStream_t s, const float *hand, const void *a_river, const void *fac, int acc64,
                       int64_t n, double expo, double b, double size, float *gfi, float *lnhlh) {
  if (n == 0) return DT_OK;
  int vec_ok = (((uintptr_t)hand | (uintptr_t)a_river | (uintptr_t)fac | (uintptr_t)gfi | (uintptr_t)lnhlh) & 15) == 0;
  int64_t blocks = (n + 1023) / 1024;
  if (blocks > 256 * 16) blocks = 256 * 16;
  const double c0 = log(b) + expo * log(size * size);
  if (acc64)
    hipLaunchKernelGGL(k_gfi_both<long long>, dim3((unsigned)blocks), dim3(256), 0, s, hand, (const long long *)a_river,
                       (const long long *)fac, n, expo, c0, gfi, lnhlh, dt_math_device_table(s), vec_ok);
  else
    hipLaunchKernelGGL(k_gfi_both<int32_t>, dim3((unsigned)blocks), dim3(256), 0, s, hand, (const int32_t *)a_river,
                       (const int32_t *)fac, n, expo, c0, gfi, lnhlh, dt_math_device_table(s), vec_ok);
  return DT_OK;
}

int dt_launch_gfi(hipStream_t s, const float *hand, const int32_t *area, int64_t n, double expo,
                  double b, double size, float *out, int own_cell) {
  if (n == 0) return DT_OK;
  dim3 g((unsigned)((n + 255) / 256)), bl(256);
  if (own_cell) hipLaunchKernelGGL((k_gfi<true, int32_t>), g, bl, 0, s, hand, area, n, expo, log(b) + expo * log(size * size), out, dt_math_device_table(s));
  else hipLaunchKernelGGL((k_gfi<false, int32_t>), g, bl, 0, s, hand, area, n, expo, log(b) + expo * log(size * size), out, dt_math_device_table(s));
  return DT_OK;
}
int dt_launch_gfi_i64(hipStream_t s, const float *hand, const int64_t *area, int64_t n, double expo,
                      double b, double size, float *out, int own_cell) {
  if (n == 0) return DT_OK;
  dim3 g((unsigned)((n + 255) / 256)), bl(256);
  if (own_cell) hipLaunchKernelGGL((k_gfi<true, int64_t>), g, bl, 0, s, hand, area, n, expo, log(b) + expo * log(size * size), out, dt_math_device_table(s));
  else hipLaunchKernelGGL((k_gfi<false, int64_t>), g, bl, 0, s, hand, area, n, expo, log(b) + expo * log(size * size), out, dt_math_device_table(s));
  return DT_OK;
}
// gfi.river_accumulation (gfi.py:119-147): A_r = fac.flat[idx] where idx != -100 else fac.flat[0]
__global__ __launch_bounds__(256) void k_river_acc_i64(const int64_t *__restrict__ fac,
                                                      const int64_t *__restrict__ idx, int64_t n,
                                                      int64_t *__restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  int64_t k = idx[i];
  out[i] = (k != -100 && k >= 0 && k < n) ? fac[k] : fac[0];
}
int dt_launch_river_acc_i64(hipStream_t s, const int64_t *fac, const int64_t *idx, int64_t n, int64_t *out) {
  if (n) hipLaunchKernelGGL(k_river_acc_i64, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, fac, idx, n, out);
  return DT_OK;
}

// ===========================================================================================
// Downslope index (D1-D3), v1: one direct walk per cell restating kernel + repair
// (downslope.py:435-532 and :161-314); sequential float64 path-length sum as the reference.
// ===========================================================================================
__global__ __launch_bounds__(256) void k_downslope(const float *__restrict__ dem,
                                                  const uint8_t *__restrict__ fdr, int H, int W,
                                                  double px, double dz, int raw,
                                                  float *__restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)H * W) return;
  float z0 = dem[i];
  if (z0 <= DT_NODATA) {
    out[i] = DT_NODATA;
    return;
  }
  const double dcard = px, ddiag = px * sqrt(2.0);
  int64_t pos = i;
  double dist = 0.0;
  float drop = 0.0f;
  int loop = 0;
  bool failed = false;  // what downslope_gpu alone marks -50 (downslope.py:526-529)
  while ((double)drop < dz) {
    bool diag = false;
    int64_t t = dt_step(pos, fdr[pos], H, W, diag);
    if (t == -2) { failed = true; break; }  // raster-edge exit: stop (downslope.py:209-228)
    if (t >= 0) {
      float zt = dem[t];
      if (zt == DT_NODATA) { failed = true; break; }  // nodata ahead: stop without moving (:231-281)
      pos = t;
      dist += diag ? ddiag : dcard;
      drop = z0 - zt;
    }
    if (++loop == 5000) { failed = true; break; }  // :303-304 / :518-521
  }
  if (raw && failed) out[i] = -50.0f;
  else out[i] = dist == 0.0 ? 0.0f : (float)((double)drop / dist);
}
// Long walks.  On real, conditioned terrain (the bundled Example with its GIS D8 raster: 9.5 ms for 1.6 M valid cells,
// against 2.1 ms for the 268 M cells of the synthetic benchmark DEM) the cells of flats and valley floors walk
// thousands of moves before the elevation has dropped by dz, far outside the LDS window, one dependent global load per
// move -- and the kernel lasts as long as its longest walk.  With a workspace (dt_downslope_lift_bytes) such walks
// are queued as soon as they leave the window and finished by k_ds_finish with a SKIP TABLE: for every cell, where the
// walk stands 64 moves on (or where it cannot go on), how many of those moves are diagonal, and the lowest height on
// the way -- the 8-move table by walking, three rounds of pointer doubling from there, only when the queue holds at least DS_LIFT_MIN walks and one
// cell in 256 (fewer are walked out move by move by k_ds_finish: the table costs a pass over the raster).  A skip is
// taken when no cell of it can end the walk: the lowest height still leaves the drop below dz (the float32
// subtraction is monotone in the height, so the test on the minimum is exact), no move of it fails, and the 5000-move
// cap is not reached within it; skips of 8 moves (the table of the third round, kept) follow, and the moves that
// remain (< 8 + 8) are made one by one by the code above, so every
// exit of the reference's walk keeps its exact meaning.
#define DS_LIFT_MIN 256u /* and at least one cell in 256 (one in 128 while the entries had 16 bytes): the table costs ~170 bytes of traffic per CELL of the raster */
struct DsQueue {
  uint4 *entries;     // {start cell, cell the walk stands on (both y * W + x), moves made, diagonal moves}
  uint32_t *count;    // walks queued (may exceed capacity: the excess stayed in the main kernel)
  uint32_t capacity;
  __host__ __device__ DsQueue() : entries(nullptr), count(nullptr), capacity(0) {}
};
// Cells of the long-walk path are named by their index in the window's MEMORY (core + halo, row stride ld): for a
// single raster (no halo, ld = W) that is y * W + x
__device__ __forceinline__ uint32_t ds_mem_index(const DtWin &w, int y, int x) {
  return (uint32_t)((long long)(y + w.halo) * w.ld + (x + w.halo));
}
__device__ __forceinline__ void ds_mem_cell(const DtWin &w, uint32_t e, int &y, int &x) {
  const uint32_t r = e / (uint32_t)w.ld;
  y = (int)r - w.halo;
  x = (int)(e - r * (uint32_t)w.ld) - w.halo;
}
// skip-table entry, 8 bytes: y = float bits of the lowest height on the way (-inf for a NaN height); x = where the walk
// stands after the skip RELATIVE to its start (a skip is at most 64 moves: row and column offsets + 64 in bits 0-7 and
// 8-15), moves in bits 16-22, the diagonal ones in bits 23-29, bit 30 when the walk cannot go on from there.  (Round 3
// began with 16-byte entries holding the cell index: every kernel of this path is bound by the bytes of its table.)
#ifndef DS_Q_CAP
#define DS_Q_CAP 31 /* with the queue: a walk still running after DS_Q_CAP + 1 moves in the window is handed over */
#endif
#define DS_LIFT_STOP (1u << 30)
#define DS_LIFT_BIAS2 (64u | (64u << 8))
__device__ __forceinline__ uint32_t ds_lift_pack(int dy, int dx, uint32_t len, uint32_t nd) {
  return (uint32_t)(dy + 64) | ((uint32_t)(dx + 64) << 8) | (len << 16) | (nd << 23);
}
__device__ __forceinline__ int ds_lift_dy(uint32_t e) { return (int)(e & 0xFFu) - 64; }
__device__ __forceinline__ int ds_lift_dx(uint32_t e) { return (int)((e >> 8) & 0xFFu) - 64; }
__device__ __forceinline__ uint32_t ds_lift_len(uint32_t e) { return (e >> 16) & 0x7Fu; }
__device__ __forceinline__ uint32_t ds_lift_nd(uint32_t e) { return (e >> 23) & 0x7Fu; }

// ---- walkers: downslope walks that leave a rank's memory ------------------------------------------------------------
// A walk that reaches the end of its rank's memory (core + halo) belongs to another rank from there on.  Where it
// leaves, the kernel EMITS its state as a walker record (round 4; until then the cell was only marked -50 and
// descriptools_amd/tiling.finish_downslope started every such walk again at its start cell, move by move):
//   r0 = {start cell (global row, column), cell the walk stands on (global row, column)}
//   r1 = {moves made, diagonal moves, float bits of the start cell's height, flags}
//   r2 = {the reference's sequential float64 path length (DSW_SEQ walkers only), float bits of the result (DSW_DONE)}
// A walker normally only COUNTS its moves (the count form of the path length with its rounding-safety test gives the
// reference's float32 quotient, ds_quotient below); the ~1e-7 of the walks whose quotient is not provably
// order-independent start again as DSW_SEQ walkers, which carry the reference's own sequential sum from rank to rank.
// tiling.finish_downslope sends the records on as device buffers (all-to-all); k_ds_walk advances them.
#define DSW_SEQ 1u
#define DSW_DONE 2u
#define DSW_HOME 4u /* a finished walker whose value has been written at its start cell (k_ds_walk with `out`) */
#define DSW_WORDS 12 /* 32-bit words per record */
struct DsWalkOut {
  uint32_t *count;  // records emitted (may exceed capacity: the excess cells are only marked -50, as before)
  uint4 *rec;       // 3 x uint4 per record
  uint32_t capacity;
  __host__ __device__ DsWalkOut() : count(nullptr), rec(nullptr), capacity(0) {}
};
// called by the lanes whose walk leaves the rank (divergent code): ONE atomic per wave reserves their records
__device__ __forceinline__ void ds_emit_walker(const DsWalkOut &wo, const DtWin &w, int y0, int x0, int y, int x,
                                               uint32_t moves, uint32_t nd, float z0, uint32_t flags, double dist) {
  const unsigned long long m = __ballot(1);
  const uint32_t lane = __lane_id();
  const int leader = __ffsll((long long)m) - 1;
  uint32_t base = 0u;
  if ((int)lane == leader) base = atomicAdd(wo.count, (uint32_t)__popcll(m));
  base = (uint32_t)__shfl((int)base, leader);
  const uint32_t slot = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
  if (slot < wo.capacity) {
    uint4 *r = wo.rec + 3ull * slot;
    const unsigned long long db = (unsigned long long)__double_as_longlong(dist);
    r[0] = make_uint4((uint32_t)(y0 + w.gy0), (uint32_t)(x0 + w.gx0), (uint32_t)(y + w.gy0), (uint32_t)(x + w.gx0));
    r[1] = make_uint4(moves, nd, __float_as_uint(z0), flags);
    r[2] = make_uint4((uint32_t)db, (uint32_t)(db >> 32), 0u, 0u);
  }
}

// float32(drop / path length) from the COUNTS of the walk, or `safe` = false when the reference's sequential float64
// sum could round differently: see ds_finish_cell
__device__ __forceinline__ float ds_quotient(float drop, uint32_t loop, uint32_t nd, double dcard, double ddiag,
                                             bool &safe) {
  const double dist = dcard * (double)(loop - nd) + ddiag * (double)nd;
  double rc = __builtin_amdgcn_rcp(dist);
  rc = fma(fma(-dist, rc, 1.0), rc, rc);
  rc = fma(fma(-dist, rc, 1.0), rc, rc);
  const double q = (double)drop * rc;
  const uint64_t qb = (uint64_t)__double_as_longlong(q);
  const uint32_t qlo = (uint32_t)qb, qe = (uint32_t)(qb >> 52) & 0x7FFu;
  const uint32_t de = (uint32_t)((uint64_t)__double_as_longlong(dist) >> 52) & 0x7FFu;
  const uint32_t m = loop + 24u;
  const bool near_mid = ((qlo & 0x1FFFFFFFu) - (0x10000000u - m)) <= 2u * m;
  // A drop of exactly zero gives a zero quotient whatever the rounding of the path length.  (Without this exemption the
  // exponent test below sent every such walk to the reference's move-by-move form: on real terrain a quarter of the
  // long walks -- flats that end at nodata or at the raster's edge -- and 6.0 of k_ds_finish's 6.6 ms on the tiled
  // Example: profiles/r4/downslope_long_walks.txt.)
  safe = !(near_mid || ((qe - 897u) > 253u && drop != 0.0f) || (de - 523u) > 1000u);
  return (float)q;
}

// What one cell's fast walk in the LDS window hands over, turned into the stored value.  drop: z0 - z(cell the walk stands on) (+inf: it stepped onto nodata); loop / nd: moves
// made / diagonal ones; stop_fail: the walk stopped on a cell that cannot be left (non-D8 code, move off the raster);
// (y, x): rank coordinates of the cell it stands on.
// RANKED: the window is a rank's (core + halo inside a larger raster): the last ring of its memory has heights but no
// D8 codes (dt_has_code), and a walk that gets there belongs to another rank.  A single raster needs none of that.
template <bool RANKED = true>
__device__ __forceinline__ void ds_finish_cell(const DtWin &w, const float *__restrict__ dem,
                                               const uint8_t *__restrict__ fdr, int y0, int x0, float z0, float drop,
                                               uint32_t loop, uint32_t nd, bool stop_fail, int y, int x, double dcard,
                                               double ddiag, double dz, float dzf, int raw, float *__restrict__ outp,
                                               uint32_t &n_unres, const DsWalkOut &wo) {
  bool failed = false, slow = false, unresolved = false;
  uint32_t wflags = 0u;  // of the walker this walk becomes if it leaves the rank
  double wdist = 0.0;
  bool cont = false;  // continue on global memory from the cell the fast walk stopped on
  if (drop < dzf) {
    // stopped on a move word: a move off the raster stops the walk (downslope.py:209-228); a non-D8
    // code never moves again and the reference spins to its cap: same outcome, "failed" with the walk so
    // far; the ring and the 256-move limit continue below
    if (stop_fail) failed = true;
    else cont = true;
  } else if (drop == __builtin_inff()) {
    slow = true;  // stepped onto a nodata cell (staged as -inf): the reference stops one move earlier
  }
  if (cont) {
    // The few walks that reach the window ring (0.006 % of the cells of the 16384^2 DEM, but one in seven
    // windows has one) go on from where they are, still only counting moves: a handful of global loads
    // instead of the whole walk again.
    // the code of the cell the walk stands on is fetched together with the height of the cell before it: one memory
    // round trip per move on the walk's dependent chain, not two
    auto has_code = [&](int yy, int xx) { return RANKED ? dt_has_code(w, yy, xx) : dt_readable(w, yy, xx); };
    uint32_t code = has_code(y, x) ? (uint32_t)fdr[(long long)y * w.ld + x] : 0u;
    while ((double)drop < dz) {
      // at the end of this rank's memory (the last ring of the halo has heights, not codes): another rank's walk
      if (!has_code(y, x)) { unresolved = true; break; }
      if (!dt_d8_valid(code)) { failed = true; break; }
      int dy, dx;
      dt_d8_delta(code, dy, dx);
      int ny = y + dy, nx = x + dx;
      if (!dt_in_global(w, ny, nx)) { failed = true; break; }
      if (!dt_readable(w, ny, nx)) { unresolved = true; break; }
      const long long on = (long long)ny * w.ld + nx;
      float zt = dem[on];
      code = fdr[on];
      if (zt == DT_NODATA) { failed = true; break; }
      y = ny;
      x = nx;
      nd += (dy != 0 && dx != 0) ? 1u : 0u;
      drop = z0 - zt;
      if (++loop == 5000u) { failed = true; break; }  // downslope.py:303-304
    }
  }
  float res = 0.0f;  // no move: distance 0 -> 0 (downslope.py:306-309)
  if (!slow && !unresolved && loop != 0u) {
    // Count form of the path length, px*nc + px*sqrt(2)*nd, within (n + 8) * 2^-53 (n = moves) of the
    // reference's sequential float64 sum; the quotient by reciprocal + two Newton steps, within 2^-50 of the
    // reference's division.  The float32 result is accepted only if every value that close rounds to the same
    // float32: q's low 29 mantissa bits stay more than n + 24 float64 ulps away from the float32 rounding
    // midpoint 2^28 (and q is a normal float32, dist far from the ends of the double range).  One IEEE
    // float64 division, two products and three conversions per cell gave way to 4 FMAs and integer tests:
    // this kernel is bound by VALU issue and float64 runs at half rate.
    bool safe;
    res = ds_quotient(drop, loop, nd, dcard, ddiag, safe);
    // 2^-126 <= |q| < 2^128 (biased exponent 897..1150); 2^-500 <= |dist| < 2^501
    if (!safe) slow = true;
  }
  if (slow) {  // the reference's own walk with its sequential float64 sum, from the start, on global memory
    y = y0;
    x = x0;
    double dist = 0.0;
    drop = 0.0f;
    failed = false;
    unresolved = false;
    uint32_t moves = 0;
    nd = 0u;
    while ((double)drop < dz) {
      if (!(RANKED ? dt_has_code(w, y, x) : dt_readable(w, y, x))) { unresolved = true; break; }  // the end of this rank's memory
      uint32_t code = fdr[(long long)y * w.ld + x];
      if (!dt_d8_valid(code)) { failed = true; break; }
      int dy, dx;
      dt_d8_delta(code, dy, dx);
      int ny = y + dy, nx = x + dx;
      if (!dt_in_global(w, ny, nx)) { failed = true; break; }
      if (!dt_readable(w, ny, nx)) { unresolved = true; break; }
      float zt = dem[(long long)ny * w.ld + nx];
      if (zt == DT_NODATA) { failed = true; break; }
      y = ny;
      x = nx;
      dist += (dy != 0 && dx != 0) ? ddiag : dcard;
      drop = z0 - zt;
      if (++moves == 5000u) { failed = true; break; }
    }
    res = dist == 0.0 ? 0.0f : (float)((double)drop / dist);
    loop = moves;
    wflags = DSW_SEQ;
    wdist = dist;
  }
  if (unresolved) {  // the walk left the memory of this rank: it goes on as a walker (tiling.finish_downslope)
    *outp = -50.0f;
    n_unres++;  // counted by the caller, which adds up before it touches the device counter
    if (RANKED && wo.rec) ds_emit_walker(wo, w, y0, x0, y, x, loop, nd, z0, wflags, wdist);
  } else if (raw && failed) *outp = -50.0f;
  else *outp = res;
}

// the rest of a walk that found the queue full, from the cell it stands on
template <bool RANKED>
__device__ __attribute__((noinline)) void ds_finish_overflow(const DtWin &w, const float *__restrict__ dem,
                                                             const uint8_t *__restrict__ fdr, int y0, int x0, float z0,
                                                             uint32_t loop, uint32_t nd, uint32_t pos, double px,
                                                             double dz, float dzf, int raw, float *__restrict__ out,
                                                             int *__restrict__ n_unresolved, DsWalkOut wo) {
  int y, x;
  ds_mem_cell(w, pos, y, x);
  uint32_t unres = 0u;
  ds_finish_cell<RANKED>(w, dem, fdr, y0, x0, z0, z0 - dem[(long long)y * w.ld + x], loop, nd, false, y, x, px,
                        px * sqrt(2.0), dz, dzf, raw, out + (long long)y0 * w.ld + x0, unres, wo);
  if (unres && n_unresolved) atomicAdd(n_unresolved, (int)unres);
}

// Windowed version: a 1024-thread workgroup stages a 112 x 112 window (64 x 64 core + 24-cell
// margin) in LDS as float32 heights plus one pre-decoded 16-bit "move word" per cell
//   bits 0-9  BYTE offset of the D8 successor's move word from this cell's, biased by 512
//   bit  10   the move is diagonal
//   bits 11-13 why the fast walk must stop here: non-D8 code / the move leaves the raster / the cell is
//             on the window's outer ring (successor may be outside the window)
//   bit  15   any of those
// so a move is 9 VALU instructions and one LDS round trip (the global walk is VALU-bound on D8 decoding
// and bounds tests, not latency-bound).  The fast walk only COUNTS cardinal and diagonal moves; the
// reference's path length is a sequential float64 sum, whose rounding depends on the order of the moves,
// so the count form px*nc + px*sqrt(2)*nd is accepted only when the float32 result provably does not
// depend on that (the whole error interval rounds to one float); the rare other cells, and walks that
// reach the window ring, are redone by the generic sequential walk on global memory.  Results are
// bit-identical to k_downslope.  75 KiB of LDS: two workgroups per CU; workgroups are banded per XCD so
// overlapping margins come from L2.
#define DW_CORE 64
// margin DW_M (template parameter: 24 by default; DT_DBG_DS_MARGIN tries 16 / 20), window DW_WIN = core + 2 margins
// LDS row stride in cells.  Lanes that merged onto one flow path trail each other by a few cells; with a
// stride of 112 dwords (= 16 mod 32 banks) two cells 2 rows apart in one column share a bank, the common
// case on south-flowing terrain.  116 = 20 mod 32: same column conflicts only 8 rows apart, the diagonals
// 32 apart; rows stay 16-byte aligned for the float4 staging stores.
// DW_LD = DW_WIN + 4 (116 for the 112-cell window)
#define MW_OFF 0x3FFu
#define MW_BIAS 512
#define MW_DIAG 0x400u
#define MW_BADCODE 0x800u
#define MW_EDGE 0x1000u
#define MW_RING 0x2000u
#define MW_STOP 0x8000u

template <int DW_M, bool QUEUE, bool RANKED>
__device__ __forceinline__ void ds_win_body(const float *__restrict__ dem,
                                                       const uint8_t *__restrict__ fdr, DtWin w,
                                                       double px, double dz, float dzf, int raw,
                                                       float *__restrict__ out, int tiles_x, int ntiles,
                                                       int *__restrict__ n_unresolved, DsQueue queue, DsWalkOut wo) {
  constexpr int DW_WIN = DW_CORE + 2 * DW_M, DW_LD = DW_WIN + 4;
  // one LDS block: heights at byte 0, move words at byte DW_LD*DW_WIN*4 (the walk reads both from one
  // address register)
  __shared__ __attribute__((aligned(16))) unsigned char smem[DW_LD * DW_WIN * 6];
  __shared__ uint16_t s_lut[256];
  // QUEUE: walks this workgroup hands over, and where its block of the queue starts; RANKED: [2] = its unresolved walks
  __shared__ uint32_t s_queued[3];
  float *s_z = reinterpret_cast<float *>(smem);
  uint16_t *s_w = reinterpret_cast<uint16_t *>(smem + DW_LD * DW_WIN * 4);
  // LDS byte address of smem (0 when it is the kernel's only LDS object, but do not rely on it)
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;
  int b = blockIdx.x, tile;
  {
    int xcd = b & 7, j = b >> 3;
    int q = ntiles >> 3, rem = ntiles & 7;
    // When the band is whole tile rows, walk it in strips of 8 tile columns, top to bottom: the 24 + 24 margin rows
    // a window shares with the windows above and below are then re-read from L2 a few workgroups later, not from
    // HBM a whole tile row later (PMC: 8.75 B/cell fetched for 5 of dem + fdr = exactly the 1.75 x of the vertical
    // overlap)
    if (rem == 0 && q % tiles_x == 0 && (tiles_x & 7) == 0) {
      const int rows = q / tiles_x, per_strip = rows * 8;
      const int strip = j / per_strip, r = (j - strip * per_strip) >> 3, c = (j & 7) + strip * 8;
      tile = xcd * q + r * tiles_x + c;
    } else {
      tile = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + j;
    }
  }
  const int tyi = tile / tiles_x, txi = tile - tyi * tiles_x;
  const int wy0 = tyi * DW_CORE - DW_M, wx0 = txi * DW_CORE - DW_M;
  const bool vec = (w.ld % 4 == 0) && (((uintptr_t)dem & 15) == 0) && (((uintptr_t)fdr & 3) == 0);
  // readable cells (inside the global raster and in memory): rows [ya, yb), columns [xa, xb)
  const int ya = max(-w.halo, -w.gy0), yb = min(w.H + w.halo, w.Hg - w.gy0);
  const int xa = max(-w.halo, -w.gx0), xb = min(w.W + w.halo, w.Wg - w.gx0);
  // cells whose D8 code is in memory (dt_has_code): the last ring of the halo is not, unless it is the raster's edge
  const int yca = (RANKED && -w.halo > -w.gy0) ? ya + 1 : ya, ycb = (RANKED && w.H + w.halo < w.Hg - w.gy0) ? yb - 1 : yb;
  const int xca = (RANKED && -w.halo > -w.gx0) ? xa + 1 : xa, xcb = (RANKED && w.W + w.halo < w.Wg - w.gx0) ? xb - 1 : xb;
  // block-uniform: every window cell is in memory with its code and no move from it can leave the global raster
  const bool interior = vec && wy0 >= yca && wy0 + DW_WIN <= ycb && wx0 >= xca && wx0 + DW_WIN <= xcb &&
                        w.gy0 + wy0 >= 1 && w.gy0 + wy0 + DW_WIN <= w.Hg - 1 && w.gx0 + wx0 >= 1 &&
                        w.gx0 + wx0 + DW_WIN <= w.Wg - 1;
  const float ninf = -__builtin_inff();
  // 112 rows x 28 groups of 4 cells (wx0 is a multiple of 8: float4 / uchar4 stay aligned).  A nodata
  // height is staged as -inf: the walk never moves onto nodata (downslope.py:231-281); a lane that does
  // sees an infinite drop, stops, and is redone by the generic walk.
  constexpr int NG = DW_WIN * (DW_WIN / 4);  // 3136 groups of 4 cells: up to 4 per thread
  // all of a thread's loads first (one memory round trip per workgroup instead of three), then the table
  // and its barrier (which so overlap the loads), then decode
  float4 vv[4];
  uint32_t cc[4];
  int nod_here = 0;  // this thread staged a nodata height
  if (interior) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
      int i = threadIdx.x + 1024 * u;
      if (i < NG) {
        int r = i / (DW_WIN / 4), c4 = (i - r * (DW_WIN / 4)) * 4;
        long long g = (long long)(wy0 + r) * w.ld + wx0 + c4;
        vv[u] = *reinterpret_cast<const float4 *>(dem + g);
        cc[u] = *reinterpret_cast<const uint32_t *>(fdr + g);
      }
    }
  }
  // D8 code -> move word (256 entries: one LDS read per cell instead of ~15 VALU instructions)
  if (threadIdx.x < 256) {
    uint32_t code = threadIdx.x, mw = MW_BADCODE | MW_STOP | MW_BIAS;
    if (dt_d8_valid(code)) {
      int dy, dx;
      dt_d8_delta(code, dy, dx);
      mw = (uint32_t)(2 * (dy * DW_LD + dx) + MW_BIAS);
      if (dy != 0 && dx != 0) mw |= MW_DIAG;
    }
    s_lut[code] = (uint16_t)mw;
  }
  if ((QUEUE || RANKED) && threadIdx.x == 0) s_queued[0] = s_queued[2] = 0u;
  __syncthreads();
  if (interior) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
      int i = threadIdx.x + 1024 * u;
      if (i < NG) {
        int r = i / (DW_WIN / 4), c4 = (i - r * (DW_WIN / 4)) * 4;
        const uint32_t rowring = (r == 0 || r == DW_WIN - 1) ? (MW_RING | MW_STOP) : 0u;
        uint32_t m0 = (uint32_t)s_lut[cc[u] & 0xFFu] | rowring | (c4 == 0 ? (MW_RING | MW_STOP) : 0u);
        uint32_t m1 = (uint32_t)s_lut[(cc[u] >> 8) & 0xFFu] | rowring;
        uint32_t m2 = (uint32_t)s_lut[(cc[u] >> 16) & 0xFFu] | rowring;
        uint32_t m3 = (uint32_t)s_lut[cc[u] >> 24] | rowring | (c4 == DW_WIN - 4 ? (MW_RING | MW_STOP) : 0u);
        float4 v = vv[u];
        nod_here |= (v.x == DT_NODATA || v.y == DT_NODATA || v.z == DT_NODATA || v.w == DT_NODATA) ? 1 : 0;
        v.x = v.x == DT_NODATA ? ninf : v.x;
        v.y = v.y == DT_NODATA ? ninf : v.y;
        v.z = v.z == DT_NODATA ? ninf : v.z;
        v.w = v.w == DT_NODATA ? ninf : v.w;
        *reinterpret_cast<float4 *>(&s_z[r * DW_LD + c4]) = v;
        *reinterpret_cast<uint2 *>(&s_w[r * DW_LD + c4]) = make_uint2(m0 | (m1 << 16), m2 | (m3 << 16));
      }
    }
  } else {
    for (int i = threadIdx.x; i < NG; i += 1024) {
      int r = i / (DW_WIN / 4), c4 = (i - r * (DW_WIN / 4)) * 4;
      int gy = wy0 + r, gx = wx0 + c4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      uint32_t mwv[4];
      uint32_t codes = 0;
      const bool row_ok = gy >= ya && gy < yb;
      bool rd[4];
#pragma unroll
      for (int k = 0; k < 4; k++) rd[k] = row_ok && gx + k >= xa && gx + k < xb;
      if (vec && rd[0] && rd[3]) {
        v = *reinterpret_cast<const float4 *>(dem + (long long)gy * w.ld + gx);
        codes = *reinterpret_cast<const uint32_t *>(fdr + (long long)gy * w.ld + gx);
      } else {
        const float *p = dem + (long long)gy * w.ld;
        const uint8_t *f = fdr + (long long)gy * w.ld;
        if (rd[0]) { v.x = p[gx]; codes |= (uint32_t)f[gx]; }
        if (rd[1]) { v.y = p[gx + 1]; codes |= (uint32_t)f[gx + 1] << 8; }
        if (rd[2]) { v.z = p[gx + 2]; codes |= (uint32_t)f[gx + 2] << 16; }
        if (rd[3]) { v.w = p[gx + 3]; codes |= (uint32_t)f[gx + 3] << 24; }
      }
#pragma unroll
      for (int k = 0; k < 4; k++) {
        uint32_t code = (codes >> (8 * k)) & 0xFFu;
        int rx = c4 + k;
        uint32_t mw = s_lut[code];
        if (!(mw & MW_STOP)) {
          int dy, dx;
          dt_d8_delta(code, dy, dx);
          int ny = w.gy0 + gy + dy, nx = w.gx0 + gx + k + dx;
          if (ny < 0 || ny >= w.Hg || nx < 0 || nx >= w.Wg) mw |= MW_EDGE | MW_STOP;
          // successor inside the raster but not in this rank's memory (halo narrower than the margin)
          else if (gy + dy < ya || gy + dy >= yb || gx + k + dx < xa || gx + k + dx >= xb) mw |= MW_RING | MW_STOP;
        }
        // ring of the window, or the edge of what is in memory: hand over to the global walk
        if (r == 0 || r == DW_WIN - 1 || rx == 0 || rx == DW_WIN - 1 || !rd[k]) mw |= MW_RING | MW_STOP;
        // a cell of the halo's last ring has a height but no code (0 in memory is not "a non-D8 code" there): the
        // walk may step onto it, and goes on in another rank's memory
        if (RANKED && rd[k] && !(gy >= yca && gy < ycb && gx + k >= xca && gx + k < xcb)) mw = MW_RING | MW_STOP | MW_BIAS;
        mwv[k] = mw;
      }
      nod_here |= (v.x == DT_NODATA || v.y == DT_NODATA || v.z == DT_NODATA || v.w == DT_NODATA) ? 1 : 0;
      v.x = v.x == DT_NODATA ? ninf : v.x;
      v.y = v.y == DT_NODATA ? ninf : v.y;
      v.z = v.z == DT_NODATA ? ninf : v.z;
      v.w = v.w == DT_NODATA ? ninf : v.w;
      *reinterpret_cast<float4 *>(&s_z[r * DW_LD + c4]) = v;
      *reinterpret_cast<uint2 *>(&s_w[r * DW_LD + c4]) =
          make_uint2(mwv[0] | (mwv[1] << 16), mwv[2] | (mwv[3] << 16));
    }
  }
  // NODATA AHEAD (round 4).  The walk does not look before it moves: a lane that steps onto nodata (staged as -inf)
  // sees an infinite drop, and its cell was then walked again, move by move on global memory, to find where the
  // reference had stopped -- one move earlier (downslope.py:231-281).  Rare on the synthetic terrain; but the D8 codes
  // of a conditioned DEM drain every flat next to nodata INTO it, and a GIS raster its whole basin: 0.33 % of the
  // cells of the conditioned rough raster, one wave in five waiting ~10 us for such a walk.  A window that holds
  // nodata at all (block-uniform) now marks the cells whose successor is nodata as "the move leaves the raster" --
  // the same outcome in the reference: the walk so far, failed -- and the walk stops there by itself.  (A height of
  // -inf is staged like nodata: the candidate's successor is looked up in the raster before it is marked.)  Only in the
  // kernels with the queue -- what real and conditioned terrain run: the pass costs the synthetic benchmark raster,
  // whose nodata blobs sit on hill tops where no walk ends, 0.09 ms of the plain kernel's 2.04 and gains it nothing.
  if (QUEUE && __syncthreads_or(nod_here)) {
    for (int i = threadIdx.x; i < DW_WIN * DW_WIN; i += 1024) {
      const int r = i / DW_WIN, c = i - r * DW_WIN;
      const int p = r * DW_LD + c;
      const uint32_t mwc = s_w[p];
      if (mwc & MW_STOP) continue;
      const int sp = p + (((int)(mwc & MW_OFF) - MW_BIAS) >> 1);  // (the offset field counts bytes of move words)
      if (s_z[sp] != ninf) continue;
      const int sr = sp / DW_LD, sc = sp - sr * DW_LD;
      if (dem[(long long)(wy0 + sr) * w.ld + wx0 + sc] == DT_NODATA) s_w[p] = (uint16_t)(mwc | MW_EDGE | MW_STOP);
    }
  }
  __syncthreads();
  const double dcard = px, ddiag = px * sqrt(2.0);
  const uint32_t neg2lds0 = 0u - 2u * lds0 + lds0;  // z address = 2 * (q2 - lds0) + lds0 = (q2 << 1) + neg2lds0
  // QUEUE: a thread's walks to hand over (bit j of pend: its j-th cell), where each stands and moves | diagonal ones << 16
  uint32_t unres = 0u;  // walks that left this rank's memory
  uint32_t pend = 0u, pq0 = 0u, pq1 = 0u, pq2 = 0u, pq3 = 0u, pl0 = 0u, pl1 = 0u, pl2 = 0u, pl3 = 0u;
  static_assert((DW_CORE * DW_CORE) / 1024 == 4, "four cells per thread");
  for (int j = 0; j < (DW_CORE * DW_CORE) / 1024; j++) {
    int c = threadIdx.x + 1024 * j;
    int cy = c / DW_CORE, cx = c - cy * DW_CORE;
    int y0 = tyi * DW_CORE + cy, x0 = txi * DW_CORE + cx;
    if (y0 >= w.H || x0 >= w.W) continue;
    const int pos0 = (cy + DW_M) * DW_LD + cx + DW_M;
    float z0 = s_z[pos0];
    long long o = (long long)y0 * w.ld + x0;
    if (z0 <= DT_NODATA) {
      out[o] = DT_NODATA;
      continue;
    }
    float drop = 0.0f;
    // Fast walk inside the window.  (double)drop < dz  <=>  drop < dzf (dzf = smallest float >= dz).
    // q2 = LDS byte address of the current cell's move word minus the array's offset; the next cell's
    // height and move word are fetched together.  The loop is written out: the compiler's version spends
    // as many scalar instructions on exec-mask bookkeeping as vector instructions on the walk, and the
    // kernel is bound by instruction issue.  Here the two stop tests are v_cmpx (they clear the lanes
    // that stop straight in EXEC) and the bookkeeping is ONE add: acc += move word + 2^19, i.e. the number
    // of moves in bits 19.. and the sum of the move words below (offsets, biased, + 0x400 per diagonal
    // move; < 2^19 for up to 256 moves).  The offsets sum to the distance walked in LDS, so the
    // diagonal count falls out afterwards.  7 VALU + 2 LDS instructions per move.  A walk still running
    // after 256 moves inside the window (a spiral or a D8 cycle) is left to the generic walk, which also
    // owns the 5000-move cap of downslope.py:303-304.
    const uint32_t q2_0 = lds0 + 2u * (uint32_t)pos0;
    uint32_t q2 = q2_0, mw = s_w[pos0], acc = 0;
    {
      uint32_t t, cnt;
      float zt;
      uint64_t sv;
      asm volatile(
          "s_mov_b64 %[sv], exec\n\t"
          "s_movk_i32 %[cnt], %[cap]\n\t"
          "v_cmpx_gt_f32 vcc, %[dzf], %[drop]\n\t"
          "v_cmpx_gt_u32 vcc, 0x8000, %[mw]\n\t"
          "s_cbranch_execz 2f\n"
          "1:\n\t"
          "v_and_b32 %[t], 0x3ff, %[mw]\n\t"
          "v_add3_u32 %[acc], %[acc], %[mw], %[k19]\n\t"
          "v_add3_u32 %[q2], %[q2], %[t], %[nbias]\n\t"
          "v_lshl_add_u32 %[t], %[q2], 1, %[n2l]\n\t"
          "ds_read_b32 %[zt], %[t]\n\t"
          "ds_read_u16 %[mw], %[q2] offset:%[woff]\n\t"
          "s_sub_u32 %[cnt], %[cnt], 1\n\t"
          "s_waitcnt lgkmcnt(0)\n\t"
          "v_sub_f32 %[drop], %[z0], %[zt]\n\t"
          "s_cbranch_scc1 2f\n\t"  // borrow: that was the 256th move, hand over
          "v_cmpx_gt_f32 vcc, %[dzf], %[drop]\n\t"
          "v_cmpx_gt_u32 vcc, 0x8000, %[mw]\n\t"
          "s_cbranch_execnz 1b\n"
          "2:\n\t"
          "s_mov_b64 exec, %[sv]"
          : [q2] "+v"(q2), [mw] "+v"(mw), [acc] "+v"(acc), [drop] "+v"(drop), [t] "=&v"(t), [zt] "=&v"(zt),
            [sv] "=&s"(sv), [cnt] "=&s"(cnt)
          : [z0] "v"(z0), [dzf] "s"(dzf), [nbias] "s"(0u - (uint32_t)MW_BIAS), [n2l] "s"(neg2lds0),
            [k19] "s"(1u << 19), [woff] "n"(DW_LD * DW_WIN * 4),
            // with the queue a walk still running after 32 moves is a long one: hand it over at once (the lanes that
            // are done wait for the wave's longest walk)
            [cap] "n"(QUEUE ? DS_Q_CAP : 255)
          : "vcc", "scc", "memory");
    }
    uint32_t loop = acc >> 19;  // moves made
    // sum of the move words = (q2 - q2_0) + MW_BIAS * moves + MW_DIAG * diagonal moves
    uint32_t nd = ((acc & 0x7FFFFu) - (q2 - q2_0) - (uint32_t)MW_BIAS * loop) / MW_DIAG;
    const uint32_t pos = (q2 - lds0) >> 1;
    const int ys = wy0 + (int)(pos / DW_LD), xs = wx0 + (int)(pos % DW_LD);
    const bool stop_fail = (mw & (MW_BADCODE | MW_EDGE)) != 0u;
    if (QUEUE && drop < dzf && !stop_fail && (RANKED ? dt_has_code(w, ys, xs) : dt_readable(w, ys, xs))) {
      // A LONG walk (a flat, a valley floor: real conditioned terrain has walks of thousands of moves, each a
      // dependent global load in ds_finish_cell): left for k_ds_finish, which crosses it in skips of 64 moves
      const uint32_t pq = ds_mem_index(w, ys, xs), pl = loop | (nd << 16);
      if (j == 0) { pq0 = pq; pl0 = pl; }
      else if (j == 1) { pq1 = pq; pl1 = pl; }
      else if (j == 2) { pq2 = pq; pl2 = pl; }
      else { pq3 = pq; pl3 = pl; }
      pend |= 1u << j;
      continue;
    }
    ds_finish_cell<RANKED>(w, dem, fdr, y0, x0, z0, drop, loop, nd, stop_fail, ys, xs, dcard, ddiag, dz, dzf, raw,
                           out + o, unres, wo);
  }
  if (RANKED) {
    // on real terrain every border of a rank has such walks by the ten thousand: counted per workgroup (LDS), ONE
    // device atomic each -- one per cell queues up on the one address like the queue's reservations did
    if (unres) atomicAdd(&s_queued[2], unres);
    __syncthreads();
    if (threadIdx.x == 0 && s_queued[2] && n_unresolved) atomicAdd(n_unresolved, (int)s_queued[2]);
  } else if (unres && n_unresolved) {
    atomicAdd(n_unresolved, (int)unres);  // (a single raster has no memory to run out of)
  }
  if (QUEUE) {
    // ONE global atomic per workgroup reserves its block of the queue (one per wave and pass -- 3 M of them on one
    // address for a 214 M-cell raster -- took 4.3 of the kernel's 6.7 ms); inside the block the threads order
    // themselves with an LDS atomic
    const uint32_t mine = (uint32_t)__popc(pend);
    uint32_t slot = 0u;
    if (mine) slot = atomicAdd(&s_queued[0], mine);
    __syncthreads();
    if (threadIdx.x == 0 && s_queued[0]) s_queued[1] = atomicAdd(queue.count, s_queued[0]);
    __syncthreads();
    slot += s_queued[1];
    for (int j = 0; pend >> j; j++) {
      if (!((pend >> j) & 1u)) continue;
      const uint32_t pq = j == 0 ? pq0 : j == 1 ? pq1 : j == 2 ? pq2 : pq3;
      const uint32_t pl = j == 0 ? pl0 : j == 1 ? pl1 : j == 2 ? pl2 : pl3;
      const int c = threadIdx.x + 1024 * j;
      const int cy = c / DW_CORE, cx = c - cy * DW_CORE;
      const int y0 = tyi * DW_CORE + cy, x0 = txi * DW_CORE + cx;
      if (slot < queue.capacity) {
        queue.entries[slot] = make_uint4(ds_mem_index(w, y0, x0), pq, pl & 0xFFFFu, pl >> 16);
      } else {
        // a full queue (more than half the raster's cells): the walk is made here after all
        const float z0 = s_z[(cy + DW_M) * DW_LD + cx + DW_M];
        ds_finish_overflow<RANKED>(w, dem, fdr, y0, x0, z0, pl & 0xFFFFu, pl >> 16, pq, px, dz, dzf, raw, out,
                                   n_unresolved, wo);
      }
      slot++;
    }
  }
}

template <int DW_M>
__global__ __launch_bounds__(1024, 8) void k_downslope_win(const float *__restrict__ dem,
                                                       const uint8_t *__restrict__ fdr, DtWin w, double px, double dz,
                                                       float dzf, int raw, float *__restrict__ out, int tiles_x,
                                                       int ntiles, int *__restrict__ n_unresolved) {
  ds_win_body<DW_M, false, false>(dem, fdr, w, px, dz, dzf, raw, out, tiles_x, ntiles, n_unresolved, DsQueue(),
                                  DsWalkOut());
}
// a rank's window (core + halo inside a larger raster): see ds_finish_cell
__global__ __launch_bounds__(1024, 8) void k_downslope_win_r(const float *__restrict__ dem,
                                                         const uint8_t *__restrict__ fdr, DtWin w, double px,
                                                         double dz, float dzf, int raw, float *__restrict__ out,
                                                         int tiles_x, int ntiles, int *__restrict__ n_unresolved,
                                                         DsWalkOut wo) {
  ds_win_body<24, false, true>(dem, fdr, w, px, dz, dzf, raw, out, tiles_x, ntiles, n_unresolved, DsQueue(), wo);
}
// the same with the hand-over of long walks to the queue (a kernel of its own: the plain one keeps its registers)
__global__ __launch_bounds__(1024, 8) void k_downslope_win_q(const float *__restrict__ dem,
                                                         const uint8_t *__restrict__ fdr, DtWin w, double px,
                                                         double dz, float dzf, int raw, float *__restrict__ out,
                                                         int tiles_x, int ntiles, int *__restrict__ n_unresolved,
                                                         DsQueue queue) {
  ds_win_body<24, true, false>(dem, fdr, w, px, dz, dzf, raw, out, tiles_x, ntiles, n_unresolved, queue, DsWalkOut());
}

// ... of a rank's window: the long walks that stay in the rank's memory; the ones that leave it are counted as ever
__global__ __launch_bounds__(1024, 8) void k_downslope_win_rq(const float *__restrict__ dem,
                                                          const uint8_t *__restrict__ fdr, DtWin w, double px,
                                                          double dz, float dzf, int raw, float *__restrict__ out,
                                                          int tiles_x, int ntiles, int *__restrict__ n_unresolved,
                                                          DsQueue queue, DsWalkOut wo) {
  ds_win_body<24, true, true>(dem, fdr, w, px, dz, dzf, raw, out, tiles_x, ntiles, n_unresolved, queue, wo);
}

// ---- skip table ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float ds_lift_z(float z) { return z != z ? -__builtin_inff() : z; }  // a NaN height ends a walk
// the 8-move table straight from the rasters, for every cell of the window's memory (core + halo; a single raster: the
// raster) instead of three rounds of doubling over one-move entries.  A skip ends (stop flag) where the move-by-move
// code would have to decide something: a cell without a code in memory (non-D8, or the last ring of a rank's halo), a
// move off the raster or out of the rank's memory, nodata ahead.
// One workgroup per 64 x 64 tile, the tile and 8 cells around it staged in LDS (round 4; a lane walked its cell's
// eight moves on global memory before -- 16 dependent loads that mostly hit the caches, and 2.0 ms for 214 M cells) as
// one 8-byte entry per cell: the height (a NaN as -inf, what is not in memory as nodata) and a MOVE WORD
//   bits 0-15  byte offset of the D8 successor's entry from this one (signed)
//   bits 16-19 1 (a move)        bits 24-27 1 when the move is diagonal
//   0          no move from here: no code / non-D8 code / successor not in memory / successor nodata / window ring
// so a move is one 64-bit LDS read and four vector instructions without a branch -- a walk that has stopped adds
// zeros and re-reads the entry it stands on.  Each lane walks 16 cells, four at a time.  (The first LDS version spent
// 220 instructions per cell, most of them on bounds tests and D8 decoding in the staging: 1.45 ms, issue-bound.  The
// move words now come from a 256-entry table by code, and a tile whose window lies inside the memory stages without
// tests.)
#define LT_M 8
#define LT_WIN (64 + 2 * LT_M)
#define LT_LD (LT_WIN + 1)
template <bool RANKED>
__global__ __launch_bounds__(256) void k_ds_lift_init(const float *__restrict__ dem, const uint8_t *__restrict__ fdr,
                                                     DtWin w, uint2 *__restrict__ T, const uint32_t *__restrict__ qcount,
                                                     uint32_t lift_min, int tiles_x) {
  if (*qcount < lift_min) return;
  __shared__ uint2 s_c[LT_WIN * LT_LD];
  __shared__ uint32_t s_tab[256];
  {  // move word by D8 code
    const uint32_t code = threadIdx.x;
    uint32_t m = 0u;
    if (dt_d8_valid(code)) {
      int dy, dx;
      dt_d8_delta(code, dy, dx);
      m = ((uint32_t)((dy * LT_LD + dx) * 8) & 0xFFFFu) | (1u << 16) | ((dy != 0 && dx != 0) ? (1u << 24) : 0u);
    }
    s_tab[code] = m;
  }
  const int ty = (int)blockIdx.x / tiles_x, tx = (int)blockIdx.x - ty * tiles_x;
  const int y0 = ty * 64 - w.halo, x0 = tx * 64 - w.halo;  // the tile's first cell (rank coordinates)
  const int wy0 = y0 - LT_M, wx0 = x0 - LT_M;
  // (LT_WIN^2 = 25 * 256: every lane stages 25 cells, all loads in flight before the first is used)
  static_assert(LT_WIN * LT_WIN % 256 == 0, "the staging loops are unrolled");
  constexpr int LT_CPT = LT_WIN * LT_WIN / 256;
  float zs[LT_CPT];
  uint8_t cs[LT_CPT];
  // the whole window in memory, inside the raster, and (a rank's) clear of the code-less last ring of the halo
  const int ring = RANKED ? 1 : 0;
  const bool inside = wy0 >= -w.halo + ring && wy0 + LT_WIN <= w.H + w.halo - ring && wx0 >= -w.halo + ring &&
                      wx0 + LT_WIN <= w.W + w.halo - ring && w.gy0 + wy0 >= 0 && w.gy0 + wy0 + LT_WIN <= w.Hg &&
                      w.gx0 + wx0 >= 0 && w.gx0 + wx0 + LT_WIN <= w.Wg;
  if (inside) {
    const float *__restrict__ d0 = dem + ((long long)wy0 * w.ld + wx0);
    const uint8_t *__restrict__ f0 = fdr + ((long long)wy0 * w.ld + wx0);
    const int ld = (int)w.ld;
#pragma unroll
    for (int j = 0; j < LT_CPT; j++) {
      const int i = (int)threadIdx.x + 256 * j;
      const int r = i / LT_WIN, c = i - r * LT_WIN;
      zs[j] = d0[r * ld + c];
      cs[j] = f0[r * ld + c];
    }
  } else {
#pragma unroll
    for (int j = 0; j < LT_CPT; j++) {
      const int i = (int)threadIdx.x + 256 * j;
      const int r = i / LT_WIN, c = i - r * LT_WIN;
      const int y = wy0 + r, x = wx0 + c;
      zs[j] = dt_readable(w, y, x) ? dem[(long long)y * w.ld + x] : DT_NODATA;
      cs[j] = (RANKED ? dt_has_code(w, y, x) : dt_readable(w, y, x)) ? fdr[(long long)y * w.ld + x] : (uint8_t)0;
    }
  }
  // a tile without a single valid cell (real rasters are basins inside a rectangle of nodata: half of the tiled
  // Example) has no entry anybody reads -- a walk never stands on nodata -- and is left as it is
  int valid = 0;
#pragma unroll
  for (int j = 0; j < LT_CPT; j++) {
    const int i = (int)threadIdx.x + 256 * j;
    const int r = i / LT_WIN, c = i - r * LT_WIN;
    valid |= (r >= LT_M && r < LT_M + 64 && c >= LT_M && c < LT_M + 64 && zs[j] != DT_NODATA) ? 1 : 0;
  }
  if (!__syncthreads_or(valid)) return;
#pragma unroll
  for (int j = 0; j < LT_CPT; j++) {
    const int i = (int)threadIdx.x + 256 * j;
    const int r = i / LT_WIN, c = i - r * LT_WIN;
    s_c[r * LT_LD + c].x = __float_as_uint(ds_lift_z(zs[j]));
  }
  __syncthreads();
  const char *__restrict__ base = (const char *)s_c;
#pragma unroll
  for (int j = 0; j < LT_CPT; j++) {
    const int i = (int)threadIdx.x + 256 * j;
    const int r = i / LT_WIN, c = i - r * LT_WIN;
    // (no walk of 8 moves from the tile leaves from the window's ring: its successors may lie outside the window)
    const bool edge = r == 0 || r == LT_WIN - 1 || c == 0 || c == LT_WIN - 1;
    uint32_t m = edge ? 0u : s_tab[cs[j]];
    const int q = (r * LT_LD + c) * 8;
    // (a successor that is not in memory was staged as nodata)
    if (__uint_as_float(*(const uint32_t *)(base + q + (int)(short)m)) == DT_NODATA) m = 0u;
    s_c[r * LT_LD + c].y = m;
  }
  __syncthreads();
  const int rows = w.H + w.halo, cols = w.W + w.halo;  // (exclusive ends of the memory, rank coordinates)
#pragma unroll 1
  for (int jb = 0; jb < 16; jb += 4) {
    int p[4], p0[4];
    uint32_t m[4], acc[4];
    float mz[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int c = (int)threadIdx.x + 256 * (jb + k);
      p0[k] = p[k] = (((c >> 6) + LT_M) * LT_LD + (c & 63) + LT_M) * 8;
      m[k] = (*(const uint2 *)(base + p[k])).y;
      acc[k] = 0u;
      mz[k] = __builtin_inff();
    }
#pragma unroll
    for (int step = 0; step < 8; step++) {
#pragma unroll
      for (int k = 0; k < 4; k++) {
        p[k] += (int)(short)m[k];
        acc[k] += m[k] >> 16;
        const uint2 e = *(const uint2 *)(base + p[k]);
        mz[k] = fminf(mz[k], __uint_as_float(e.x));
        m[k] = e.y;
      }
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int c = (int)threadIdx.x + 256 * (jb + k);
      const int y = y0 + (c >> 6), x = x0 + (c & 63);
      if (y >= rows || x >= cols) continue;
      const uint32_t len = acc[k] & 0xFFu, nd = acc[k] >> 8;
      const int dp = (p[k] - p0[k]) / 8;  // dy * LT_LD + dx with |dx| <= 8
      const int dy = (dp + 8 * LT_LD + LT_LD / 2) / LT_LD - 8, dx = dp - dy * LT_LD;
      // (a walk that made no move only ever read the cell it stands on: its skip has no lowest height)
      T[ds_mem_index(w, y, x)] = make_uint2(ds_lift_pack(dy, dx, len, nd) | (len < 8u ? DS_LIFT_STOP : 0u),
                                            len ? __float_as_uint(mz[k]) : __float_as_uint(__builtin_inff()));
    }
  }
}
// SPARSE upper levels (round 4).  A walk that passes through a cell ends no later than that cell's OWN walk does
// wherever the heights do not rise along the way, so skips of 16 .. 64 moves are worth having only at cells whose own
// walk is long: the queued ones -- 13 % of the valid cells of the tiled Example, where a doubling pass over all 214 M
// cells took 0.79 ms per level.  The tables stay dense arrays (a lookup is an index), but above the 8-move level only
// the entries of the QUEUED cells are computed, and a byte per cell (`dom`, cleared by the launcher, set by the first
// round) says which exist; where there is none a walk uses the 8-move table, which every cell has.  (With the 8-move
// entries sparse as well a few walks of the Example -- out of a pit and across steep ground, where no cell's own walk
// is long -- made a thousand single moves and the kernel waited 0.6 ms for them.)
// skips of up to 2 L moves from skips of L, for the queued cells; a skip whose end has no entry stays as it is.
// FIRST: from the 8-move table (every cell has an entry); marks the queued cells in `dom`
template <bool FIRST>
__global__ __launch_bounds__(256) void k_ds_tab_double(DsQueue queue, const uint2 *__restrict__ A,
                                                      uint2 *__restrict__ B, uint8_t *__restrict__ dom, int ld,
                                                      uint32_t lift_min) {
  if (*queue.count < lift_min) return;
  const uint32_t total = min(*queue.count, queue.capacity);
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    const uint32_t cell = queue.entries[i].x;
    uint2 a = A[cell];
    if (!(a.x & DS_LIFT_STOP)) {
      const uint32_t t = (uint32_t)((int)cell + ds_lift_dy(a.x) * ld + ds_lift_dx(a.x));
      if (FIRST || dom[t]) {
        const uint2 b = A[t];
        // the four fields add without carries (offsets within +-64 of the start, <= 64 moves): one addition, less the
        // second entry's two biases; the stop bit is the second entry's
        a.x += b.x - DS_LIFT_BIAS2;
        a.y = __float_as_uint(fminf(__uint_as_float(a.y), __uint_as_float(b.y)));
      }
    }
    B[cell] = a;
    if (FIRST) dom[cell] = (uint8_t)1;
  }
}
// the queued walks: skips while no cell of a skip can end the walk, then the moves that remain, one by one
template <bool RANKED>
__global__ __launch_bounds__(256, 8) void k_ds_finish(const float *__restrict__ dem, const uint8_t *__restrict__ fdr, DtWin w,
                                                  double px, double dz, float dzf, int raw, float *__restrict__ out,
                                                  DsQueue queue, const uint2 *__restrict__ T,
                                                  const uint2 *__restrict__ T8, const uint8_t *__restrict__ dom,
                                                  int *__restrict__ n_unresolved, uint32_t lift_min, DsWalkOut wo) {
  const uint32_t total = min(*queue.count, queue.capacity);
  const bool lifted = T != nullptr && *queue.count >= lift_min;
  const double dcard = px, ddiag = px * sqrt(2.0);
  uint32_t unres = 0u;
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    const uint4 e = queue.entries[i];
    int y0, x0, y, x;
    ds_mem_cell(w, e.x, y0, x0);
    ds_mem_cell(w, e.y, y, x);
    const float z0 = dem[(long long)y0 * w.ld + x0];
    uint32_t loop = e.z, nd = e.w;
    if (lifted) {
      // skips of 64 moves, then of 8 (the table of the third doubling round is kept): < 16 moves are left for the
      // move-by-move code
#pragma unroll
      for (int level = 0; level < 2; level++) {
        const uint2 *__restrict__ tab = level ? T8 : T;
        for (;;) {
          const uint32_t at = ds_mem_index(w, y, x);
          // (the long skips exist for the queued cells only -- `dom`, fetched with the entry; the 8-move table is dense)
          const uint32_t have = level ? 1u : (uint32_t)dom[at];
          const uint2 t = tab[at];
          const uint32_t len = have ? ds_lift_len(t.x) : 0u;
          // every cell of the skip leaves the drop below dz (the lowest one does), and the cap is beyond it
          if (len == 0u || !(z0 - __uint_as_float(t.y) < dzf) || loop + len > 4999u) break;
          y += ds_lift_dy(t.x);
          x += ds_lift_dx(t.x);
          loop += len;
          nd += ds_lift_nd(t.x);
        }
      }
    }
    const float drop = z0 - dem[(long long)y * w.ld + x];
    ds_finish_cell<RANKED>(w, dem, fdr, y0, x0, z0, drop, loop, nd, false, y, x, dcard, ddiag, dz, dzf, raw,
                           out + (long long)y0 * w.ld + x0, unres, wo);
  }
  // (a rank's walks that leave its memory: one atomic per wave)
  for (int o = 32; o; o >>= 1) unres += (uint32_t)__shfl_xor((int)unres, o);
  if ((threadIdx.x & 63u) == 0u && unres && n_unresolved) atomicAdd(n_unresolved, (int)unres);
}

// ---- walks across rank borders: walker records, see DsWalkOut -------------------------------------------------------
// Walker records (see DsWalkOut) standing in this rank's memory, advanced in place until they finish (DSW_DONE, result
// in r2.z) or reach the end of the rank's memory again.  Counting walkers cross the rank in skips of 64 and 8 moves when
// the rank's skip tables exist (T / T8 of the long-walk workspace, built when *qcount >= lift_min), then move by move
// with the exact exits of the reference's walk (ds_finish_cell's continuation loop); a finished counting walker whose
// quotient is not provably the reference's turns into a DSW_SEQ walker at its start cell (the caller sends it to the
// owner of that cell), and DSW_SEQ walkers make the reference's own moves with its sequential float64 sum.
__global__ __launch_bounds__(256) void k_ds_walk(const float *__restrict__ dem, const uint8_t *__restrict__ fdr, DtWin w,
                                                double px, double dz, float dzf, int64_t n, uint4 *__restrict__ rec,
                                                const uint2 *__restrict__ T, const uint2 *__restrict__ T8,
                                                const uint8_t *__restrict__ dom, const uint32_t *__restrict__ qcount,
                                                uint32_t lift_min, float *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint4 r0 = rec[3 * i], r1 = rec[3 * i + 1];
  if (r1.w & DSW_DONE) {
    // a finished walker is sent to the owner of its start cell: with `out` (the route form) its value is written here
    // and the record marked DSW_HOME, which drops it from the next exchange
    if (out && !(r1.w & DSW_HOME)) {
      const int ys = (int)r0.x - w.gy0, xs = (int)r0.y - w.gx0;
      if (dt_in_core(w, ys, xs)) {
        out[(long long)ys * w.ld + xs] = __uint_as_float(rec[3 * i + 2].z);
        r1.w |= DSW_HOME;
        rec[3 * i + 1] = r1;
      }
    }
    return;
  }
  int y = (int)r0.z - w.gy0, x = (int)r0.w - w.gx0;
  if (!dt_has_code(w, y, x)) return;  // not mine: somebody else's walker
  const double dcard = px, ddiag = px * sqrt(2.0);
  const float z0 = __uint_as_float(r1.z);
  uint32_t loop = r1.x, nd = r1.y;
  bool done = false;
  float value = 0.0f;
  if (!(r1.w & DSW_SEQ)) {
    if (T != nullptr && qcount != nullptr && *qcount >= lift_min) {
#pragma unroll
      for (int level = 0; level < 2; level++) {
        const uint2 *__restrict__ tab = level ? T8 : T;
        for (;;) {
          const uint32_t at = ds_mem_index(w, y, x);
          const uint32_t have = level ? 1u : (uint32_t)dom[at];
          const uint2 t = tab[at];
          const uint32_t len = have ? ds_lift_len(t.x) : 0u;
          if (len == 0u || !(z0 - __uint_as_float(t.y) < dzf) || loop + len > 4999u) break;
          y += ds_lift_dy(t.x);
          x += ds_lift_dx(t.x);
          loop += len;
          nd += ds_lift_nd(t.x);
        }
      }
    }
    float drop = loop > 0u ? z0 - dem[(long long)y * w.ld + x] : 0.0f;
    uint32_t code = dt_has_code(w, y, x) ? (uint32_t)fdr[(long long)y * w.ld + x] : 0u;
    bool handover = false;
    while ((double)drop < dz) {
      if (!dt_has_code(w, y, x)) { handover = true; break; }
      if (!dt_d8_valid(code)) break;                       // failed: the walk so far is the result
      int dy, dx;
      dt_d8_delta(code, dy, dx);
      const int ny = y + dy, nx = x + dx;
      if (!dt_in_global(w, ny, nx)) break;                 // downslope.py:209-228
      if (!dt_readable(w, ny, nx)) { handover = true; break; }
      const long long on = (long long)ny * w.ld + nx;
      const float zt = dem[on];
      code = fdr[on];
      if (zt == DT_NODATA) break;                          // nodata ahead: stop without moving (:231-281)
      y = ny;
      x = nx;
      nd += (dy != 0 && dx != 0) ? 1u : 0u;
      drop = z0 - zt;
      if (++loop == 5000u) break;                          // downslope.py:303-304
    }
    if (!handover) {
      if (loop == 0u) {
        done = true;  // no move: distance 0 -> 0 (downslope.py:306-309)
      } else {
        bool safe;
        value = ds_quotient(drop, loop, nd, dcard, ddiag, safe);
        if (safe) {
          done = true;
        } else {  // start again with the reference's sequential sum, at the start cell (whoever owns it)
          r0.z = r0.x;
          r0.w = r0.y;
          rec[3 * i] = r0;
          rec[3 * i + 1] = make_uint4(0u, 0u, r1.z, DSW_SEQ);
          rec[3 * i + 2] = make_uint4(0u, 0u, 0u, 0u);
          return;
        }
      }
    }
    r0.z = (uint32_t)(y + w.gy0);
    r0.w = (uint32_t)(x + w.gx0);
    rec[3 * i] = r0;
    rec[3 * i + 1] = make_uint4(loop, nd, r1.z, done ? DSW_DONE : 0u);
    if (done) rec[3 * i + 2] = make_uint4(0u, 0u, __float_as_uint(value), 0u);
    return;
  }
  // DSW_SEQ: the reference's own walk, its float64 sum carried in the record
  const uint4 r2 = rec[3 * i + 2];
  double d = __longlong_as_double((long long)(((unsigned long long)r2.y << 32) | (unsigned long long)r2.x));
  float drop = loop > 0u ? z0 - dem[(long long)y * w.ld + x] : 0.0f;
  while ((double)drop < dz) {
    if (!dt_has_code(w, y, x)) break;  // the end of my memory: hand over
    const uint32_t code = fdr[(long long)y * w.ld + x];
    if (!dt_d8_valid(code)) { done = true; break; }
    int dy, dx;
    dt_d8_delta(code, dy, dx);
    const int ny = y + dy, nx = x + dx;
    if (!dt_in_global(w, ny, nx)) { done = true; break; }
    if (!dt_readable(w, ny, nx)) break;
    const float zt = dem[(long long)ny * w.ld + nx];
    if (zt == DT_NODATA) { done = true; break; }
    y = ny;
    x = nx;
    d += (dy != 0 && dx != 0) ? ddiag : dcard;
    drop = z0 - zt;
    if (++loop == 5000u) { done = true; break; }
  }
  if (!((double)drop < dz)) done = true;
  r0.z = (uint32_t)(y + w.gy0);
  r0.w = (uint32_t)(x + w.gx0);
  rec[3 * i] = r0;
  rec[3 * i + 1] = make_uint4(loop, 0u, r1.z, DSW_SEQ | (done ? DSW_DONE : 0u));
  const unsigned long long db = (unsigned long long)__double_as_longlong(d);
  value = d == 0.0 ? 0.0f : (float)((double)drop / d);
  rec[3 * i + 2] = make_uint4((uint32_t)db, (uint32_t)(db >> 32), __float_as_uint(value), 0u);
}
// work (optional): the rank's long-walk workspace as dt_dev_downslope_lift_w left it (queue | tables)
int dt_launch_ds_walk(hipStream_t s, const DtWin &w, const float *dem, const uint8_t *fdr, double px, double dz,
                      int64_t n, void *rec, void *work, float *out) {
  if (n <= 0) return DT_OK;
  float dzf = (float)dz;
  if ((double)dzf < dz) dzf = nextafterf(dzf, INFINITY);
  const uint2 *T = nullptr, *T8 = nullptr;
  const uint8_t *dom = nullptr;
  const uint32_t *qc = nullptr;
  if (work) {
    const size_t cells = (size_t)(w.H + 2 * (int64_t)w.halo) * (size_t)w.ld;
    char *t0 = (char *)work + dt_downslope_queue_bytes(w.H, w.W);
    qc = (const uint32_t *)work;
    T = (const uint2 *)t0;
    T8 = (const uint2 *)(t0 + 2 * dt_align256(cells * 8));
    dom = (const uint8_t *)(t0 + 3 * dt_align256(cells * 8));
  }
  hipLaunchKernelGGL(k_ds_walk, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dem, fdr, w, px, dz, dzf, n, (uint4 *)rec,
                     T, T8, dom, qc, dt_downslope_lift_min(w.H, w.W), out);
  return DT_OK;
}
// ---- where the walker records go next (tiling.finish_downslope's exchange, prepared on the device) -------------------
// The layout of the ranks: row_starts[0 .. ty] / col_starts[0 .. tx] = first global row / column of every rank row /
// column and the raster's end; rank = rank row * tx + rank column.
#define DSW_MAX_RANKS 1024
__device__ __forceinline__ int dsw_band(const int32_t *__restrict__ starts, int n, int v) {
  int k = 0;
  for (int j = 1; j < n; j++) k += (v >= starts[j]) ? 1 : 0;  // (a handful of bands: a scan, no search)
  return k;
}
// dest[i] = the rank record i goes to (-1: it stays -- a finished walker that is home, DSW_HOME): the owner of its start
// cell when it has finished, the owner of the cell it stands on otherwise; counts[d] += records for rank d,
// counts[n_ranks] += the ones still on their way
__global__ __launch_bounds__(256) void k_dsw_classify(const uint4 *__restrict__ rec, int64_t n,
                                                     const int32_t *__restrict__ row_starts, int ty,
                                                     const int32_t *__restrict__ col_starts, int tx,
                                                     int32_t *__restrict__ dest, int32_t *__restrict__ counts) {
  __shared__ int s_cnt[DSW_MAX_RANKS + 1];
  const int nr = ty * tx;
  for (int j = threadIdx.x; j <= nr; j += 256) s_cnt[j] = 0;
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    const uint4 r0 = rec[3 * i], r1 = rec[3 * i + 1];
    int d = -1;
    if (!(r1.w & DSW_HOME)) {
      const bool done = (r1.w & DSW_DONE) != 0u;
      const int gy = (int)(done ? r0.x : r0.z), gx = (int)(done ? r0.y : r0.w);
      d = dsw_band(row_starts, ty, gy) * tx + dsw_band(col_starts, tx, gx);
      atomicAdd(&s_cnt[d], 1);
      if (!done) atomicAdd(&s_cnt[nr], 1);
    }
    dest[i] = d;
  }
  __syncthreads();
  for (int j = threadIdx.x; j <= nr; j += 256)
    if (s_cnt[j]) atomicAdd(&counts[j], s_cnt[j]);
}
// send = the records grouped by destination rank (the order inside a group is whatever the atomics give: records are
// independent of one another); cursor[d]: zeroed by the launcher
__global__ __launch_bounds__(256) void k_dsw_scatter(const uint4 *__restrict__ rec, int64_t n,
                                                    const int32_t *__restrict__ dest,
                                                    const int32_t *__restrict__ counts, int nr,
                                                    int32_t *__restrict__ cursor, uint4 *__restrict__ send) {
  __shared__ int s_cnt[DSW_MAX_RANKS], s_base[DSW_MAX_RANKS];
  for (int j = threadIdx.x; j < nr; j += 256) {
    s_cnt[j] = 0;
    s_base[j] = counts[j];
  }
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int d = i < n ? dest[i] : -1;
  int r = 0;
  if (d >= 0) r = atomicAdd(&s_cnt[d], 1);
  // first slot of every rank's group: an exclusive scan of the counts (Hillis-Steele in LDS: log2(nr) steps, up to
  // four entries per thread)
  for (int step = 1; step < nr; step <<= 1) {
    int t[DSW_MAX_RANKS / 256];
#pragma unroll
    for (int q = 0; q < DSW_MAX_RANKS / 256; q++) {
      const int j = (int)threadIdx.x + 256 * q;
      t[q] = (j < nr && j >= step) ? s_base[j - step] : 0;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < DSW_MAX_RANKS / 256; q++) {
      const int j = (int)threadIdx.x + 256 * q;
      if (j < nr) s_base[j] += t[q];
    }
    __syncthreads();
  }
  // (inclusive -> exclusive, plus what the other workgroups have taken of the group)
  for (int j = threadIdx.x; j < nr; j += 256) {
    const int mine = s_cnt[j];
    s_base[j] = s_base[j] - counts[j] + (mine ? atomicAdd(&cursor[j], mine) : 0);
  }
  __syncthreads();
  if (d >= 0) {
    const int64_t slot = (int64_t)s_base[d] + r;
    send[3 * slot] = rec[3 * i];
    send[3 * slot + 1] = rec[3 * i + 1];
    send[3 * slot + 2] = rec[3 * i + 2];
  }
}
// scratch: int32[n + n_ranks]; counts: int32[n_ranks + 1]
int dt_launch_ds_route(hipStream_t s, int64_t n, const void *rec, const int32_t *row_starts, int ty,
                       const int32_t *col_starts, int tx, void *send, int32_t *counts, int32_t *scratch) {
  const int nr = ty * tx;
  DT_REQUIRE(nr >= 1 && nr <= DSW_MAX_RANKS, "1 .. 1024 ranks");
  DT_HIP(hipMemsetAsync(counts, 0, sizeof(int32_t) * (size_t)(nr + 1), s));
  if (n <= 0) return DT_OK;
  int32_t *dest = scratch, *cursor = scratch + n;
  DT_HIP(hipMemsetAsync(cursor, 0, sizeof(int32_t) * (size_t)nr, s));
  const dim3 g((unsigned)((n + 255) / 256)), b(256);
  hipLaunchKernelGGL(k_dsw_classify, g, b, 0, s, (const uint4 *)rec, n, row_starts, ty, col_starts, tx, dest, counts);
  hipLaunchKernelGGL(k_dsw_scatter, g, b, 0, s, (const uint4 *)rec, n, (const int32_t *)dest, (const int32_t *)counts, nr,
                     cursor, (uint4 *)send);
  return DT_OK;
}
// start records for cells marked -50 (the fallback when the emission buffer was too small, or a tile without one):
// walkers at their start cells, no move made
__global__ __launch_bounds__(256) void k_ds_walk_seed(const float *__restrict__ dem, DtWin w, int64_t n,
                                                     const int32_t *__restrict__ ys, const int32_t *__restrict__ xs,
                                                     uint4 *__restrict__ rec) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int y = ys[i], x = xs[i];
  const uint32_t gy = (uint32_t)(y + w.gy0), gx = (uint32_t)(x + w.gx0);
  rec[3 * i] = make_uint4(gy, gx, gy, gx);
  rec[3 * i + 1] = make_uint4(0u, 0u, __float_as_uint(dem[(long long)y * w.ld + x]), 0u);
  rec[3 * i + 2] = make_uint4(0u, 0u, 0u, 0u);
}
int dt_launch_ds_walk_seed(hipStream_t s, const DtWin &w, const float *dem, int64_t n, const int32_t *ys,
                           const int32_t *xs, void *rec) {
  if (n > 0)
    hipLaunchKernelGGL(k_ds_walk_seed, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dem, w, n, ys, xs, (uint4 *)rec);
  return DT_OK;
}
// workspaces of the long-walk acceleration for an H x W raster: the QUEUE (counter | one entry per two cells) and the
// TABLES (two ping-pong skip tables and the 8-move table that is kept); dt_downslope_lift_bytes = both, back to back
static size_t ds_queue_capacity(int64_t H, int64_t W) { return (size_t)((H * W + 1) / 2); }
size_t dt_downslope_queue_bytes(int64_t H, int64_t W) { return 256 + dt_align256(ds_queue_capacity(H, W) * 16); }
size_t dt_downslope_tables_bytes(int64_t H, int64_t W) {
  return 3 * dt_align256((size_t)H * W * 8) + dt_align256((size_t)H * W);  // three tables | which entries exist
}
size_t dt_downslope_lift_bytes(int64_t H, int64_t W) {
  return dt_downslope_queue_bytes(H, W) + dt_downslope_tables_bytes(H, W);
}
uint32_t dt_downslope_lift_min(int64_t H, int64_t W) { return (uint32_t)std::max<int64_t>(DS_LIFT_MIN, H * W / 256); }
// ... of a window: the queue holds core cells, the tables cover the window's memory (core + halo, row stride ld)
static size_t ds_mem_cells(const DtWin &w) { return (size_t)(w.H + 2 * (int64_t)w.halo) * (size_t)w.ld; }
size_t dt_downslope_tables_bytes_w(const DtWin &w) {
  return 3 * dt_align256(ds_mem_cells(w) * 8) + dt_align256(ds_mem_cells(w));
}
size_t dt_downslope_lift_bytes_w(const DtWin &w) {
  return dt_downslope_queue_bytes(w.H, w.W) + dt_downslope_tables_bytes_w(w);
}
// qwork / twork (optional): long walks are queued and finished with skip tables (see DsQueue); in a rank's window the
// ones that stay in the rank's memory -- the others are counted in n_unresolved as without the workspace.  phase 0: everything; 1: the window kernel with the queue only; 2: the
// tables (when twork is given; built only if enough walks were queued) and the queued walks -- so that a caller who may
// synchronise can look at the queue's counter (the first word of qwork) after phase 1 and allocate tables only when a
// raster needs them.
int dt_launch_downslope(hipStream_t s, const DtWin &w, const float *dem, const uint8_t *fdr, double px,
                        double dz, int raw, float *out, int *n_unresolved, void *qwork, void *twork, int phase,
                        void *walkers, size_t walkers_bytes) {
  const int64_t H = w.H, W = w.W;
  int64_t n = H * W;
  if (n == 0) return DT_OK;
  int tiles_x = (int)((W + DW_CORE - 1) / DW_CORE), tiles_y = (int)((H + DW_CORE - 1) / DW_CORE);
  int64_t ntiles = (int64_t)tiles_x * tiles_y;
  // (double)drop < dz  <=>  drop < dzf with dzf the smallest float >= dz
  float dzf = (float)dz;
  if ((double)dzf < dz) dzf = nextafterf(dzf, INFINITY);
  DsQueue q;
  // walkers: [count u32, pad to 256 bytes | records of DSW_WORDS words] -- where the walks that leave a rank's memory
  // are emitted (ranked kernels only)
  DsWalkOut wo;
  if (walkers && walkers_bytes >= 256 + DSW_WORDS * 4) {
    wo.count = (uint32_t *)walkers;
    wo.rec = (uint4 *)((char *)walkers + 256);
    wo.capacity = (uint32_t)std::min<size_t>((walkers_bytes - 256) / (DSW_WORDS * 4), 0x7FFFFFFFu);
    if (phase != 2) DT_HIP(hipMemsetAsync(wo.count, 0, sizeof(uint32_t), s));
  }
  uint2 *tab[3] = {nullptr, nullptr, nullptr};  // two ping-pong tables and the 8-move table that is kept
  const bool ranked = !(w.halo == 0 && w.gy0 == 0 && w.gx0 == 0 && w.Hg == w.H && w.Wg == w.W);
  // margin of the LDS window around the 64 x 64 core: walks that reach the window's ring carry on in global memory
  const int m = dt_debug_get(DT_DBG_DS_MARGIN);
  if (qwork && ds_mem_cells(w) < 0x7FFFFFFFull && m != 16 && m != 20) {  // (the A/B margins run without the queue)
    q.count = (uint32_t *)qwork;
    q.entries = (uint4 *)((char *)qwork + 256);
    q.capacity = (uint32_t)ds_queue_capacity(H, W);
    if (twork) {
      tab[0] = (uint2 *)twork;
      tab[1] = (uint2 *)((char *)tab[0] + dt_align256(ds_mem_cells(w) * 8));
      tab[2] = (uint2 *)((char *)tab[1] + dt_align256(ds_mem_cells(w) * 8));
    }
  }
  DT_REQUIRE(phase == 0 || q.entries, "the phases of the long-walk form need the queue workspace (and a window of < 2^31 cells)");
  if (phase != 2) {
    if (q.entries) DT_HIP(hipMemsetAsync(q.count, 0, sizeof(uint32_t), s));
    if (m == 16)
      hipLaunchKernelGGL(k_downslope_win<16>, dim3((unsigned)ntiles), dim3(1024), 0, s, dem, fdr, w, px, dz, dzf, raw,
                         out, tiles_x, (int)ntiles, n_unresolved);
    else if (m == 20)
      hipLaunchKernelGGL(k_downslope_win<20>, dim3((unsigned)ntiles), dim3(1024), 0, s, dem, fdr, w, px, dz, dzf, raw,
                         out, tiles_x, (int)ntiles, n_unresolved);
    else if (ranked && q.entries)
      hipLaunchKernelGGL(k_downslope_win_rq, dim3((unsigned)ntiles), dim3(1024), 0, s, dem, fdr, w, px, dz, dzf, raw,
                         out, tiles_x, (int)ntiles, n_unresolved, q, wo);
    else if (ranked)
      hipLaunchKernelGGL(k_downslope_win_r, dim3((unsigned)ntiles), dim3(1024), 0, s, dem, fdr, w, px, dz, dzf, raw,
                         out, tiles_x, (int)ntiles, n_unresolved, wo);
    else if (q.entries)
      hipLaunchKernelGGL(k_downslope_win_q, dim3((unsigned)ntiles), dim3(1024), 0, s, dem, fdr, w, px, dz, dzf, raw,
                         out, tiles_x, (int)ntiles, n_unresolved, q);
    else
      hipLaunchKernelGGL(k_downslope_win<24>, dim3((unsigned)ntiles), dim3(1024), 0, s, dem, fdr, w, px, dz, dzf, raw,
                         out, tiles_x, (int)ntiles, n_unresolved);
  }
  if (q.entries && phase != 1) {
    const dim3 b(256);
    const uint32_t lift_min = dt_downslope_lift_min(H, W);
    uint2 *src = nullptr;
    uint8_t *dom = nullptr;
    const unsigned fin_blocks = (unsigned)std::min<size_t>((q.capacity + 255) / 256, 8192);
    if (tab[0]) {
      // every kernel of the tables returns at once when fewer than lift_min walks were queued
      // 8 moves per skip for every cell (tab[2], kept) -> 16 -> 32 -> 64 for the queued cells (`dom`)
      dom = (uint8_t *)tab[2] + dt_align256(ds_mem_cells(w) * 8);
      DT_HIP(hipMemsetAsync(dom, 0, ds_mem_cells(w), s));
      const int rows = (int)(H + 2 * w.halo), cols = (int)(W + 2 * w.halo);
      const int ltx = (cols + 63) / 64, lty = (rows + 63) / 64;
      const dim3 gt((unsigned)(ltx * lty));
      const uint32_t *qc = (const uint32_t *)q.count;
      if (ranked) hipLaunchKernelGGL(k_ds_lift_init<true>, gt, b, 0, s, dem, fdr, w, tab[2], qc, lift_min, ltx);
      else hipLaunchKernelGGL(k_ds_lift_init<false>, gt, b, 0, s, dem, fdr, w, tab[2], qc, lift_min, ltx);
      hipLaunchKernelGGL(k_ds_tab_double<true>, dim3(fin_blocks), b, 0, s, q, (const uint2 *)tab[2], tab[0], dom, (int)w.ld, lift_min);
      hipLaunchKernelGGL(k_ds_tab_double<false>, dim3(fin_blocks), b, 0, s, q, (const uint2 *)tab[0], tab[1], dom, (int)w.ld, lift_min);
      hipLaunchKernelGGL(k_ds_tab_double<false>, dim3(fin_blocks), b, 0, s, q, (const uint2 *)tab[1], tab[0], dom, (int)w.ld, lift_min);
      src = tab[0];
    }
    if (ranked)
      hipLaunchKernelGGL(k_ds_finish<true>, dim3(fin_blocks), b, 0, s, dem, fdr, w, px, dz, dzf, raw, out, q,
                         (const uint2 *)src, (const uint2 *)tab[2], (const uint8_t *)dom, n_unresolved, lift_min, wo);
    else
      hipLaunchKernelGGL(k_ds_finish<false>, dim3(fin_blocks), b, 0, s, dem, fdr, w, px, dz, dzf, raw, out, q,
                         (const uint2 *)src, (const uint2 *)tab[2], (const uint8_t *)dom, n_unresolved, lift_min, DsWalkOut());
  }
  return DT_OK;
}
int dt_launch_downslope_v1(hipStream_t s, const float *dem, const uint8_t *fdr, int64_t H, int64_t W,
                           double px, double dz, int raw, float *out) {
  int64_t n = H * W;
  if (n == 0) return DT_OK;
  hipLaunchKernelGGL(k_downslope, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dem, fdr, (int)H,
                     (int)W, px, dz, raw, out);
  return DT_OK;
}

// flowhand.hand_calculator alone (flowhand.py:414-442) on int64 indices
__global__ __launch_bounds__(256) void k_hand_i64(const float *__restrict__ dem,
                                                 const int64_t *__restrict__ idx, int64_t n,
                                                 float *__restrict__ hand) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float z = dem[i], h = DT_NODATA;
  int64_t k = idx[i];
  if (z != DT_NODATA && k != -100) {
    if (k < 0) k += n;  // numpy negative indexing, as the reference's dem[indices]
    if (k >= 0 && k < n) {
      h = z - dem[k];
      if (h < 0.0f && h != DT_NODATA) h = 0.0f;
    }
  }
  hand[i] = h;
}
int dt_launch_hand_i64(hipStream_t s, const float *dem, const int64_t *idx, int64_t n, float *hand) {
  if (n) hipLaunchKernelGGL(k_hand_i64, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dem, idx, n, hand);
  return DT_OK;
}

// ===========================================================================================
// Confusion counts for up to 24 thresholds in one pass (E2+E3, evaluation.py:90-171).
// Per lane: B0[t] / B2[t] = #cells classified 1 whose remapped benchmark value is 0 / 2;
// benchmark values other than {0, 2} (never produced by the reference's inputs) take a slow
// exact path.  Wave shuffle reduction, then one 64-bit atomic per counter per wave.
// ===========================================================================================
#define CF_MAXT 24
struct CfThresholds {
  double v[CF_MAXT];
};

__global__ __launch_bounds__(256) void k_confusion(const double *__restrict__ desc,
                                                  const int8_t *__restrict__ flood, int64_t n,
                                                  double nodata, int nth, int under,
                                                  CfThresholds th,
                                                  unsigned long long *__restrict__ counts4) {
  unsigned int b0[CF_MAXT], b2[CF_MAXT];
#pragma unroll
  for (int t = 0; t < CF_MAXT; t++) b0[t] = b2[t] = 0u;
  unsigned int t0 = 0u, t2 = 0u, t3 = 0u;  // cells with g == 0 / 2 / 3
  int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    double v = desc[i];
    int g = flood[i];
    if (g == 1) g = 2;            // evaluation.py:149
    else if (g == -100) g = 0;    // evaluation.py:150
    bool isn = (v == nodata) || (v != v);  // evaluation.py:111-121
    if (g == 0 || g == 2) {
      if (g == 0) t0++; else t2++;
#pragma unroll
      for (int t = 0; t < CF_MAXT; t++) {
        bool bin = !isn && (t < nth) && (under ? (v <= th.v[t]) : (v >= th.v[t]));
        b0[t] += (bin && g == 0) ? 1u : 0u;
        b2[t] += (bin && g == 2) ? 1u : 0u;
      }
    } else {
      // generic value: binary + g counted when it lands in 0..3
      for (int t = 0; t < nth; t++) {
        int bin = (!isn && (under ? (v <= th.v[t]) : (v >= th.v[t]))) ? 1 : 0;
        int r = bin + g;
        if (r >= 0 && r <= 3) atomicAdd(&counts4[t * 4 + r], 1ull);
      }
    }
  }
  (void)t3;
  // wave reduction
  for (int off = 32; off > 0; off >>= 1) {
    t0 += __shfl_down(t0, off);
    t2 += __shfl_down(t2, off);
#pragma unroll
    for (int t = 0; t < CF_MAXT; t++) {
      b0[t] += __shfl_down(b0[t], off);
      b2[t] += __shfl_down(b2[t], off);
    }
  }
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int t = 0; t < CF_MAXT; t++) {
      if (t < nth) {
        // class 0: g == 0, bin 0 | class 1: g == 0, bin 1 | class 2: g == 2, bin 0 | class 3: g == 2, bin 1
        atomicAdd(&counts4[t * 4 + 0], (unsigned long long)(t0 - b0[t]));
        atomicAdd(&counts4[t * 4 + 1], (unsigned long long)b0[t]);
        atomicAdd(&counts4[t * 4 + 2], (unsigned long long)(t2 - b2[t]));
        atomicAdd(&counts4[t * 4 + 3], (unsigned long long)b2[t]);
      }
    }
  }
}

int dt_launch_confusion(hipStream_t s, const double *desc, const int8_t *flood, int64_t n,
                        double nodata, const double *th_host, int nth, int under,
                        unsigned long long *counts4) {
  DT_REQUIRE(nth >= 1 && nth <= CF_MAXT, "1..24 thresholds per call");
  DT_HIP(hipMemsetAsync(counts4, 0, sizeof(unsigned long long) * 4 * nth, s));
  if (n == 0) return DT_OK;
  CfThresholds th;
  for (int t = 0; t < CF_MAXT; t++) th.v[t] = t < nth ? th_host[t] : 0.0;
  // each lane counts in 32 bits: keep <= 2^31 cells per lane (grid-stride over >= 1024 blocks)
  int64_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(k_confusion, dim3((unsigned)blocks), dim3(256), 0, s, desc, flood, n, nodata, nth,
                     under, th, counts4);
  return DT_OK;
}

// ===========================================================================================
// Device-resident evaluation helpers (E1)
// ===========================================================================================
// order-preserving map float -> uint32 for atomicMin / atomicMax on floats
__device__ __forceinline__ uint32_t dt_f2ord(float f) {
  uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float dt_ord2f(uint32_t o) {
  return __uint_as_float((o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o);
}
// out[0] = min, out[1] = max over cells with x > lower (NaN skipped); ordered-uint encoding
__global__ __launch_bounds__(256) void k_minmax_above(const float *__restrict__ x, int64_t n, float lower,
                                                     int use_lower, uint32_t *__restrict__ out) {
  uint32_t mn = 0xFFFFFFFFu, mx = 0u;
  int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    float v = x[i];
    if (v != v) continue;
    if (use_lower && !(v > lower)) continue;
    uint32_t o = dt_f2ord(v);
    mn = min(mn, o);
    mx = max(mx, o);
  }
  for (int off = 32; off > 0; off >>= 1) {
    mn = min(mn, (uint32_t)__shfl_down((int)mn, off));
    mx = max(mx, (uint32_t)__shfl_down((int)mx, off));
  }
  if ((threadIdx.x & 63) == 0) {
    atomicMin(&out[0], mn);
    atomicMax(&out[1], mx);
  }
}
__global__ void k_minmax_decode(const uint32_t *__restrict__ a, const uint32_t *__restrict__ b,
                                float *__restrict__ out3) {
  // a = {min, max} over all cells, b = {min over cells > global min, -}: np.unique(x)[0], [1], [-1]
  out3[0] = a[0] == 0xFFFFFFFFu ? NAN : dt_ord2f(a[0]);
  out3[1] = b[0] == 0xFFFFFFFFu ? NAN : dt_ord2f(b[0]);
  out3[2] = a[1] == 0u ? NAN : dt_ord2f(a[1]);
}
// smallest, second-smallest distinct and largest value of a float32 raster (what the example derives
// from np.unique(hand): elements[0], elements[1], elements[-1]; Example/example.py:113-115)
int dt_launch_unique_extremes(hipStream_t s, const float *x, int64_t n, uint32_t *work4, float *out3) {
  uint32_t init[4] = {0xFFFFFFFFu, 0u, 0xFFFFFFFFu, 0u};
  DT_HIP(hipMemcpyAsync(work4, init, sizeof(init), hipMemcpyHostToDevice, s));
  DT_HIP(hipStreamSynchronize(s));  // init[] is on the stack
  int64_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(k_minmax_above, dim3((unsigned)blocks), dim3(256), 0, s, x, n, 0.0f, 0, work4);
  // second pass needs the global minimum: read it back (tiny)
  uint32_t a0;
  DT_HIP(hipMemcpyAsync(&a0, work4, 4, hipMemcpyDeviceToHost, s));
  DT_HIP(hipStreamSynchronize(s));
  float lower = a0 == 0xFFFFFFFFu ? 0.0f : __builtin_bit_cast(float, (a0 & 0x80000000u) ? (a0 & 0x7FFFFFFFu) : ~a0);
  hipLaunchKernelGGL(k_minmax_above, dim3((unsigned)blocks), dim3(256), 0, s, x, n, lower, 1, work4 + 2);
  hipLaunchKernelGGL(k_minmax_decode, dim3(1), dim3(1), 0, s, work4, work4 + 2, out3);
  return DT_OK;
}

// evaluation.minMaxScale on a float32 raster (evaluation.py:5-9): numpy keeps float32 there, so the
// arithmetic is float32; NaN where x == nodata (or x is NaN); widened to the float64 the confusion kernel
// reads
__global__ __launch_bounds__(256) void k_minmax_scale(const float *__restrict__ x, int64_t n, float mn,
                                                     float mx, float nodata, double *__restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float v = x[i];
  float d = (v == nodata) ? NAN : (v - mn) / (mx - mn);
  out[i] = (double)d;
}
int dt_launch_minmax_scale(hipStream_t s, const float *x, int64_t n, float mn, float mx, float nodata,
                           double *out) {
  if (n) hipLaunchKernelGGL(k_minmax_scale, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, n, mn, mx,
                            nodata, out);
  return DT_OK;
}

// ===========================================================================================
// evaluation.minMaxScale / binary_map / avaliacao as kernels (E1-E3, evaluation.py:5-9, 90-123, 126-171)
// ===========================================================================================
// minMaxScale in the arithmetic numpy would use for the raster's dtype (T = float for a float32 raster,
// double for integer / float64 rasters): NaN where x == nodata or x is NaN, else (x - mn) / (mx - mn)
template <typename TI, typename T>
__global__ __launch_bounds__(256) void k_minmax_scale_t(const TI *__restrict__ x, int64_t n, T mn, T mx, T nodata,
                                                       T *__restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  T v = (T)x[i];
  out[i] = (v == nodata || v != v) ? (T)NAN : (v - mn) / (mx - mn);
}
int dt_launch_minmax_scale_f32f32(hipStream_t s, const float *x, int64_t n, float mn, float mx, float nodata,
                                  float *out) {
  if (n) hipLaunchKernelGGL((k_minmax_scale_t<float, float>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, n,
                            mn, mx, nodata, out);
  return DT_OK;
}
int dt_launch_minmax_scale_f64(hipStream_t s, const double *x, int64_t n, double mn, double mx, double nodata,
                               double *out) {
  if (n) hipLaunchKernelGGL((k_minmax_scale_t<double, double>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x,
                            n, mn, mx, nodata, out);
  return DT_OK;
}
// float32 raster scaled in float64 (an integer-valued HAND kept as float32 on the device: numpy scales the
// example's int16 HAND in float64)
int dt_launch_minmax_scale_f32f64(hipStream_t s, const float *x, int64_t n, double mn, double mx, double nodata,
                                  double *out) {
  if (n) hipLaunchKernelGGL((k_minmax_scale_t<float, double>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x,
                            n, mn, mx, nodata, out);
  return DT_OK;
}

// One pass over a descriptor and a benchmark flood map at ONE threshold (what Example/example.py:139-147 does
// with binary_map + avaliacao): optional outputs
//   binary[i]  = 1 where the descriptor is on the flooded side of the threshold, 0 where it is not, is NaN or
//                equals `nodata` (the caller passes desc[0, 0], evaluation.py:111)
//   flood[i]   remapped in place 1 -> 2, -100 -> 0 (evaluation.py:149-150) when `remap` is set
//   klass[i]   = binary + remapped flood (evaluation.py:151)
//   counts4[v] = number of cells of class v = 0..3
// With bin_in the binary map is an input (avaliacao on its own), desc is not read.
template <typename T>
__global__ __launch_bounds__(256) void k_classify(const T *__restrict__ desc, const int32_t *__restrict__ bin_in,
                                                 int8_t *__restrict__ flood, int64_t n, T nodata, T th, int under,
                                                 int remap, uint8_t *__restrict__ binary,
                                                 int32_t *__restrict__ klass,
                                                 unsigned long long *__restrict__ counts4) {
  unsigned int cnt[4] = {0u, 0u, 0u, 0u};
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    int b;
    if (bin_in) {
      b = bin_in[i];
    } else {
      const T v = desc[i];
      const bool on = under ? (v <= th) : (v >= th);  // false for NaN
      b = (on && !(v == nodata)) ? 1 : 0;
    }
    int g = flood[i];
    if (g == 1) g = 2;
    else if (g == -100) g = 0;
    if (remap) flood[i] = (int8_t)g;
    const int r = b + g;
    if (binary) binary[i] = (uint8_t)b;
    if (klass) klass[i] = r;
    if (r >= 0 && r <= 3) cnt[r]++;
  }
#pragma unroll
  for (int v = 0; v < 4; v++) {
    unsigned int c = cnt[v];
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(&counts4[v], (unsigned long long)c);
  }
}
template <typename T>
static int dt_launch_classify_t(hipStream_t s, const T *desc, const int32_t *bin_in, int8_t *flood, int64_t n,
                                T nodata, T th, int under, int remap, uint8_t *binary, int32_t *klass,
                                unsigned long long *counts4) {
  DT_HIP(hipMemsetAsync(counts4, 0, 4 * sizeof(unsigned long long), s));
  if (n == 0) return DT_OK;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(k_classify<T>, dim3((unsigned)blocks), dim3(256), 0, s, desc, bin_in, flood, n, nodata, th, under,
                     remap, binary, klass, counts4);
  return DT_OK;
}
int dt_launch_classify_f64(hipStream_t s, const double *desc, const int32_t *bin_in, int8_t *flood, int64_t n,
                           double nodata, double th, int under, int remap, uint8_t *binary, int32_t *klass,
                           unsigned long long *counts4) {
  return dt_launch_classify_t<double>(s, desc, bin_in, flood, n, nodata, th, under, remap, binary, klass, counts4);
}
int dt_launch_classify_f32(hipStream_t s, const float *desc, const int32_t *bin_in, int8_t *flood, int64_t n,
                           float nodata, float th, int under, int remap, uint8_t *binary, int32_t *klass,
                           unsigned long long *counts4) {
  return dt_launch_classify_t<float>(s, desc, bin_in, flood, n, nodata, th, under, remap, binary, klass, counts4);
}

// ---- HBM copy micro-benchmark (the practical bandwidth ceiling the roofline fractions compare with) ----
template <int UNROLL>
__global__ __launch_bounds__(256) void k_membench_copy(const float4 *__restrict__ a, float4 *__restrict__ b,
                                                      int64_t n4) {
  int64_t stride = (int64_t)gridDim.x * 256 * UNROLL;
  for (int64_t i = (int64_t)blockIdx.x * 256 * UNROLL + threadIdx.x; i < n4; i += stride) {
    float4 v[UNROLL];
#pragma unroll
    for (int k = 0; k < UNROLL; k++)
      if (i + k * 256 < n4) v[k] = a[i + k * 256];
#pragma unroll
    for (int k = 0; k < UNROLL; k++)
      if (i + k * 256 < n4) b[i + k * 256] = v[k];
  }
}
// The same copy walked as a raster of 16384-float rows in 1024 x 4 patches, one per workgroup: a workgroup's
// four 4-KiB pieces sit 64 KiB apart, which spreads every workgroup over more HBM channels than 16 contiguous
// KiB do -- measured 6.1 TB/s against 5.4 TB/s for the linear form on the same device.  The better of the two
// is the practical ceiling.
__global__ __launch_bounds__(256) void k_membench_patch(const float *__restrict__ a, float *__restrict__ b,
                                                       int patches_x) {
  constexpr int RL = 16384, PW = 1024, PH = 4;
  const int py = blockIdx.x / patches_x, px = blockIdx.x - py * patches_x;
  float4 v[PH];
#pragma unroll
  for (int r = 0; r < PH; r++)
    v[r] = *reinterpret_cast<const float4 *>(a + (long long)(py * PH + r) * RL + px * PW + threadIdx.x * 4);
#pragma unroll
  for (int r = 0; r < PH; r++)
    *reinterpret_cast<float4 *>(b + (long long)(py * PH + r) * RL + px * PW + threadIdx.x * 4) = v[r];
}
// NR read streams summed into NW write streams, the same 1024 x 4 patches: what the memory system gives a given
// read / write mix with no arithmetic in the way (the fused slope + TI + MTI stencil is 2 reads + 3 writes)
typedef float mb_v4f __attribute__((ext_vector_type(4)));
template <int NR, int NW, bool NT>
__global__ __launch_bounds__(256) void k_membench_mix(const float *__restrict__ r0, const float *__restrict__ r1,
                                                     float *__restrict__ w0, float *__restrict__ w1,
                                                     float *__restrict__ w2, int patches_x) {
  constexpr int RL = 16384, PW = 1024, PH = 4;
  const int py = blockIdx.x / patches_x, px = blockIdx.x - py * patches_x;
  mb_v4f v[PH];
#pragma unroll
  for (int r = 0; r < PH; r++) {
    const long long o = (long long)(py * PH + r) * RL + px * PW + threadIdx.x * 4;
    mb_v4f a = {1.0f, 2.0f, 3.0f, 4.0f};
    if (NR >= 1) a = NT ? __builtin_nontemporal_load(reinterpret_cast<const mb_v4f *>(r0 + o)) : *reinterpret_cast<const mb_v4f *>(r0 + o);
    if (NR >= 2) a += NT ? __builtin_nontemporal_load(reinterpret_cast<const mb_v4f *>(r1 + o)) : *reinterpret_cast<const mb_v4f *>(r1 + o);
    v[r] = a;
  }
#pragma unroll
  for (int r = 0; r < PH; r++) {
    const long long o = (long long)(py * PH + r) * RL + px * PW + threadIdx.x * 4;
    float *ws[3] = {w0, w1, w2};
#pragma unroll
    for (int k = 0; k < NW; k++) {
      if (NT) __builtin_nontemporal_store(v[r], reinterpret_cast<mb_v4f *>(ws[k] + o));
      else *reinterpret_cast<mb_v4f *>(ws[k] + o) = v[r];
    }
  }
}
int dt_launch_membench_mix(hipStream_t s, const float *r0, const float *r1, float *w0, float *w1, float *w2, int64_t n,
                           int nr, int nw, int nt) {
  DT_REQUIRE(n % (16384 * 4) == 0 && n > 0, "the mix benchmark needs a multiple of 4 rows of 16384 floats");
  DT_REQUIRE(nr >= 0 && nr <= 2 && nw >= 1 && nw <= 3, "0-2 read streams, 1-3 write streams");
  dim3 g((unsigned)(n / 4096)), b(256);
#define MB_GO(R, W)                                                                                        \
  if (nr == R && nw == W) {                                                                                \
    if (nt) hipLaunchKernelGGL((k_membench_mix<R, W, true>), g, b, 0, s, r0, r1, w0, w1, w2, 16);          \
    else hipLaunchKernelGGL((k_membench_mix<R, W, false>), g, b, 0, s, r0, r1, w0, w1, w2, 16);            \
  }
  MB_GO(0, 1) MB_GO(0, 2) MB_GO(0, 3) MB_GO(1, 1) MB_GO(1, 2) MB_GO(1, 3) MB_GO(2, 1) MB_GO(2, 2) MB_GO(2, 3)
#undef MB_GO
  return DT_OK;
}

int dt_launch_membench_copy(hipStream_t s, const float *a, float *b, int64_t n, int blocks) {
  int64_t n4 = n / 4;
  if (n4 == 0) return DT_OK;
  if (blocks < 0) {  // patch form; needs whole rows of 16384 floats in groups of 4
    DT_REQUIRE(n % (16384 * 4) == 0, "patch copy needs a multiple of 4 rows of 16384 floats");
    hipLaunchKernelGGL(k_membench_patch, dim3((unsigned)(n / 4096)), dim3(256), 0, s, a, b, 16);
    return DT_OK;
  }
  hipLaunchKernelGGL(k_membench_copy<4>, dim3((unsigned)blocks), dim3(256), 0, s, (const float4 *)a, (float4 *)b, n4);
  return DT_OK;
}

// ---- dtype helpers ----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_i32_to_i64(const int32_t *__restrict__ a, int64_t n,
                                                   int64_t *__restrict__ b) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) b[i] = a[i];
}
__global__ __launch_bounds__(256) void k_i64_to_i32(const int64_t *__restrict__ a, int64_t n,
                                                   int32_t *__restrict__ b) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    int64_t v = a[i];
    b[i] = v > 2147483647ll ? 2147483647 : (v < -2147483647ll - 1 ? (int32_t)(-2147483647 - 1) : (int32_t)v);
  }
}
int dt_launch_i32_to_i64(hipStream_t s, const int32_t *a, int64_t n, int64_t *b) {
  if (n) hipLaunchKernelGGL(k_i32_to_i64, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a, n, b);
  return DT_OK;
}
int dt_launch_i64_to_i32(hipStream_t s, const int64_t *a, int64_t n, int32_t *b) {
  if (n) hipLaunchKernelGGL(k_i64_to_i32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a, n, b);
  return DT_OK;
}

