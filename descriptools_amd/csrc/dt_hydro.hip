// dt_hydro.hip -- hydrological conditioning for the net-new D8 kernel: depression filling and
// flat resolution, so that D8 -> flow accumulation works on unconditioned real DEMs.  The reference has no
// counterpart (its `fdr` comes from a GIS tool, Example/example.py:36); definitions:
//
//   fill      W(c) = min over paths from c to an OUTLET of the highest cell on the path (the "priority-flood"
//             surface).  Outlets: valid cells on the raster edge or next to a nodata cell.  Computed as the
//             greatest fixed point of  W(c) = max(z(c), min over the 8 neighbours of W)  below the start
//             W = z on outlets, +inf elsewhere (Planchon & Darboux); only max / min of float32 heights, so the
//             result is exact and identical to a sequential priority flood, whatever the update order.
//   D8 on W   k_stencil on the filled surface: every cell with a strictly lower neighbour gets its code.
//   flats     the remaining valid cells (no lower neighbour on W): a cell next to nodata drains into the first
//             nodata neighbour in scan order; the others get the hop distance, through 8-connected cells of the
//             SAME filled height, to the nearest cell that already has a code, and point at a neighbour that is one
//             hop closer: the first of N, W, E, S, else the first of NW, NE, SW, SE (cardinal steps first, as the
//             steepest descent ranks equal drops; on the bundled Example this reproduces 88 % of the GIS tool's
//             codes on flats, scan order alone 16 %).  Distances strictly decrease along the directions: no
//             cycles, every flat cell reaches a coded cell.
//
// Both fixed points are iterated tile by tile: a 256-thread workgroup keeps a 64 x 64 tile with a one-cell halo in
// LDS and relaxes it with four directional in-place sweeps, so a global round moves information a whole tile at a time;
// rounds are launched in small batches with one device flag read back per batch.  Single raster only (one rank).
#include "dt_common.h"
#include "dt_kernels.h"

#define HT 64
#define HLD (HT + 2)
// LDS row strides.  Two of a tile's four sweeps walk COLUMNS with a lane per row: with the window's natural stride of
// 66 words (= 2 mod 32 banks) the 64 lanes of such a wave fall on 16 banks, four deep, and with the 64-word stride of
// the tile's own heights on ONE bank, sixty-four deep -- round 3's relaxation rounds spent most of their time in
// those two reads (rough 16384^2: 3.9-4.4 ms for a round that visits every tile).  Odd strides put the lanes of a
// column sweep on all 32 banks; the row sweeps (a lane per column) do not care.
#define HLS (HLD + 1) /* 67 */
#define HZS (HT + 1)  /* 65 */
#define H_CPT (HT * HT / 256)
#define H_INF_DIST 0x7FFFFFFFu

__device__ __forceinline__ bool hy_nodata(float z) { return z == DT_NODATA; }
// v of the lane before / after mine; `edge` for lane 0 / 63 (wave_shr:1 / wave_shl:1 keep the old value there)
__device__ __forceinline__ float hy_from_prev_lane(float edge, float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(edge), __float_as_int(v), 0x138, 0xF, 0xF, false));
}
__device__ __forceinline__ float hy_from_next_lane(float edge, float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(edge), __float_as_int(v), 0x130, 0xF, 0xF, false));
}

// min / max as the bare instructions.  fminf / fmaxf cost a canonicalising `v_max x, x, x` per operand (IEEE mode: a
// signalling NaN must come out quiet) -- seven of them in a step of the fill sweep, which is bound by vector issue.
// The surface never holds a NaN (heights are staged through comparisons, +inf stands for "unknown"), and for a quiet
// NaN height the instructions return the other operand, as fminf / fmaxf do.
__device__ __forceinline__ float hy_min2(float a, float b) {
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float hy_min3(float a, float b, float c) {
  float r;
  asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ float hy_max2(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// One directional in-place sweep of the fill over the tile in LDS (see k_fill_relax): BY_ROWS: a lane per column, the
// sweep walks rows; DIR: +1 from the first line to the last, -1 back.  Direction and strides are template parameters
// so that every LDS address of a step is the running position plus an immediate offset (with run-time strides a step
// spent five instructions on address arithmetic).  Returns bit 0: the sweep lowered something; bit 1: on the tile's
// outer ring -- the cells the neighbouring tiles read as their halo.
template <bool BY_ROWS, int DIR>
__device__ __forceinline__ int hy_fill_sweep(float *__restrict__ s_w, const float *__restrict__ s_z, int lane) {
  constexpr int SA = BY_ROWS ? DIR * HLS : DIR;  // one step along the sweep
  constexpr int SC = BY_ROWS ? 1 : HLS;          // one lane across it
  constexpr int ZA = BY_ROWS ? DIR * HZS : DIR;
  constexpr int K0 = DIR < 0 ? HT - 1 : 0;
  int p = BY_ROWS ? (K0 + 1) * HLS + lane + 1 : (lane + 1) * HLS + K0 + 1;
  int zi = BY_ROWS ? K0 * HZS + lane : lane * HZS + K0;
  float up = s_w[p - SA];                              // the line before the tile (halo: nobody writes it)
  float hl = s_w[p - SA - SC], hr = s_w[p - SA + SC];  // its cells beside lanes 0 / 63
  float cur = s_w[p], lf = s_w[p - SC], rt = s_w[p + SC];
  const bool on_side = lane == 0 || lane == HT - 1;
  auto *s_wv = (__attribute__((address_space(3))) const float *)s_w;
  asm volatile("" : "+v"(s_wv));
  int ch = 0;
#pragma unroll 4
  for (int step = 0; step < HT; step++) {
    // (through a copy of the base address the compiler cannot see through, made once per sweep: another wave may have
    // lowered the cell since this lane fetched it as the line ahead, and with compile-time strides the compiler would
    // reuse that value -- a store could then RAISE the cell)
    const float fresh = s_wv[p];
    const float d0 = s_w[p + SA - SC], d1 = s_w[p + SA], d2 = s_w[p + SA + SC];
    const float zc = s_z[zi];
    const float upm = hy_from_prev_lane(hl, up), upp = hy_from_next_lane(hr, up);
    const float m = hy_min2(hy_min3(hy_min3(upm, up, upp), lf, rt), hy_min3(d0, d1, d2));
    const float nw = hy_max2(zc, m);
    const bool valid = !hy_nodata(zc);
    if (valid && nw < fresh) {  // (a value equal to its height cannot get lower: nw >= zc)
      s_w[p] = nw;
      ch |= (on_side || step == 0 || step == HT - 1) ? 3 : 1;  // bit 1: a cell of the tile's outer ring
    }
    up = (valid && nw < cur) ? nw : cur;  // what this lane leaves behind, without waiting for `fresh`
    hl = lf;   // the next step's line before is this line: beside lanes 0 / 63 lie its halo cells
    hr = rt;
    cur = d1;
    lf = d0;
    rt = d2;
    p += SA;
    zi += ZA;
  }
  return ch;
}

// W = z on outlets (edge of the GLOBAL raster / next to nodata), +inf on the other valid cells, nodata stays nodata.
// Windowed like every tile kernel: the core of `w`, neighbours read from the halo where the core ends inside the raster.
__global__ __launch_bounds__(256) void k_fill_init(const float *__restrict__ dem, DtWin w, float *__restrict__ wout) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)w.H * w.W) return;
  int y = (int)(i / w.W), x = (int)(i - (int64_t)y * w.W);
  const long long o = (long long)y * w.ld + x;
  float z = dem[o];
  if (hy_nodata(z)) {
    wout[o] = DT_NODATA;
    return;
  }
  const int gy = w.gy0 + y, gx = w.gx0 + x;
  bool outlet = gy == 0 || gx == 0 || gy == w.Hg - 1 || gx == w.Wg - 1;
  if (!outlet) {
#pragma unroll
    for (int dy = -1; dy <= 1; dy++)
#pragma unroll
      for (int dx = -1; dx <= 1; dx++)
        if ((dy || dx) && hy_nodata(dem[(long long)(y + dy) * w.ld + x + dx])) outlet = true;
  }
  wout[o] = outlet ? z : __builtin_inff();
}

// ---- staging a tile's 66 x 66 window into LDS ---------------------------------------------------------------------
// The common case -- a tile whose window lies inside the raster -- takes the 64 x 64 core as 16-byte loads (four
// per thread) and the ring around it as one or two 4-byte loads: six load instructions per thread where the 66-wide
// rows, which start one cell before a 256-byte boundary, took eighteen (bare reads: 6.3 instead of 3.8 TB/s,
// profiles/r4/micro_tile_read.txt).  Loads and LDS stores are separate functions so that a kernel that stages two
// rasters (or a raster and the tile's heights) has ALL its loads in flight before it waits for the first.
typedef uint32_t hy_v4u __attribute__((ext_vector_type(4)));
// the same for rows that are only 4-byte aligned (a raster whose width is not a multiple of 4: the Example's 1534):
// global memory takes a 16-byte access at any 4-byte address
typedef uint32_t hy_v4u_a4 __attribute__((ext_vector_type(4), aligned(4)));
typedef float hy_v4f_a4 __attribute__((ext_vector_type(4), aligned(4)));
struct HyTileRegs {
  hy_v4u c[4];
  uint32_t r1, r2;
};
__device__ __forceinline__ void hy_ring_cell(int j, int &r, int &c) {  // 66 above, 66 below, 64 left, 64 right
  if (j < HLD) { r = 0; c = j; }
  else if (j < 2 * HLD) { r = HLD - 1; c = j - HLD; }
  else if (j < 2 * HLD + HT) { r = j - 2 * HLD + 1; c = 0; }
  else { r = j - 2 * HLD - HT + 1; c = HLD - 1; }
}
// block-uniform: the whole window is readable (the core's rows are then read 16 bytes at a time)
__device__ __forceinline__ bool hy_tile_fast(const void *src, const DtWin &w, int y0, int x0) {
  return dt_readable(w, y0 - 1, x0 - 1) && dt_readable(w, y0 + HT, x0 + HT) && ((uintptr_t)src & 3) == 0;
}
__device__ __forceinline__ void hy_tile_load(HyTileRegs &t, const void *src, const DtWin &w, int y0, int x0) {
  const uint32_t *__restrict__ p32 = reinterpret_cast<const uint32_t *>(src);
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int i = (int)threadIdx.x + 256 * k;  // 16 groups of four cells per row
    t.c[k] = *reinterpret_cast<const hy_v4u_a4 *>(p32 + (long long)(y0 + (i >> 4)) * w.ld + x0 + (i & 15) * 4);
  }
  int r, c;
  hy_ring_cell((int)threadIdx.x, r, c);
  t.r1 = p32[(long long)(y0 - 1 + r) * w.ld + x0 - 1 + c];
  t.r2 = 0u;
  if ((int)threadIdx.x + 256 < 2 * HLD + 2 * HT) {
    hy_ring_cell((int)threadIdx.x + 256, r, c);
    t.r2 = p32[(long long)(y0 - 1 + r) * w.ld + x0 - 1 + c];
  }
}
// NODATA_INF: a nodata height is stored as +inf (the fill's surface: nodata never lowers a minimum)
template <bool NODATA_INF = false>
__device__ __forceinline__ void hy_tile_store(const HyTileRegs &t, void *s) {
  uint32_t *s32 = reinterpret_cast<uint32_t *>(s);
  const uint32_t nod = __float_as_uint(DT_NODATA), inf = __float_as_uint(__builtin_inff());
  auto f = [&](uint32_t v) { return (NODATA_INF && v == nod) ? inf : v; };  // (-100.0f has one bit pattern)
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int i = (int)threadIdx.x + 256 * k;
    uint32_t *d = s32 + ((i >> 4) + 1) * HLS + 1 + (i & 15) * 4;
    d[0] = f(t.c[k].x);
    d[1] = f(t.c[k].y);
    d[2] = f(t.c[k].z);
    d[3] = f(t.c[k].w);
  }
  int r, c;
  hy_ring_cell((int)threadIdx.x, r, c);
  s32[r * HLS + c] = f(t.r1);
  if ((int)threadIdx.x + 256 < 2 * HLD + 2 * HT) {
    hy_ring_cell((int)threadIdx.x + 256, r, c);
    s32[r * HLS + c] = f(t.r2);
  }
}
// the general form: cells outside the raster (or the rank's memory) read as `outside`
template <typename T>
__device__ __forceinline__ void hy_stage_slow(T *s, const T *__restrict__ src, const DtWin &w, int y0, int x0, T outside) {
  // all of a thread's (up to 18) loads are issued before the first is used: one memory round trip per staging, not
  // eighteen (the loop form waited for each load before it computed the next address)
  constexpr int N = (HLD * HLD + 255) / 256;
  T v[N];
  // block-uniform: the whole 66 x 66 window is readable (rows that are not 16-byte aligned bring a tile here too)
  const bool all_in = dt_readable(w, y0 - 1, x0 - 1) && dt_readable(w, y0 + HT, x0 + HT);
#pragma unroll
  for (int k = 0; k < N; k++) {
    const int i = threadIdx.x + 256 * k;
    v[k] = outside;
    if (i < HLD * HLD) {
      const int r = i / HLD, c = i - r * HLD;
      const int y = y0 - 1 + r, x = x0 - 1 + c;
      // inside the global raster and in this rank's memory (its core or its halo); a tile of a ragged last row /
      // column reaches beyond the core into the halo, which is fine: those cells are read, never written
      if (all_in || dt_readable(w, y, x)) v[k] = src[(long long)y * w.ld + x];
    }
  }
#pragma unroll
  for (int k = 0; k < N; k++) {
    const int i = threadIdx.x + 256 * k;
    if (i < HLD * HLD) s[(i / HLD) * HLS + (i - (i / HLD) * HLD)] = v[k];
  }
}
// stage the tile's 66 x 66 window of `src` into LDS; cells outside the raster read as `outside`
template <typename T>
__device__ __forceinline__ void hy_stage(T *s, const T *__restrict__ src, const DtWin &w, int y0, int x0, T outside) {
  static_assert(sizeof(T) == 4, "32-bit rasters");
  if (hy_tile_fast(src, w, y0, x0)) {
    HyTileRegs t;
    hy_tile_load(t, src, w, y0, x0);
    hy_tile_store(t, s);
  } else {
    hy_stage_slow<T>(s, src, w, y0, x0, outside);
  }
}
// ... of two rasters: both windows' loads are in flight before the first store waits
template <typename TA, typename TB>
__device__ __forceinline__ void hy_stage2(TA *sa, const TA *__restrict__ a, TA outside_a, TB *sb,
                                          const TB *__restrict__ b, TB outside_b, const DtWin &w, int y0, int x0) {
  if (hy_tile_fast(a, w, y0, x0) && hy_tile_fast(b, w, y0, x0)) {
    HyTileRegs ta, tb;
    hy_tile_load(ta, a, w, y0, x0);
    hy_tile_load(tb, b, w, y0, x0);
    hy_tile_store(ta, sa);
    hy_tile_store(tb, sb);
  } else {
    hy_stage_slow<TA>(sa, a, w, y0, x0, outside_a);
    hy_stage_slow<TB>(sb, b, w, y0, x0, outside_b);
  }
}

// Which tiles a round has to visit.  A visit makes up to `sweeps` rounds of four directional sweeps over its tile and
// leaves one byte per tile: HY_CHANGED -- its cells changed, the eight tiles around it have a new halo to look at --
// and HY_OPEN -- its last round of sweeps still moved something, so it must be visited again whatever its neighbours
// do.  A round visits the tiles that have a changed NEIGHBOUR or are open themselves; a tile at its local fixed point
// rests until its halo changes.  (With one round of sweeps per visit -- the measured optimum, HY_FILL_SWEEPS -- a
// tile that changed is open: round 3's rule.)  act_prev == NULL: the first round of a phase, every tile is visited.
#define HY_CHANGED 1 /* (round 4, late: only when a cell of the tile's OUTER RING changed -- what its neighbours read) */
#define HY_OPEN 2
// rounds of sweeps per visit at most (dt_debug_set(6 / 7, n) overrides: tools/condition_bench.py N sweeps).  Measured,
// round 4 (profiles/r4/conditioning_sweeps.txt): ONE is best for both relaxations on rough 8192^2 terrain and on the
// Example (6.8 / 1.08 ms against 8.2 / 1.28 with up to six flat sweeps, 7.4 / 1.14 with two fill sweeps) -- what bounds
// the number of global rounds is how many tiles a depression or a flat spans, not how far a tile is from its local
// fixed point, so sweeping a tile to that point only repeats work the next visit does anyway.
#define HY_FILL_SWEEPS 1
#define HY_FLAT_SWEEPS 1
// COLOURED rounds (round 4).  A round that visits all tiles at once reads, in every tile, what the neighbours held
// BEFORE the round: a front moves one tile per round.  The tiles are coloured 2 x 2 (colour = 2 * (row & 1) + (column &
// 1): all eight neighbours of a tile have other colours) and a round is four launches, one colour each: a tile sees what
// the colours before it did in this very round, and a front that crosses tile borders moves two tiles per round for the
// same number of tile visits -- rough 16384^2 terrain: 16 -> 9 fill rounds, 12 -> 7 flat rounds.  The activity flags
// live in ONE array used in place: when a tile of colour c is visited, every neighbour's latest visit lies after the
// tile's own previous one, so the flags it reads are exactly the changes it has not seen yet, and nobody writes them
// during this launch.  colour < 0: every tile (the first round of the fill, which initialises the surface).
__device__ __forceinline__ void hy_tile_of_block(int colour, int tiles_x, int &ty, int &tx) {
  if (colour < 0) {
    ty = (int)blockIdx.x / tiles_x;
    tx = (int)blockIdx.x - ty * tiles_x;
  } else {
    const int cx = (tiles_x - (colour & 1) + 1) >> 1;  // tiles of this colour in a row of tiles
    const int i = (int)blockIdx.x / cx, j = (int)blockIdx.x - i * cx;
    ty = 2 * i + (colour >> 1);
    tx = 2 * j + (colour & 1);
  }
}
// workgroups of a launch over the tiles of one colour
static unsigned hy_colour_blocks(int colour, int tiles_x, int tiles_y) {
  if (colour < 0) return (unsigned)(tiles_x * tiles_y);
  return (unsigned)(((tiles_x - (colour & 1) + 1) >> 1) * ((tiles_y - (colour >> 1) + 1) >> 1));
}
__device__ __forceinline__ bool hy_tile_active(const uint8_t *__restrict__ act_prev, int ty, int tx, int tiles_x,
                                               int tiles_y) {
  if (!act_prev) return true;
  int v = 0;
  if (threadIdx.x < 9) {
    const int y = ty + (int)threadIdx.x / 3 - 1, x = tx + (int)threadIdx.x % 3 - 1;
    if (y >= 0 && y < tiles_y && x >= 0 && x < tiles_x)
      v = act_prev[(size_t)y * tiles_x + x] & (threadIdx.x == 4 ? HY_OPEN : HY_CHANGED);
  }
  return __syncthreads_or(v) != 0;
}

// one round of the fill: one round of four directional sweeps on every tile that has to be visited
// `prev` (may be NULL): the previous round's flag -- a round that follows a quiet one has nothing to do and
// returns at once (the asynchronous form enqueues a fixed budget of rounds and never asks the host)
// INIT: the first round of a single raster's fill starts from the heights themselves -- W = z on the outlets (edge of
// the raster / next to nodata), +inf elsewhere, the other tiles' cells taken as +inf (an upper bound, like every
// intermediate value of this relaxation) -- and writes every cell: no separate initialisation pass over the raster.
template <bool INIT>
__global__ __launch_bounds__(256) void k_fill_relax(const float *__restrict__ dem, float *__restrict__ wsurf, DtWin w,
                                                   int tiles_x, int *__restrict__ changed,
                                                   const int *__restrict__ prev,
                                                   const uint8_t *__restrict__ act_prev,
                                                   uint8_t *__restrict__ act_cur, int tiles_y, int sweeps,
                                                   int colour) {
  const int H = w.H, W = w.W;
  __shared__ float s_w[HLD * HLS];
  __shared__ float s_z[HT * HZS];  // the tile's own heights (nodata beyond the raster)
  // (the previous round's flag was written by the previous KERNEL: a plain, scalar load sees it.  Round 3 read it with
  // a device-scope atomic load in every thread -- 260 K wave-level requests on one address at 16384^2, served one
  // after the other by the memory system: the rounds that had a `prev` took 3.9-4.4 ms, the ones without 1.5)
  if (prev && *prev == 0) return;
  int ty, tx;
  hy_tile_of_block(colour, tiles_x, ty, tx);
  const int tile = ty * tiles_x + tx;
  const int y0 = ty * HT, x0 = tx * HT;
  if (!hy_tile_active(act_prev, ty, tx, tiles_x, tiles_y)) {
    if (act_cur && threadIdx.x == 0) act_cur[tile] = 0;
    return;
  }
  // outside the raster and nodata both read as +inf: they never lower a minimum (cells next to them are outlets
  // and already hold their final value)
  float z[H_CPT];
  // block-uniform: a whole tile -- heights in and surface out go 16 bytes at a time
  const bool fastio = y0 + HT <= H && x0 + HT <= W;
  if (INIT) {
    // the HEIGHTS of the window: a cell is an outlet when it lies on the raster's edge or has a nodata neighbour
    // (cells beyond the raster are staged as +inf: not nodata)
    hy_stage<float>(s_w, dem, w, y0, x0, __builtin_inff());
    __syncthreads();
    float w0[H_CPT];
#pragma unroll
    for (int j = 0; j < H_CPT; j++) {
      const int c = threadIdx.x + 256 * j;
      const int ly = c / HT, lx = c % HT;
      const int y = y0 + ly, x = x0 + lx;
      const int p = (ly + 1) * HLS + lx + 1;
      z[j] = (y < H && x < W) ? s_w[p] : DT_NODATA;
      const int gy = w.gy0 + y, gx = w.gx0 + x;
      bool outlet = gy == 0 || gx == 0 || gy == w.Hg - 1 || gx == w.Wg - 1;
      outlet = outlet || hy_nodata(s_w[p - HLS - 1]) || hy_nodata(s_w[p - HLS]) || hy_nodata(s_w[p - HLS + 1]) ||
               hy_nodata(s_w[p - 1]) || hy_nodata(s_w[p + 1]) || hy_nodata(s_w[p + HLS - 1]) ||
               hy_nodata(s_w[p + HLS]) || hy_nodata(s_w[p + HLS + 1]);
      w0[j] = (!hy_nodata(z[j]) && outlet) ? z[j] : __builtin_inff();
    }
    __syncthreads();
    for (int i = threadIdx.x; i < HLD * HLS; i += 256) s_w[i] = __builtin_inff();  // the other tiles' cells: unknown yet
    __syncthreads();
#pragma unroll
    for (int j = 0; j < H_CPT; j++) {
      const int c = threadIdx.x + 256 * j;
      s_w[(c / HT + 1) * HLS + (c % HT) + 1] = w0[j];
    }
  } else {
    // the tile's heights first: their loads are in flight while the surface is staged (they were issued after its
    // barrier before: a second memory round trip on every visit)
    hy_v4u z4[4];
    if (fastio) {
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int i = (int)threadIdx.x + 256 * k;
        z4[k] = *reinterpret_cast<const hy_v4u_a4 *>(dem + (long long)(y0 + (i >> 4)) * w.ld + x0 + (i & 15) * 4);
      }
    } else {
#pragma unroll
      for (int j = 0; j < H_CPT; j++) {
        int c = threadIdx.x + 256 * j;
        int ly = c / HT, lx = c % HT;
        int y = y0 + ly, x = x0 + lx;
        z[j] = (y < H && x < W) ? dem[(long long)y * w.ld + x] : DT_NODATA;
      }
    }
    const bool fast_w = hy_tile_fast(wsurf, w, y0, x0);
    if (fast_w) {  // nodata -> +inf on the way into LDS: no pass over the LDS image and no barrier for it
      HyTileRegs t;
      hy_tile_load(t, wsurf, w, y0, x0);
      hy_tile_store<true>(t, s_w);
    } else {
      hy_stage_slow<float>(s_w, wsurf, w, y0, x0, __builtin_inff());
    }
    if (fastio) {
      uint32_t *sz32 = reinterpret_cast<uint32_t *>(s_z);
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int i = (int)threadIdx.x + 256 * k;
        uint32_t *d = sz32 + (i >> 4) * HZS + (i & 15) * 4;
        d[0] = z4[k].x;
        d[1] = z4[k].y;
        d[2] = z4[k].z;
        d[3] = z4[k].w;
      }
    }
    if (!fast_w) {
      __syncthreads();
      for (int i = threadIdx.x; i < HLD * HLS; i += 256)
        if (hy_nodata(s_w[i])) s_w[i] = __builtin_inff();  // (the pad column holds garbage nobody reads)
    }
  }
  if (INIT || !fastio) {
#pragma unroll
    for (int j = 0; j < H_CPT; j++) {
      const int c = threadIdx.x + 256 * j;
      s_z[(c / HT) * HZS + (c % HT)] = z[j];
    }
  }
  __syncthreads();
  // DIRECTIONAL in-place sweeps (round 3; Jacobi sweeps to the tile's local fixed point before: one cell of progress
  // per sweep and barrier, ~100 of them for a front crossing the tile).  Each of the four waves walks the whole tile
  // in its own direction -- wave 0 top to bottom, 1 bottom to top (a lane per column), 2 left to right, 3 right to
  // left (a lane per row) -- 64 steps in lockstep, every step the full 8-neighbour relaxation in place: what a
  // step lowers is seen by the steps after it, so a front crosses the tile in one sweep along a monotone path and in a
  // few sweeps along a winding one.  The four sweeps run concurrently on the same LDS image; the operator is
  // monotone (values only decrease, towards the same greatest fixed point), so any interleaving and any stale read are
  // harmless, and a round of four sweeps that lowers nothing proves the fixed point.
  // (Round 3 made ONE such round of four sweeps per visit, because a tile that had changed was visited again in the
  // next global round anyway; with the HY_CHANGED / HY_OPEN rule a visit iterates to the tile's local fixed point.)
  // A step's dependent chain is short: the line the sweep comes from is in REGISTERS (a lane's own previous result,
  // its two neighbours' through DPP; the halo cells beside the edge lanes from LDS), the line it stands on and the
  // line ahead were fetched a step earlier (the line ahead of this step IS the next step's line), and the one fresh
  // read of the cell itself -- another wave may have lowered it since -- only gates the store.  ~15 vector
  // instructions per step instead of two LDS round trips.
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  int any = 0, open = 0, ring = 0;
  // rounds of four sweeps until one of them lowers nothing (round 4; `sweeps` = 1: one round per visit)
  for (int it = 0; it < sweeps; it++) {
    int ch;
    if (wave == 0) ch = hy_fill_sweep<true, 1>(s_w, s_z, lane);         // top to bottom
    else if (wave == 1) ch = hy_fill_sweep<true, -1>(s_w, s_z, lane);   // bottom to top
    else if (wave == 2) ch = hy_fill_sweep<false, 1>(s_w, s_z, lane);   // left to right (a lane per row)
    else ch = hy_fill_sweep<false, -1>(s_w, s_z, lane);                 // right to left
    open = __syncthreads_or(ch);
    if (!open) break;
    any = 1;
    ring |= __syncthreads_or(ch & 2);
  }
  if (INIT) any = ring = 1;  // every cell is written, and the neighbours have yet to see this tile
  if (act_cur && threadIdx.x == 0) act_cur[tile] = (uint8_t)((ring ? HY_CHANGED : 0) | (open ? HY_OPEN : 0));
  if (!any) return;
  if (fastio) {
    // (a nodata cell holds DT_NODATA since the first round: writing it again changes nothing)
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int i = (int)threadIdx.x + 256 * k;
      const int r = i >> 4, c4 = (i & 15) * 4;
      const float *sw = s_w + (r + 1) * HLS + 1 + c4, *sz = s_z + r * HZS + c4;
      hy_v4f_a4 v;
      v.x = hy_nodata(sz[0]) ? DT_NODATA : sw[0];
      v.y = hy_nodata(sz[1]) ? DT_NODATA : sw[1];
      v.z = hy_nodata(sz[2]) ? DT_NODATA : sw[2];
      v.w = hy_nodata(sz[3]) ? DT_NODATA : sw[3];
      *reinterpret_cast<hy_v4f_a4 *>(wsurf + (long long)(y0 + r) * w.ld + x0 + c4) = v;
    }
  } else {
#pragma unroll
    for (int j = 0; j < H_CPT; j++) {
      int c = threadIdx.x + 256 * j;
      int y = y0 + c / HT, x = x0 + c % HT;
      if (y < H && x < W) {
        if (!hy_nodata(z[j])) wsurf[(long long)y * w.ld + x] = s_w[(c / HT + 1) * HLS + (c % HT) + 1];
        else if (INIT) wsurf[(long long)y * w.ld + x] = DT_NODATA;
      }
    }
  }
  if (threadIdx.x == 0) atomicOr(changed, 1);
}

// D8 code towards neighbour k of the scan order NW N NE W E SW S SE
__device__ __forceinline__ uint8_t hy_code_of_scan(int k) {
  const uint8_t codes[8] = {32, 64, 128, 16, 1, 8, 4, 2};
  return codes[k];
}
__device__ __forceinline__ void hy_scan_delta(int k, int &dy, int &dx) {
  const int8_t ddy[8] = {-1, -1, -1, 0, 0, 1, 1, 1}, ddx[8] = {-1, 0, 1, -1, 1, -1, 0, 1};
  dy = ddy[k];
  dx = ddx[k];
}

// flats: dist = 0 for cells that have a code (and for nodata, which nobody asks), "infinite" for valid cells
// without one; a code-less cell next to nodata drains into its first nodata neighbour right away.  One workgroup per
// 64 x 64 tile with the surface staged in LDS (round 4; one thread per cell on global memory before: 2.4 ms at
// 16384^2, up to 9 scattered loads per code-less cell); has_flat[tile] (may be NULL) = the tile has cells that need a
// distance: the relaxation rounds and the assignment never look at the others.
// nsame (may be NULL; single rasters): one byte per cell, bit k (scan order NW N NE W E SW S SE) SET when neighbour k
// does NOT have the cell's height (0xFF for nodata) -- all the relaxation rounds need of the surface: k_flat_relax_m
__global__ __launch_bounds__(256) void k_flat_init(const float *__restrict__ wsurf, uint8_t *__restrict__ fdr, DtWin w,
                                                  uint32_t *__restrict__ dist, int tiles_x,
                                                  uint8_t *__restrict__ has_flat, uint8_t *__restrict__ nsame) {
  __shared__ float s_w[HLD * HLS];
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
  const int y0 = ty * HT, x0 = tx * HT;
  // a code-less valid cell: distance "infinite", unless it lies next to nodata -- then it drains there right away
  // (returns the cell's distance; *code != 0: the cell's new D8 code)
  auto init_cell = [&](int p, uint32_t f, uint32_t *code) -> uint32_t {
    *code = 0u;
    if (hy_nodata(s_w[p]) || f != 0u) return 0u;
    // scan order NW N NE W E SW S SE
    uint32_t c = 0u;
    if (hy_nodata(s_w[p - HLS - 1])) c = 32u;
    else if (hy_nodata(s_w[p - HLS])) c = 64u;
    else if (hy_nodata(s_w[p - HLS + 1])) c = 128u;
    else if (hy_nodata(s_w[p - 1])) c = 16u;
    else if (hy_nodata(s_w[p + 1])) c = 1u;
    else if (hy_nodata(s_w[p + HLS - 1])) c = 8u;
    else if (hy_nodata(s_w[p + HLS])) c = 4u;
    else if (hy_nodata(s_w[p + HLS + 1])) c = 2u;
    *code = c;
    return c ? 0u : H_INF_DIST;
  };
  auto nsame_of = [&](int p) -> uint32_t {
    const float wc = s_w[p];
    if (hy_nodata(wc)) return 0xFFu;
    return (s_w[p - HLS - 1] == wc ? 0u : 1u) | (s_w[p - HLS] == wc ? 0u : 2u) | (s_w[p - HLS + 1] == wc ? 0u : 4u) |
           (s_w[p - 1] == wc ? 0u : 8u) | (s_w[p + 1] == wc ? 0u : 16u) | (s_w[p + HLS - 1] == wc ? 0u : 32u) |
           (s_w[p + HLS] == wc ? 0u : 64u) | (s_w[p + HLS + 1] == wc ? 0u : 128u);
  };
  int any = 0;
  // block-uniform: a whole tile of aligned rows -- four codes per 32-bit load, four distances per 16-byte store
  const bool fast = y0 + HT <= w.H && x0 + HT <= w.W && (w.ld & 3) == 0 && ((uintptr_t)fdr & 3) == 0;
  if (fast) {
    uint32_t f4[4];  // the tile's codes, all loads in flight with the staging's
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int i = (int)threadIdx.x + 256 * k;
      f4[k] = *reinterpret_cast<const uint32_t *>(fdr + (long long)(y0 + (i >> 4)) * w.ld + x0 + (i & 15) * 4);
    }
    hy_stage<float>(s_w, wsurf, w, y0, x0, __builtin_inff());
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int i = (int)threadIdx.x + 256 * k;
      const int r = i >> 4, c4 = (i & 15) * 4;
      const long long o = (long long)(y0 + r) * w.ld + x0 + c4;
      uint32_t d[4];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        uint32_t code;
        d[q] = init_cell((r + 1) * HLS + 1 + c4 + q, (f4[k] >> (8 * q)) & 0xFFu, &code);
        if (code) fdr[o + q] = (uint8_t)code;
        any |= d[q] == H_INF_DIST ? 1 : 0;
      }
      hy_v4u_a4 v = {d[0], d[1], d[2], d[3]};
      *reinterpret_cast<hy_v4u_a4 *>(dist + o) = v;
      if (nsame) {
        const int p = (r + 1) * HLS + 1 + c4;
        *reinterpret_cast<uint32_t *>(nsame + o) = nsame_of(p) | (nsame_of(p + 1) << 8) | (nsame_of(p + 2) << 16) |
                                                   (nsame_of(p + 3) << 24);
      }
    }
  } else {
    uint8_t f[H_CPT];
#pragma unroll
    for (int j = 0; j < H_CPT; j++) {
      const int c = threadIdx.x + 256 * j;
      const int y = y0 + c / HT, x = x0 + c % HT;
      f[j] = (y < w.H && x < w.W) ? fdr[(long long)y * w.ld + x] : (uint8_t)1;
    }
    // (beyond the raster: not nodata -- those cells got their outward code from the stencil --, never equal to anything)
    hy_stage<float>(s_w, wsurf, w, y0, x0, __builtin_inff());
    __syncthreads();
#pragma unroll
    for (int j = 0; j < H_CPT; j++) {
      const int c = threadIdx.x + 256 * j;
      const int ly = c / HT, lx = c % HT;
      const int y = y0 + ly, x = x0 + lx;
      if (y >= w.H || x >= w.W) continue;
      const long long o = (long long)y * w.ld + x;
      uint32_t code;
      const uint32_t d = init_cell((ly + 1) * HLS + lx + 1, f[j], &code);
      if (code) fdr[o] = (uint8_t)code;
      any |= d == H_INF_DIST ? 1 : 0;
      dist[o] = d;
      if (nsame) nsame[o] = (uint8_t)nsame_of((ly + 1) * HLS + lx + 1);
    }
  }
  any = __syncthreads_or(any);
  // (read as the activity byte of "round -1" by the first relaxation round: the tile itself is open, its neighbours
  // have something to look at)
  if (has_flat && threadIdx.x == 0) has_flat[blockIdx.x] = (uint8_t)(any ? (HY_CHANGED | HY_OPEN) : 0);
}

// One directional in-place sweep of the flat distances over the tile in LDS (see k_flat_relax; direction and strides
// are template parameters as in hy_fill_sweep).  Returns hy_fill_sweep's two bits.
template <bool BY_ROWS, int DIR>
__device__ __forceinline__ int hy_flat_sweep(const float *__restrict__ s_w, uint32_t *__restrict__ s_d, int lane) {
  constexpr int SA = BY_ROWS ? DIR * HLS : DIR, SC = BY_ROWS ? 1 : HLS;
  constexpr int K0 = DIR < 0 ? HT - 1 : 0;
  auto dpp_prev = [](uint32_t edge, uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)edge, (int)v, 0x138, 0xF, 0xF, false);
  };
  auto dpp_next = [](uint32_t edge, uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)edge, (int)v, 0x130, 0xF, 0xF, false);
  };
#define HY_M(wn, dn) m = min(m, (wn) == wc ? (dn) : H_INF_DIST);
  int p = BY_ROWS ? (K0 + 1) * HLS + lane + 1 : (lane + 1) * HLS + K0 + 1;
  float wu = s_w[p - SA], whl = s_w[p - SA - SC], whr = s_w[p - SA + SC];   // heights: line before, its edge cells
  uint32_t du = s_d[p - SA], dhl = s_d[p - SA - SC], dhr = s_d[p - SA + SC];
  float wcur = s_w[p], wlf = s_w[p - SC], wrt = s_w[p + SC];
  uint32_t dcur = s_d[p], dlf = s_d[p - SC], drt = s_d[p + SC];
  const bool on_side = lane == 0 || lane == HT - 1;
  int ch = 0;
#pragma unroll 4
  for (int step = 0; step < HT; step++) {
    int pf = p;  // (see hy_fill_sweep: the fresh read must not be merged with the value fetched a step earlier)
    asm volatile("" : "+v"(pf));
    const uint32_t fresh = s_d[pf];
    const float w0 = s_w[p + SA - SC], w1 = s_w[p + SA], w2 = s_w[p + SA + SC];
    const uint32_t d0 = s_d[p + SA - SC], d1 = s_d[p + SA], d2 = s_d[p + SA + SC];
    const float wc = wcur;
    const float wum = hy_from_prev_lane(whl, wu), wup = hy_from_next_lane(whr, wu);
    const uint32_t dum = dpp_prev(dhl, du), dup = dpp_next(dhr, du);
    uint32_t m = H_INF_DIST;
    HY_M(wum, dum) HY_M(wu, du) HY_M(wup, dup) HY_M(wlf, dlf) HY_M(wrt, drt) HY_M(w0, d0) HY_M(w1, d1) HY_M(w2, d2)
    // coded cells (0) and cells next to one (1) are final; positions beyond the raster edge are staged as nodata
    const bool lower = dcur > 1u && !hy_nodata(wc) && m != H_INF_DIST;
    const uint32_t nd = m + 1u;
    if (lower && nd < fresh) {
      s_d[p] = nd;
      ch |= (on_side || step == 0 || step == HT - 1) ? 3 : 1;  // bit 1: a cell of the tile's outer ring
    }
    du = (lower && nd < dcur) ? nd : dcur;  // what this lane leaves behind
    wu = wc;
    whl = wlf;
    whr = wrt;
    dhl = dlf;
    dhr = drt;
    wcur = w1;
    wlf = w0;
    wrt = w2;
    dcur = d1;
    dlf = d0;
    drt = d2;
    p += SA;
  }
#undef HY_M
  return ch;
}

// one round of the flat distances: d(c) = 1 + min d(n) over neighbours of the same filled height
__global__ __launch_bounds__(256) void k_flat_relax(const float *__restrict__ wsurf, uint32_t *__restrict__ dist, DtWin w,
                                                   int tiles_x, int *__restrict__ changed,
                                                   const int *__restrict__ prev,
                                                   const uint8_t *__restrict__ act_prev,
                                                   uint8_t *__restrict__ act_cur, int tiles_y, int sweeps,
                                                   int colour) {
  const int H = w.H, W = w.W;
  __shared__ float s_w[HLD * HLS];
  __shared__ uint32_t s_d[HLD * HLS];
  if (prev && *prev == 0) return;  // (a plain scalar load: see k_fill_relax)
  int ty, tx;
  hy_tile_of_block(colour, tiles_x, ty, tx);  // (coloured rounds: see hy_tile_of_block)
  const int tile = ty * tiles_x + tx;
  const int y0 = ty * HT, x0 = tx * HT;
  if (!hy_tile_active(act_prev, ty, tx, tiles_x, tiles_y)) {
    if (act_cur && threadIdx.x == 0) act_cur[tile] = 0;
    return;
  }
  hy_stage2<float, uint32_t>(s_w, wsurf, DT_NODATA, s_d, dist, H_INF_DIST, w, y0, x0);
  __syncthreads();
  // directional in-place sweeps, as k_fill_relax: distances only decrease, towards the same fixed point; rounds of four
  // sweeps until one changes nothing.  The heights never change, and the line a sweep comes from is in registers
  // (heights and distances: the lane's own, its neighbours' through DPP), the line it stands on and the line ahead were
  // fetched a step earlier: 7 LDS reads per step where round 3 made 17.
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  int any = 0, open = 0, ring = 0;
  for (int it = 0; it < sweeps; it++) {
    int ch;
    if (wave == 0) ch = hy_flat_sweep<true, 1>(s_w, s_d, lane);
    else if (wave == 1) ch = hy_flat_sweep<true, -1>(s_w, s_d, lane);
    else if (wave == 2) ch = hy_flat_sweep<false, 1>(s_w, s_d, lane);
    else ch = hy_flat_sweep<false, -1>(s_w, s_d, lane);
    open = __syncthreads_or(ch);
    if (!open) break;
    any = 1;
    ring |= __syncthreads_or(ch & 2);
  }
  if (act_cur && threadIdx.x == 0) act_cur[tile] = (uint8_t)((ring ? HY_CHANGED : 0) | (open ? HY_OPEN : 0));
  if (!any) return;
  if (y0 + HT <= H && x0 + HT <= W) {  // 16 bytes at a time
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int i = (int)threadIdx.x + 256 * k;
      const int r = i >> 4, c4 = (i & 15) * 4;
      const uint32_t *sd = s_d + (r + 1) * HLS + 1 + c4;
      hy_v4u_a4 v = {sd[0], sd[1], sd[2], sd[3]};
      *reinterpret_cast<hy_v4u_a4 *>(dist + (long long)(y0 + r) * w.ld + x0 + c4) = v;
    }
  } else {
#pragma unroll
    for (int j = 0; j < H_CPT; j++) {
      int c = threadIdx.x + 256 * j;
      int y = y0 + c / HT, x = x0 + c % HT;
      if (y < H && x < W) dist[(long long)y * w.ld + x] = s_d[(c / HT + 1) * HLS + (c % HT) + 1];
    }
  }
  if (threadIdx.x == 0) atomicOr(changed, 1);
}

// ---- the flat rounds of a single raster: k_flat_relax without the surface -----------------------------------------
// The heights never change during the flat rounds, and a step uses them only to ask which neighbours have the cell's
// height: k_flat_init leaves that as one byte per cell (`nsame`), and a visit stages the distances (with their halo)
// and the tile's 4 KiB of bytes instead of two 17 KiB windows -- 9 instead of 12 bytes of traffic per cell, and 21.5
// instead of 34.5 KiB of LDS: seven workgroups per CU where the two windows allowed four (a visit is a chain of memory
// round trips: what hides it is workgroups in flight).
#define HMS (HT + 1) /* row stride of the byte tile: odd, so that a sweep with a lane per row spreads over the banks */
template <bool BY_ROWS, int DIR>
__device__ __forceinline__ int hy_flat_sweep_m(const uint8_t *__restrict__ s_m, uint32_t *__restrict__ s_d, int lane) {
  constexpr int SA = BY_ROWS ? DIR * HLS : DIR, SC = BY_ROWS ? 1 : HLS;
  constexpr int MA = BY_ROWS ? DIR * HMS : DIR;
  constexpr int K0 = DIR < 0 ? HT - 1 : 0;
  // which bit of the byte (NW N NE W E SW S SE) speaks of the cell before me in the lane before mine, before me, ...
  constexpr int B_UM = BY_ROWS ? (DIR > 0 ? 0 : 5) : (DIR > 0 ? 0 : 2), B_U = BY_ROWS ? (DIR > 0 ? 1 : 6) : (DIR > 0 ? 3 : 4),
                B_UP = BY_ROWS ? (DIR > 0 ? 2 : 7) : (DIR > 0 ? 5 : 7), B_LF = BY_ROWS ? 3 : 1, B_RT = BY_ROWS ? 4 : 6,
                B_D0 = BY_ROWS ? (DIR > 0 ? 5 : 0) : (DIR > 0 ? 2 : 0), B_D1 = BY_ROWS ? (DIR > 0 ? 6 : 1) : (DIR > 0 ? 4 : 3),
                B_D2 = BY_ROWS ? (DIR > 0 ? 7 : 2) : (DIR > 0 ? 7 : 5);
  auto dpp_prev = [](uint32_t edge, uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)edge, (int)v, 0x138, 0xF, 0xF, false);
  };
  auto dpp_next = [](uint32_t edge, uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)edge, (int)v, 0x130, 0xF, 0xF, false);
  };
  // a neighbour of another height counts as infinitely far: its distance OR H_INF_DIST (all the bits a distance has)
  static_assert(H_INF_DIST == 0x7FFFFFFFu, "the masking below");
  auto far = [](uint32_t mk, int bit, uint32_t d) {
    return d | ((uint32_t)__builtin_amdgcn_sbfe((int)mk, bit, 1) & H_INF_DIST);  // (one v_bfe_i32, one v_and_or_b32)
  };
  int p = BY_ROWS ? (K0 + 1) * HLS + lane + 1 : (lane + 1) * HLS + K0 + 1;
  int mi = BY_ROWS ? K0 * HMS + lane : lane * HMS + K0;
  auto *s_dv = (__attribute__((address_space(3))) const uint32_t *)s_d;
  asm volatile("" : "+v"(s_dv));
  uint32_t du = s_d[p - SA], dhl = s_d[p - SA - SC], dhr = s_d[p - SA + SC];
  uint32_t dcur = s_d[p], dlf = s_d[p - SC], drt = s_d[p + SC];
  const bool on_side = lane == 0 || lane == HT - 1;
  int ch = 0;
#pragma unroll 4
  for (int step = 0; step < HT; step++) {
    // (see hy_fill_sweep: the fresh read must not be merged with the value fetched a step earlier -- it goes through
    // a copy of the base address the compiler cannot see through, made once per sweep)
    const uint32_t fresh = s_dv[p];
    const uint32_t d0 = s_d[p + SA - SC], d1 = s_d[p + SA], d2 = s_d[p + SA + SC];
    const uint32_t mk = s_m[mi];
    const uint32_t dum = dpp_prev(dhl, du), dup = dpp_next(dhr, du);
    // The line the sweep comes from and the two cells beside this one: five of the eight neighbours.  The three of the
    // line ahead hold values this sweep has not touched yet -- the sweep in the opposite direction relaxes against
    // them -- and between them the four sweeps of a visit look at every neighbour (each diagonal twice), so a visit
    // that changes nothing still proves the tile's fixed point.  The kernel is bound by vector issue: 45 -> 36
    // instructions per step.
    uint32_t m = min(min(far(mk, B_UM, dum), far(mk, B_U, du)), min(far(mk, B_UP, dup), far(mk, B_LF, dlf)));
    m = min(m, far(mk, B_RT, drt));
    (void)B_D0;
    (void)B_D1;
    (void)B_D2;
    // No test for "coded cells (0) and cells next to one (1) are final" nor for "no neighbour of its height" (nodata,
    // cells beyond the raster: all bits set): nd = m + 1 is at least 1, so it is never below a distance of 0 or 1, and
    // with m = H_INF_DIST it is 2^31, above every distance -- `nd < fresh` says it all (four instructions of 36).
    const uint32_t nd = m + 1u;
    if (nd < fresh) {
      s_d[p] = nd;
      ch |= (on_side || step == 0 || step == HT - 1) ? 3 : 1;  // bit 1: a cell of the tile's outer ring
    }
    du = min(nd, dcur);  // what this lane leaves behind
    dhl = dlf;
    dhr = drt;
    dcur = d1;
    dlf = d0;
    drt = d2;
    p += SA;
    mi += MA;
  }
  return ch;
}
__global__ __launch_bounds__(256) void k_flat_relax_m(const uint8_t *__restrict__ nsame, uint32_t *__restrict__ dist,
                                                     DtWin w, int tiles_x, int *__restrict__ changed,
                                                     const int *__restrict__ prev,
                                                     const uint8_t *__restrict__ act_prev,
                                                     uint8_t *__restrict__ act_cur, int tiles_y, int sweeps, int colour) {
  const int H = w.H, W = w.W;
  __shared__ uint32_t s_d[HLD * HLS];
  __shared__ uint8_t s_m[HT * HMS + 3];
  if (prev && *prev == 0) return;  // (a plain scalar load: see k_fill_relax)
  int ty, tx;
  hy_tile_of_block(colour, tiles_x, ty, tx);  // (coloured rounds: see hy_tile_of_block)
  const int tile = ty * tiles_x + tx;
  const int y0 = ty * HT, x0 = tx * HT;
  if (!hy_tile_active(act_prev, ty, tx, tiles_x, tiles_y)) {
    if (act_cur && threadIdx.x == 0) act_cur[tile] = 0;
    return;
  }
  // the tile's bytes: four cells per thread and load, all in flight with the distances'
  const bool whole = y0 + HT <= H && x0 + HT <= W;
  uint32_t mk4[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int i = (int)threadIdx.x + 256 * k;
    const int r = i >> 4, c4 = (i & 15) * 4;
    const uint8_t *src = nsame + (long long)(y0 + r) * w.ld + x0 + c4;
    if (whole && (w.ld & 3) == 0 && ((uintptr_t)nsame & 3) == 0) {
      mk4[k] = *reinterpret_cast<const uint32_t *>(src);
    } else {
      mk4[k] = 0xFFFFFFFFu;  // beyond the raster: no neighbour of "their height"
      if (y0 + r < H) {
#pragma unroll
        for (int q = 0; q < 4; q++)
          if (x0 + c4 + q < W) mk4[k] = (mk4[k] & ~(0xFFu << (8 * q))) | ((uint32_t)src[q] << (8 * q));
      }
    }
  }
  hy_stage<uint32_t>(s_d, dist, w, y0, x0, H_INF_DIST);
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int i = (int)threadIdx.x + 256 * k;
    uint8_t *d = s_m + (i >> 4) * HMS + (i & 15) * 4;
    d[0] = (uint8_t)mk4[k];
    d[1] = (uint8_t)(mk4[k] >> 8);
    d[2] = (uint8_t)(mk4[k] >> 16);
    d[3] = (uint8_t)(mk4[k] >> 24);
  }
  __syncthreads();
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  int any = 0, open = 0, ring = 0;
  for (int it = 0; it < sweeps; it++) {
    int ch;
    if (wave == 0) ch = hy_flat_sweep_m<true, 1>(s_m, s_d, lane);
    else if (wave == 1) ch = hy_flat_sweep_m<true, -1>(s_m, s_d, lane);
    else if (wave == 2) ch = hy_flat_sweep_m<false, 1>(s_m, s_d, lane);
    else ch = hy_flat_sweep_m<false, -1>(s_m, s_d, lane);
    open = __syncthreads_or(ch);
    if (!open) break;
    any = 1;
    ring |= __syncthreads_or(ch & 2);
  }
  if (act_cur && threadIdx.x == 0) act_cur[tile] = (uint8_t)((ring ? HY_CHANGED : 0) | (open ? HY_OPEN : 0));
  if (!any) return;
  if (whole) {  // 16 bytes at a time
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int i = (int)threadIdx.x + 256 * k;
      const int r = i >> 4, c4 = (i & 15) * 4;
      const uint32_t *sd = s_d + (r + 1) * HLS + 1 + c4;
      hy_v4u_a4 v = {sd[0], sd[1], sd[2], sd[3]};
      *reinterpret_cast<hy_v4u_a4 *>(dist + (long long)(y0 + r) * w.ld + x0 + c4) = v;
    }
  } else {
#pragma unroll
    for (int j = 0; j < H_CPT; j++) {
      int c = threadIdx.x + 256 * j;
      int y = y0 + c / HT, x = x0 + c % HT;
      if (y < H && x < W) dist[(long long)y * w.ld + x] = s_d[(c / HT + 1) * HLS + (c % HT) + 1];
    }
  }
  if (threadIdx.x == 0) atomicOr(changed, 1);
}

// flat cells point at a neighbour of the same filled height that is one hop closer: the first of N, W, E, S, else the
// first of NW, NE, SW, SE.  One workgroup per tile that has flat cells, surface and distances staged in LDS (round 4;
// one thread per cell with up to 16 scattered loads before: 5.3 ms at 16384^2).
__global__ __launch_bounds__(256) void k_flat_assign(const float *__restrict__ wsurf, const uint32_t *__restrict__ dist,
                                                    DtWin w, uint8_t *__restrict__ fdr,
                                                    int *__restrict__ unresolved, int tiles_x,
                                                    const uint8_t *__restrict__ has_flat) {
  __shared__ float s_w[HLD * HLS];
  __shared__ uint32_t s_d[HLD * HLS];
  if (has_flat && has_flat[blockIdx.x] == 0) return;
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
  const int y0 = ty * HT, x0 = tx * HT;
  // (beyond the raster: equal to no valid height)
  hy_stage2<float, uint32_t>(s_w, wsurf, DT_NODATA, s_d, dist, H_INF_DIST, w, y0, x0);
  __syncthreads();
  int bad = 0;
#pragma unroll
  for (int j = 0; j < H_CPT; j++) {
    const int c = threadIdx.x + 256 * j;
    const int ly = c / HT, lx = c % HT;
    const int y = y0 + ly, x = x0 + lx;
    if (y >= w.H || x >= w.W) continue;
    const int p = (ly + 1) * HLS + lx + 1;
    const uint32_t d = s_d[p];
    if (d == 0u) continue;
    uint32_t code = 0u;
    if (d != H_INF_DIST) {
      const float wc = s_w[p];
      const uint32_t want = d - 1u;
#define HY_A(off, c_)                                                  \
  if (!code && s_w[p + (off)] == wc && s_d[p + (off)] == want) code = (c_);
      HY_A(-HLS, 64u) HY_A(-1, 16u) HY_A(1, 1u) HY_A(HLS, 4u)
      HY_A(-HLS - 1, 32u) HY_A(-HLS + 1, 128u) HY_A(HLS - 1, 8u) HY_A(HLS + 1, 2u)
#undef HY_A
    }
    if (!code) bad++;
    fdr[(long long)y * w.ld + x] = (uint8_t)code;
  }
  for (int o = 32; o; o >>= 1) bad += __shfl_xor(bad, o);
  if ((threadIdx.x & 63) == 0 && bad) atomicAdd(unresolved, bad);
}

// ... of a single raster: from the distances and k_flat_init's byte per cell (no surface: see k_flat_relax_m)
__global__ __launch_bounds__(256) void k_flat_assign_m(const uint8_t *__restrict__ nsame, const uint32_t *__restrict__ dist,
                                                      DtWin w, uint8_t *__restrict__ fdr, int *__restrict__ unresolved,
                                                      int tiles_x, const uint8_t *__restrict__ has_flat) {
  __shared__ uint32_t s_d[HLD * HLS];
  if (has_flat && has_flat[blockIdx.x] == 0) return;
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
  const int y0 = ty * HT, x0 = tx * HT;
  // four cells in a row per thread and group (16 groups per tile row): their bytes as one 32-bit load where the rows
  // allow it, in flight with the staging's loads
  const bool whole = y0 + HT <= w.H && x0 + HT <= w.W;
  const bool word_ok = whole && (w.ld & 3) == 0 && ((uintptr_t)nsame & 3) == 0;
  uint32_t mk4[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int i = (int)threadIdx.x + 256 * k;
    const int r = i >> 4, c4 = (i & 15) * 4;
    const uint8_t *src = nsame + (long long)(y0 + r) * w.ld + x0 + c4;
    if (word_ok) {
      mk4[k] = *reinterpret_cast<const uint32_t *>(src);
    } else {
      mk4[k] = 0xFFFFFFFFu;
      if (y0 + r < w.H) {
#pragma unroll
        for (int q = 0; q < 4; q++)
          if (x0 + c4 + q < w.W) mk4[k] = (mk4[k] & ~(0xFFu << (8 * q))) | ((uint32_t)src[q] << (8 * q));
      }
    }
  }
  hy_stage<uint32_t>(s_d, dist, w, y0, x0, H_INF_DIST);
  __syncthreads();
  int bad = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int i = (int)threadIdx.x + 256 * k;
    const int r = i >> 4, c4 = (i & 15) * 4;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int y = y0 + r, x = x0 + c4 + q;
      if (y >= w.H || x >= w.W) continue;
      const int p = (r + 1) * HLS + c4 + q + 1;
      const uint32_t d = s_d[p];
      if (d == 0u) continue;
      uint32_t code = 0u;
      if (d != H_INF_DIST) {
        const uint32_t want = d - 1u, m = (mk4[k] >> (8 * q)) & 0xFFu;
#define HY_A(bit, off, c_) \
  if (!code && !((m >> (bit)) & 1u) && s_d[p + (off)] == want) code = (c_);
        HY_A(1, -HLS, 64u) HY_A(3, -1, 16u) HY_A(4, 1, 1u) HY_A(6, HLS, 4u)
        HY_A(0, -HLS - 1, 32u) HY_A(2, -HLS + 1, 128u) HY_A(5, HLS - 1, 8u) HY_A(7, HLS + 1, 2u)
#undef HY_A
      }
      if (!code) bad++;
      fdr[(long long)y * w.ld + x] = (uint8_t)code;
    }
  }
  for (int o = 32; o; o >>= 1) bad += __shfl_xor(bad, o);
  if ((threadIdx.x & 63) == 0 && bad) atomicAdd(unresolved, bad);
}

// scratch: flag words + the distance raster
// asynchronous form: the round flags (2 x up to DT_HYDRO_MAX_ASYNC_ROUNDS + the unresolved count) live in the
// first 4 KiB
#define DT_HYDRO_FLAG_BYTES 4096
#define DT_HYDRO_MAX_ASYNC_ROUNDS 500
static size_t hy_tiles(int64_t H, int64_t W) { return (size_t)((W + HT - 1) / HT) * (size_t)((H + HT - 1) / HT); }
// flags | distance raster | two per-tile activity arrays (the rounds alternate between them) | has_flat per tile
// | one byte per cell: which neighbours have another height (k_flat_init -> k_flat_relax_m)
size_t dt_hydro_scratch(int64_t H, int64_t W) {
  return DT_HYDRO_FLAG_BYTES + dt_align256((size_t)H * W * 4) + 3 * dt_align256(hy_tiles(H, W)) +
         dt_align256((size_t)H * W);
}

// raise the context's status when the budget of rounds did not reach the fixed point, or a flat cell got no code
__global__ void k_hydro_verdict(const int *__restrict__ last_fill, const int *__restrict__ last_flat,
                                const int *__restrict__ unresolved, int *__restrict__ status) {
  if ((*last_fill != 0 || *last_flat != 0 || *unresolved != 0) && status) atomicOr(status, DT_STATUS_NOT_CONVERGED);
}

// iterate `round` (a launch of one relaxation round over all tiles) until a whole batch changes nothing
// Termination: both relaxations are monotone (values only decrease) and every round contains at least one sweep over
// ALL cells against the previous round's values, so after k rounds every cell whose controlling path (spill path /
// shortest same-height path to a coded cell) has <= k cells is final -- the Bellman-Ford argument.  A controlling path
// has at most H * W cells: `max_rounds` = H * W + 8 is a true bound, never a tuning knob.  (A tile relaxes to its
// LOCAL fixed point each round, so in practice information moves a tile per round; but a path may cross tile borders
// far more often than there are tiles -- a serpentine channel with 1-cell walls crosses a 64-cell border 32 times --
// so the number of tiles bounds nothing.)
template <typename F>
static int hy_iterate(hipStream_t s, int *flags, int64_t max_rounds, F round, int *rounds_out) {
  // `round(flag, prev, r)` launches relaxation round r of the phase; it raises *flag when it changed something and
  // returns at once when *prev (the round before it, NULL for the first of a batch) was quiet.  Batches of 4, 8, ... 64 rounds
  // with one flag read per batch; the rounds of a batch after its first quiet one cost a few microseconds each.
  int64_t rounds = 0, launched = 0;  // launched: index of the round over the whole phase (0: visit every tile)
  int batch = 4;
  for (;;) {
    DT_HIP(hipMemsetAsync(flags, 0, sizeof(int) * 64, s));
    for (int b = 0; b < batch; b++)
      round(flags + b, b ? (const int *)(flags + b - 1) : (const int *)nullptr, launched++);
    int h[64];
    DT_HIP(hipMemcpyAsync(h, flags, sizeof(int) * (size_t)batch, hipMemcpyDeviceToHost, s));
    DT_HIP(hipStreamSynchronize(s));
    int used = 0;
    while (used < batch && h[used]) used++;
    rounds += used < batch ? used + 1 : batch;  // the rounds that did something, and the quiet one that proved it
    if (used < batch) break;
    DT_REQUIRE(rounds < max_rounds, "conditioning exceeded its proven bound of H * W rounds (a defect, not a property of the DEM)");
    if (batch < 64) batch *= 2;  // long-winded rasters: fewer host round trips per round
  }
  if (rounds_out) *rounds_out = (int)(rounds > 0x7FFFFFFF ? 0x7FFFFFFF : rounds);
  return DT_OK;
}

static DtWin hy_full_window(int64_t H, int64_t W) {
  DtWin w;
  w.H = (int)H; w.W = (int)W; w.ld = W; w.gy0 = 0; w.gx0 = 0; w.Hg = (int)H; w.Wg = (int)W; w.halo = 0;
  return w;
}

// One relaxation round of a single raster.  Rasters of at least HY_COLOUR_MIN_TILES tiles run COLOURED rounds (four
// launches, activity flags in place in act0: hy_tile_of_block); smaller ones keep one launch per round over every tile
// with the flags alternating between act0 and act1 -- a colour of the Example's 840 tiles does not fill a quarter of
// the chip, and four launches in a row cost it 1.9 ms where one costs 1.07 (16 + 12 rounds instead of 28 + 18 or not).
#define HY_COLOUR_MIN_TILES 16384
static size_t hy_colour_min() {
  const int v = dt_debug_get(DT_DBG_HY_COLOUR_MIN);  // (tests run the coloured form on small rasters)
  return v > 0 ? (size_t)v : (size_t)HY_COLOUR_MIN_TILES;
}
static void hy_fill_round(hipStream_t s, bool coloured, int64_t r, const float *dem, float *filled, const DtWin &w,
                          int tiles_x, int tiles_y, int *f, const int *prev, uint8_t *act0, uint8_t *act1, int sweeps) {
  const dim3 b(256);
  if (r == 0) {  // the first round initialises the surface itself (no k_fill_init pass): every tile at once
    hipLaunchKernelGGL(k_fill_relax<true>, dim3((unsigned)(tiles_x * tiles_y)), b, 0, s, dem, filled, w, tiles_x, f, prev,
                       (const uint8_t *)nullptr, act0, tiles_y, sweeps, -1);
  } else if (coloured) {
    for (int c = 0; c < 4; c++)
      if (hy_colour_blocks(c, tiles_x, tiles_y))
        hipLaunchKernelGGL(k_fill_relax<false>, dim3(hy_colour_blocks(c, tiles_x, tiles_y)), b, 0, s, dem, filled, w,
                           tiles_x, f, prev, (const uint8_t *)act0, act0, tiles_y, sweeps, c);
  } else {
    hipLaunchKernelGGL(k_fill_relax<false>, dim3((unsigned)(tiles_x * tiles_y)), b, 0, s, dem, filled, w, tiles_x, f, prev,
                       (const uint8_t *)((r - 1) & 1 ? act1 : act0), (r & 1) ? act1 : act0, tiles_y, sweeps, -1);
  }
}
// ... of the flat distances; has_flat: k_flat_init's per-tile flags, the activity the first round starts from (the
// coloured form has them copied into act0 beforehand)
static void hy_flat_round(hipStream_t s, bool coloured, int64_t r, const uint8_t *nsame, uint32_t *dist, const DtWin &w,
                          int tiles_x, int tiles_y, int *f, const int *prev, uint8_t *act0, uint8_t *act1,
                          const uint8_t *has_flat, int sweeps) {
  const dim3 b(256);
  if (coloured) {
    for (int c = 0; c < 4; c++)
      if (hy_colour_blocks(c, tiles_x, tiles_y))
        hipLaunchKernelGGL(k_flat_relax_m, dim3(hy_colour_blocks(c, tiles_x, tiles_y)), b, 0, s, nsame, dist, w, tiles_x, f,
                           prev, (const uint8_t *)act0, act0, tiles_y, sweeps, c);
  } else {
    hipLaunchKernelGGL(k_flat_relax_m, dim3((unsigned)(tiles_x * tiles_y)), b, 0, s, nsame, dist, w, tiles_x, f, prev,
                       r ? (const uint8_t *)((r - 1) & 1 ? act1 : act0) : has_flat, (r & 1) ? act1 : act0, tiles_y,
                       sweeps, -1);
  }
}

// dem -> filled surface (may alias nothing), D8 codes with flats resolved.  *unresolved_host = flat cells left
// without a code (0 on any raster: every flat of a filled surface reaches a coded cell).  Synchronous.
int dt_launch_condition(hipStream_t s, const float *dem, int64_t H, int64_t W, double px, float *filled, uint8_t *fdr,
                        void *scratch, int *unresolved_host, int *rounds_host) {
  const int fill_sweeps = dt_debug_get(DT_DBG_HY_FILL_SWEEPS) > 0 ? dt_debug_get(DT_DBG_HY_FILL_SWEEPS) : HY_FILL_SWEEPS;
  const int flat_sweeps = dt_debug_get(DT_DBG_HY_FLAT_SWEEPS) > 0 ? dt_debug_get(DT_DBG_HY_FLAT_SWEEPS) : HY_FLAT_SWEEPS;
  (void)fill_sweeps;
  (void)flat_sweeps;
  const int64_t n = H * W;
  if (n == 0) return DT_OK;
  int *flag = (int *)scratch;
  uint32_t *dist = (uint32_t *)((char *)scratch + DT_HYDRO_FLAG_BYTES);
  const DtWin w = hy_full_window(H, W);
  const int tiles_x = (int)((W + HT - 1) / HT), tiles_y = (int)((H + HT - 1) / HT);
  dim3 gc((unsigned)((n + 255) / 256)), gt((unsigned)(tiles_x * tiles_y)), b(256);
  int r1 = 0, r2 = 0;
  const int64_t max_rounds = n + 8;
  uint8_t *act = (uint8_t *)scratch + DT_HYDRO_FLAG_BYTES + dt_align256((size_t)n * 4);
  uint8_t *act1 = act + dt_align256(hy_tiles(H, W));
  uint8_t *has_flat = act + 2 * dt_align256(hy_tiles(H, W));
  uint8_t *nsame = has_flat + dt_align256(hy_tiles(H, W));
  const bool coloured = hy_tiles(H, W) >= hy_colour_min();
  DT_TRY(hy_iterate(s, flag, max_rounds, [&](int *f, const int *prev, int64_t r) {
    hy_fill_round(s, coloured, r, dem, filled, w, tiles_x, tiles_y, f, prev, act, act1, fill_sweeps);
  }, &r1));
  if (fdr) {
    // (the distance raster is not in use yet: it lends the D8 kernel its mark / mask workspace -- the hot / cold pair
    // of the chain's first op, 0.40 ms at 16384^2, instead of the generic stencil, 0.55)
    DT_TRY(dt_launch_stencil(s, w, filled, px, nullptr, fdr, nullptr, nullptr, 0, 0.0, nullptr, nullptr,
                             dt_stencil_aux_bytes(H, W) <= dt_align256((size_t)n * 4) ? (void *)dist : nullptr));
    hipLaunchKernelGGL(k_flat_init, gt, b, 0, s, filled, fdr, w, dist, tiles_x, has_flat, nsame);
    // the rounds start from the tiles that have flat cells (and their neighbours), not from every tile
    if (coloured) DT_HIP(hipMemcpyAsync(act, has_flat, hy_tiles(H, W), hipMemcpyDeviceToDevice, s));
    DT_TRY(hy_iterate(s, flag, max_rounds, [&](int *f, const int *prev, int64_t r) {
      hy_flat_round(s, coloured, r, nsame, dist, w, tiles_x, tiles_y, f, prev, act, act1, has_flat, flat_sweeps);
    }, &r2));
    DT_HIP(hipMemsetAsync(flag + 64, 0, sizeof(int), s));
    hipLaunchKernelGGL(k_flat_assign_m, gt, b, 0, s, (const uint8_t *)nsame, dist, w, fdr, flag + 64, tiles_x,
                       (const uint8_t *)has_flat);
    int u = 0;
    DT_HIP(hipMemcpyAsync(&u, flag + 64, sizeof(int), hipMemcpyDeviceToHost, s));
    DT_HIP(hipStreamSynchronize(s));
    if (unresolved_host) *unresolved_host = u;
  }
  if (rounds_host) {
    rounds_host[0] = r1;
    rounds_host[1] = r2;
  }
  return DT_OK;
}

// The same conditioning without a single host synchronisation (the resident chain's form): a fixed budget of `rounds`
// fill rounds and `rounds` flat rounds is enqueued; every round records whether it changed anything, a round that
// follows a quiet one returns at once (a few microseconds), and a last kernel raises DT_STATUS_NOT_CONVERGED on the
// context when the budget ran out before the fixed point (or a flat cell was left without a code): the rasters are
// then NOT the conditioned ones -- run again with a larger budget, or use the synchronous form, which iterates to
// the fixed point whatever it takes.  The bundled Example raster needs 12 rounds, rough 4096^2 terrain a few dozen.
int dt_launch_condition_async(hipStream_t s, const float *dem, int64_t H, int64_t W, double px, float *filled,
                              uint8_t *fdr, void *scratch, int rounds, int *status) {
  const int fill_sweeps = dt_debug_get(DT_DBG_HY_FILL_SWEEPS) > 0 ? dt_debug_get(DT_DBG_HY_FILL_SWEEPS) : HY_FILL_SWEEPS;
  const int flat_sweeps = dt_debug_get(DT_DBG_HY_FLAT_SWEEPS) > 0 ? dt_debug_get(DT_DBG_HY_FLAT_SWEEPS) : HY_FLAT_SWEEPS;
  (void)fill_sweeps;
  (void)flat_sweeps;
  const int64_t n = H * W;
  if (n == 0) return DT_OK;
  DT_REQUIRE(fdr != nullptr, "the asynchronous conditioning writes the D8 codes");
  DT_REQUIRE(rounds >= 1 && rounds <= DT_HYDRO_MAX_ASYNC_ROUNDS, "1..500 rounds");
  int *flags = (int *)scratch;  // [0, rounds): fill rounds; [rounds, 2 rounds): flat rounds; [2 rounds]: unresolved
  uint32_t *dist = (uint32_t *)((char *)scratch + DT_HYDRO_FLAG_BYTES);
  const DtWin w = hy_full_window(H, W);
  const int tiles_x = (int)((W + HT - 1) / HT), tiles_y = (int)((H + HT - 1) / HT);
  dim3 gc((unsigned)((n + 255) / 256)), gt((unsigned)(tiles_x * tiles_y)), b(256);
  DT_HIP(hipMemsetAsync(flags, 0, DT_HYDRO_FLAG_BYTES, s));
  uint8_t *act = (uint8_t *)scratch + DT_HYDRO_FLAG_BYTES + dt_align256((size_t)n * 4);
  uint8_t *act1 = act + dt_align256(hy_tiles(H, W));
  uint8_t *has_flat = act + 2 * dt_align256(hy_tiles(H, W));
  uint8_t *nsame = has_flat + dt_align256(hy_tiles(H, W));
  const bool coloured = hy_tiles(H, W) >= hy_colour_min();
  for (int r = 0; r < rounds; r++)
    hy_fill_round(s, coloured, r, dem, filled, w, tiles_x, tiles_y, flags + r, r ? (const int *)(flags + r - 1) : nullptr,
                  act, act1, fill_sweeps);
  // (the distance raster is not in use yet: it lends the D8 kernel its mark / mask workspace -- the hot / cold pair
    // of the chain's first op, 0.40 ms at 16384^2, instead of the generic stencil, 0.55)
    DT_TRY(dt_launch_stencil(s, w, filled, px, nullptr, fdr, nullptr, nullptr, 0, 0.0, nullptr, nullptr,
                             dt_stencil_aux_bytes(H, W) <= dt_align256((size_t)n * 4) ? (void *)dist : nullptr));
  hipLaunchKernelGGL(k_flat_init, gt, b, 0, s, filled, fdr, w, dist, tiles_x, has_flat, nsame);
  if (coloured) DT_HIP(hipMemcpyAsync(act, has_flat, hy_tiles(H, W), hipMemcpyDeviceToDevice, s));
  int *fl2 = flags + rounds;
  for (int r = 0; r < rounds; r++)
    hy_flat_round(s, coloured, r, nsame, dist, w, tiles_x, tiles_y, fl2 + r, r ? (const int *)(fl2 + r - 1) : nullptr, act,
                  act1, has_flat, flat_sweeps);
  hipLaunchKernelGGL(k_flat_assign_m, gt, b, 0, s, (const uint8_t *)nsame, dist, w, fdr, flags + 2 * rounds, tiles_x,
                     (const uint8_t *)has_flat);
  hipLaunchKernelGGL(k_hydro_verdict, dim3(1), dim3(1), 0, s, (const int *)(flags + rounds - 1),
                     (const int *)(fl2 + rounds - 1), (const int *)(flags + 2 * rounds), status);
  return DT_OK;
}

// ---- the steps on one rank's window of a larger raster (multi-GPU: descriptools_amd/tiling.py iterates them with a
// halo exchange of the surface / the distances in between until no rank changes anything) ---------------------
// stage 0 init of the surface, 1 `rounds` fill rounds, 2 init of the flat distances (after D8 on the surface),
// 3 `rounds` flat rounds, 4 assignment of the flat cells' codes.  *flag_dev (device int, zeroed by the caller per
// iteration) is raised by stages 1 / 3 when something changed, and counts the unresolved cells in stage 4.
// nsame (optional; a uint8 raster laid out like the others): the flat stages run from the byte per cell that stage 2
// leaves there instead of from the surface (k_flat_relax_m / k_flat_assign_m)
int dt_launch_condition_stage(hipStream_t s, const DtWin &w, int stage, int rounds, const float *dem, float *filled,
                              uint8_t *fdr, uint32_t *dist, int *flag_dev, uint8_t *nsame) {
  const int fill_sweeps = dt_debug_get(DT_DBG_HY_FILL_SWEEPS) > 0 ? dt_debug_get(DT_DBG_HY_FILL_SWEEPS) : HY_FILL_SWEEPS;
  const int flat_sweeps = dt_debug_get(DT_DBG_HY_FLAT_SWEEPS) > 0 ? dt_debug_get(DT_DBG_HY_FLAT_SWEEPS) : HY_FLAT_SWEEPS;
  (void)fill_sweeps;
  (void)flat_sweeps;
  const int64_t n = (int64_t)w.H * w.W;
  if (n == 0) return DT_OK;
  const int tiles_x = (w.W + HT - 1) / HT, tiles_y = (w.H + HT - 1) / HT;
  dim3 gc((unsigned)((n + 255) / 256)), gt((unsigned)(tiles_x * tiles_y)), b(256);
  DT_REQUIRE(rounds >= 1 || (stage != 1 && stage != 3), "rounds < 1");
  const bool coloured = (size_t)tiles_x * tiles_y >= hy_colour_min();  // (hy_fill_round)
  switch (stage) {
    case 0:
      DT_REQUIRE(dem && filled, "NULL raster");
      hipLaunchKernelGGL(k_fill_init, gc, b, 0, s, dem, w, filled);
      break;
    case 1:
      DT_REQUIRE(dem && filled && flag_dev, "NULL pointer");
      for (int r = 0; r < rounds; r++)
        for (int c = coloured ? 0 : -1; c < (coloured ? 4 : 0); c++)
          if (hy_colour_blocks(c, tiles_x, tiles_y))
            hipLaunchKernelGGL(k_fill_relax<false>, dim3(hy_colour_blocks(c, tiles_x, tiles_y)), b, 0, s, dem, filled, w,
                               tiles_x, flag_dev, (const int *)nullptr, (const uint8_t *)nullptr, (uint8_t *)nullptr,
                               tiles_y, fill_sweeps, c);
      break;
    case 2:
      DT_REQUIRE(filled && fdr && dist, "NULL pointer");
      hipLaunchKernelGGL(k_flat_init, gt, b, 0, s, filled, fdr, w, dist, tiles_x, (uint8_t *)nullptr, nsame);
      break;
    case 3:
      DT_REQUIRE(filled && dist && flag_dev, "NULL pointer");
      for (int r = 0; r < rounds; r++)
        for (int c = coloured ? 0 : -1; c < (coloured ? 4 : 0); c++)
          if (hy_colour_blocks(c, tiles_x, tiles_y)) {
            if (nsame)
              hipLaunchKernelGGL(k_flat_relax_m, dim3(hy_colour_blocks(c, tiles_x, tiles_y)), b, 0, s,
                                 (const uint8_t *)nsame, dist, w, tiles_x, flag_dev, (const int *)nullptr,
                                 (const uint8_t *)nullptr, (uint8_t *)nullptr, tiles_y, flat_sweeps, c);
            else
              hipLaunchKernelGGL(k_flat_relax, dim3(hy_colour_blocks(c, tiles_x, tiles_y)), b, 0, s, filled, dist, w,
                                 tiles_x, flag_dev, (const int *)nullptr, (const uint8_t *)nullptr, (uint8_t *)nullptr,
                                 tiles_y, flat_sweeps, c);
          }
      break;
    case 4:
      DT_REQUIRE(filled && dist && fdr && flag_dev, "NULL pointer");
      if (nsame)
        hipLaunchKernelGGL(k_flat_assign_m, gt, b, 0, s, (const uint8_t *)nsame, dist, w, fdr, flag_dev, tiles_x,
                           (const uint8_t *)nullptr);
      else
        hipLaunchKernelGGL(k_flat_assign, gt, b, 0, s, filled, dist, w, fdr, flag_dev, tiles_x, (const uint8_t *)nullptr);
      break;
    default:
      DT_REQUIRE(false, "stage must be 0..4");
  }
  return DT_OK;
}
