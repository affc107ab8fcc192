// dt_tiles.hip -- tile-hierarchical D8 graph kernels (flow accumulation, flow distance / HAND).
//
// The D8 graph is a forest embedded in the raster, so both descriptors are tree computations whose
// naive GPU forms are latency chains through HBM (one dependent access per cell of the longest
// flow path).  Here the raster is cut into 64 x 64 tiles that are solved entirely inside LDS, and
// only the tile PERIMETER (252 of 4096 cells) takes part in a small global graph:
//
//   pass 1 (per tile, LDS)   solve the tile in isolation; emit one 8-byte record per perimeter cell
//                            describing where a path entering there leaves the tile (or ends)
//   pass 2 (perimeter graph) resolve the records globally (countdown / pointer doubling over
//                            ~6 % of the cells)
//   pass 3 (per tile, LDS)   re-solve the tile with the resolved perimeter values and write the
//                            rasters, coalesced
//
// HBM traffic is ~2 reads of the 1-byte direction raster plus the outputs; everything else is LDS.
// The same perimeter records are what a multi-GPU run exchanges.
//
// Integer accumulation is order-independent, and path counts are integers, so results are
// bit-identical to the v1 kernels and to the oracle.
#include "dt_common.h"
#include "dt_kernels.h"

#define TW 64
#define TH 64
#define NT (TW * TH)
#define PS (2 * TW + 2 * (TH - 2)) /* perimeter slots per tile: 252 */
#define CPT (NT / 256)             /* cells per thread: 16 */

#define NX_SINK 0xFFFFu /* no in-raster / valid D8 target */
#define NX_EXIT 0xFFFEu /* target is in the raster but in another tile */
#define X_NONE 0xFFFFu

// perimeter slot <-> local cell
__device__ __forceinline__ int dt_slot_of(int ly, int lx) {
  if (ly == 0) return lx;
  if (ly == TH - 1) return TW + lx;
  if (lx == 0) return 2 * TW + (ly - 1);
  if (lx == TW - 1) return 2 * TW + (TH - 2) + (ly - 1);
  return -1;
}
__device__ __forceinline__ void dt_cell_of_slot(int s, int &ly, int &lx) {
  if (s < TW) { ly = 0; lx = s; }
  else if (s < 2 * TW) { ly = TH - 1; lx = s - TW; }
  else if (s < 2 * TW + (TH - 2)) { ly = s - 2 * TW + 1; lx = 0; }
  else { ly = s - 2 * TW - (TH - 2) + 1; lx = TW - 1; }
}

// ---- tile staging -----------------------------------------------------------------------------
// loads the tile's direction codes into LDS (0 outside the raster) and derives the local successor
// of every cell: local index, NX_EXIT or NX_SINK.
__device__ __forceinline__ void dt_tile_load_fdr(const uint8_t *__restrict__ fdr, int H, int W, int y0,
                                                 int x0, uint8_t *s_fdr) {
  // 256 threads x 16 bytes = one 64-byte row per 4 threads
  int t = threadIdx.x;
  int r = t >> 2, c = (t & 3) * 16;
  int gy = y0 + r, gx = x0 + c;
  uint4 v = make_uint4(0, 0, 0, 0);
  if (gy < H) {
    const uint8_t *p = fdr + (size_t)gy * W + gx;
    if (gx + 15 < W && (((uintptr_t)p) & 15) == 0) {
      v = *reinterpret_cast<const uint4 *>(p);
    } else {
      uint8_t b[16];
#pragma unroll
      for (int k = 0; k < 16; k++) b[k] = (gx + k < W) ? p[k] : 0;
      v.x = b[0] | (b[1] << 8) | (b[2] << 16) | ((uint32_t)b[3] << 24);
      v.y = b[4] | (b[5] << 8) | (b[6] << 16) | ((uint32_t)b[7] << 24);
      v.z = b[8] | (b[9] << 8) | (b[10] << 16) | ((uint32_t)b[11] << 24);
      v.w = b[12] | (b[13] << 8) | (b[14] << 16) | ((uint32_t)b[15] << 24);
    }
  }
  *reinterpret_cast<uint4 *>(&s_fdr[r * TW + c]) = v;
}

__device__ __forceinline__ uint32_t dt_tile_next(uint32_t code, int ly, int lx, int y0, int x0, int H,
                                                int W) {
  int gy = y0 + ly, gx = x0 + lx;
  if (gy >= H || gx >= W || !dt_d8_valid(code)) return NX_SINK;
  int dy, dx;
  dt_d8_delta(code, dy, dx);
  int ny = ly + dy, nx = lx + dx;
  int ty = gy + dy, tx = gx + dx;
  if (ty < 0 || ty >= H || tx < 0 || tx >= W) return NX_SINK;
  if (ny < 0 || ny >= TH || nx < 0 || nx >= TW) return NX_EXIT;
  return (uint32_t)(ny * TW + nx);
}

// ===========================================================================================
// Flow accumulation
// ===========================================================================================
// In-tile subtree sums by pointer doubling with scatter, all in LDS, no serial chains:
//   val_k(c)  = sum of weight(u) over the cells u whose in-tile path reaches c in < 2^k moves
//   ptr_k(c)  = the cell 2^k moves downstream of c (bit 15 "alive": such a cell exists), or the
//               last in-tile cell of c's path (alive clear)
//   round k:  every alive c adds val_k(c) to recv(ptr_k(c));  val_{k+1} = val_k + recv;
//             ptr_{k+1}(c) = ptr_k(ptr_k(c))
// Exact for integers in any order.  12 rounds cover every acyclic in-tile path (< 4096 moves);
// cells still alive afterwards run into an in-tile D8 cycle, and ptr_12 of those cells enumerates
// exactly the cells ON the cycles (marked in s_cyc).
#define PT_ALIVE 0x8000u
#define PT_EXIT 0x4000u
#define PT_IDX 0x0FFFu

__device__ __forceinline__ void dt_tile_sums(uint16_t *s_ptr, uint32_t *s_val, uint32_t *s_recv,
                                             uint8_t *s_cyc) {
  uint16_t np[CPT];
  for (int round = 0; round < 12; round++) {
    int any = 0;
#pragma unroll
    for (int j = 0; j < CPT; j++) {
      int c = threadIdx.x + 256 * j;
      uint32_t p = s_ptr[c];
      np[j] = (uint16_t)p;
      if (p & PT_ALIVE) {
        uint32_t t = p & PT_IDX;
        atomicAdd(&s_recv[t], s_val[c]);
        np[j] = s_ptr[t];
        any = 1;
      }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < CPT; j++) {
      int c = threadIdx.x + 256 * j;
      uint32_t r = s_recv[c];
      if (r) {
        s_val[c] += r;
        s_recv[c] = 0;
      }
      s_ptr[c] = np[j];
    }
    if (!__syncthreads_or(any)) return;
  }
  // still alive after 2^12 moves: the path never ends inside the tile -> in-tile cycle
#pragma unroll
  for (int j = 0; j < CPT; j++) {
    int c = threadIdx.x + 256 * j;
    uint32_t p = s_ptr[c];
    if (p & PT_ALIVE) s_cyc[p & PT_IDX] = 1;
  }
  __syncthreads();
}

// perimeter record (8 bytes):  W:32 | xslot:16 | code:8 | flags:8
//   code  = the cell's D8 code when its successor is in ANOTHER tile (an "exit" cell), else 0
//   W     = cells draining through the exit cell inside the tile, itself included
//   xslot = perimeter slot of the exit cell reached by a path entering the tile at this cell
//           (X_NONE when that path ends inside the tile)
__device__ __forceinline__ unsigned long long fa_rec(uint32_t W_, uint32_t xslot, uint32_t code) {
  return ((unsigned long long)W_ << 32) | ((unsigned long long)(xslot & 0xFFFFu) << 16) |
         ((unsigned long long)(code & 0xFFu) << 8);
}

__global__ __launch_bounds__(256) void k_fa_tile1(const uint8_t *__restrict__ fdr, int H, int W,
                                                 int tiles_x, unsigned long long *__restrict__ rec,
                                                 int32_t *__restrict__ acc32) {
  __shared__ __attribute__((aligned(16))) uint8_t s_fdr[NT];  // reused as the in-tile cycle mask
  __shared__ uint16_t s_ptr[NT];                               // idx:12 | PT_EXIT | PT_ALIVE
  __shared__ uint32_t s_val[NT];
  __shared__ uint32_t s_recv[NT];
  const int tile = blockIdx.x;
  const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
  const int y0 = ty * TH, x0 = tx * TW;
  dt_tile_load_fdr(fdr, H, W, y0, x0, s_fdr);
  __syncthreads();
  uint32_t nx[CPT];
  for (int j = 0; j < CPT; j++) {
    int c = threadIdx.x + 256 * j;
    nx[j] = dt_tile_next(s_fdr[c], c / TW, c % TW, y0, x0, H, W);
  }
  uint32_t my_code = 0;  // D8 code of my perimeter cell when it is an exit cell
  if (threadIdx.x < PS) {
    int ly, lx;
    dt_cell_of_slot(threadIdx.x, ly, lx);
    uint32_t code = s_fdr[ly * TW + lx];
    if (dt_tile_next(code, ly, lx, y0, x0, H, W) == NX_EXIT) my_code = code;
  }
  __syncthreads();
  uint8_t *s_cyc = s_fdr;
  for (int j = 0; j < CPT; j++) {
    int c = threadIdx.x + 256 * j;
    uint32_t n = nx[j];
    // terminals point at themselves; an exit terminal carries PT_EXIT, which every cell whose
    // in-tile path ends there inherits through the jumps
    s_ptr[c] = (uint16_t)(n < NT ? (n | PT_ALIVE) : ((uint32_t)c | (n == NX_EXIT ? PT_EXIT : 0u)));
    s_val[c] = 1u;
    s_recv[c] = 0u;
    s_cyc[c] = 0;
  }
  __syncthreads();
  dt_tile_sums(s_ptr, s_val, s_recv, s_cyc);
  if (threadIdx.x < PS) {
    int ly, lx;
    dt_cell_of_slot(threadIdx.x, ly, lx);
    int c = ly * TW + lx;
    uint32_t p = s_ptr[c];
    uint32_t xs = X_NONE;
    if (!(p & PT_ALIVE) && (p & PT_EXIT)) {
      uint32_t f = p & PT_IDX;
      xs = (uint32_t)dt_slot_of((int)f / TW, (int)f % TW);
    }
    rec[(size_t)tile * PS + threadIdx.x] = fa_rec(my_code ? s_val[c] : 0u, xs, my_code);
  }
  // in-tile accumulation (upstream cells of this tile only); pass 3 adds what enters from outside
  for (int j = 0; j < CPT; j++) {
    int c = threadIdx.x + 256 * j;
    int gy = y0 + c / TW, gx = x0 + c % TW;
    if (gy < H && gx < W) acc32[(size_t)gy * W + gx] = s_cyc[c] ? -100 : (int32_t)(s_val[c] - 1u);
  }
}

// perimeter graph: node id = tile * PS + slot.  For every exit node find the entry node it feeds
// (the neighbouring tile's perimeter cell its D8 step lands on) and the exit node that entry's
// in-tile path leads to (its parent in the reduced forest).
#define FA_NONE 0xFFFFFFFFu
#define FA2_SH 56
#define FA2_MASK ((1ull << FA2_SH) - 1ull)
#define FA_CYCLE (1ull << 63) /* ext flag: this entry cell is fed by a cross-tile D8 cycle */

__global__ __launch_bounds__(256) void k_fa_link(const unsigned long long *__restrict__ rec, int64_t nnodes,
                                                int tiles_x, int H, int W, uint32_t *__restrict__ entry_of,
                                                uint32_t *__restrict__ parent,
                                                unsigned long long *__restrict__ state) {
  int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (n >= nnodes) return;
  unsigned long long r = rec[n];
  uint32_t code = (uint32_t)((r >> 8) & 0xFFu);
  uint32_t ent = FA_NONE, par = FA_NONE;
  if (code) {
    int tile = (int)(n / PS), slot = (int)(n - (int64_t)tile * PS);
    int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    int ly, lx, dy, dx;
    dt_cell_of_slot(slot, ly, lx);
    dt_d8_delta(code, dy, dx);
    int gy = ty * TH + ly + dy, gx = tx * TW + lx + dx;  // inside the raster by construction
    int t2y = gy / TH, t2x = gx / TW;
    int s2 = dt_slot_of(gy - t2y * TH, gx - t2x * TW);
    ent = (uint32_t)((t2y * tiles_x + t2x) * PS + s2);
    uint32_t xs = (uint32_t)((rec[ent] >> 16) & 0xFFFFu);
    if (xs != X_NONE) {
      par = (uint32_t)((t2y * tiles_x + t2x) * PS) + xs;
      atomicAdd(&state[par], 1ull << FA2_SH);
    }
  }
  entry_of[n] = ent;
  parent[n] = par;
}

// countdown over the reduced forest; A(q) = W(q) + sum of A over the exit nodes feeding q's tile
// entry cells whose in-tile path leads to q.  ext[entry] accumulates the inflow arriving at an
// entry cell from other tiles.
__global__ __launch_bounds__(256) void k_fa_reduce(const unsigned long long *__restrict__ rec, int64_t nnodes,
                                                  const uint32_t *__restrict__ entry_of,
                                                  const uint32_t *__restrict__ parent,
                                                  unsigned long long *__restrict__ state,
                                                  unsigned long long *__restrict__ ext) {
  int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (n >= nnodes) return;
  unsigned long long r = rec[n];
  if (((r >> 8) & 0xFFu) == 0) return;  // not an exit node
  if (state[n] != 0ull) return;         // not a source of the reduced forest
  uint32_t q = (uint32_t)n;
  unsigned long long A = r >> 32;
  for (int64_t it = 0; it < nnodes; it++) {
    atomicAdd(&ext[entry_of[q]], A);
    uint32_t p = parent[q];
    if (p == FA_NONE) break;
    unsigned long long old = atomicAdd(&state[p], A - (1ull << FA2_SH));
    if ((old >> FA2_SH) != 1ull) break;
    A = (rec[p] >> 32) + (old & FA2_MASK) + A;
    q = p;
  }
}

// exit nodes that never resolved sit on a D8 cycle spanning tiles: flag the entry cells they feed so
// that pass 3 marks the in-tile stretch of the cycle -100 (the oracle's "in-degree never reaches 0").
__global__ __launch_bounds__(256) void k_fa_poison(const unsigned long long *__restrict__ rec, int64_t nnodes,
                                                  const uint32_t *__restrict__ entry_of,
                                                  const unsigned long long *__restrict__ state,
                                                  unsigned long long *__restrict__ ext) {
  int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (n >= nnodes) return;
  if (((rec[n] >> 8) & 0xFFu) == 0) return;
  if ((state[n] >> FA2_SH) != 0ull) atomicOr(&ext[entry_of[n]], FA_CYCLE);
}

// pass 3: the inflow that enters the tile at a perimeter cell p (ext[p], resolved by pass 2) drains
// through every cell of p's in-tile path: one lane per entry cell walks that path adding ext[p] to
// an LDS delta raster (integer adds: order-free), then delta is added to pass 1's in-tile counts.
template <bool HAS_DEM, bool W_RIVER>
__global__ __launch_bounds__(256) void k_fa_tile3(const uint8_t *__restrict__ fdr,
                                                 const float *__restrict__ dem, int H, int W, int tiles_x,
                                                 const unsigned long long *__restrict__ ext,
                                                 int32_t *__restrict__ acc32, int32_t river_thr,
                                                 int8_t *__restrict__ river) {
  __shared__ __attribute__((aligned(16))) uint8_t s_fdr[NT];  // reused as the cross-tile cycle mask
  __shared__ uint16_t s_nxt[NT];
  __shared__ uint32_t s_delta[NT];
  const int tile = blockIdx.x;
  const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
  const int y0 = ty * TH, x0 = tx * TW;
  dt_tile_load_fdr(fdr, H, W, y0, x0, s_fdr);
  unsigned long long e = 0ull;
  if (threadIdx.x < PS) e = ext[(size_t)tile * PS + threadIdx.x];
  __syncthreads();
  uint32_t nx[CPT];
  for (int j = 0; j < CPT; j++) {
    int c = threadIdx.x + 256 * j;
    nx[j] = dt_tile_next(s_fdr[c], c / TW, c % TW, y0, x0, H, W);
  }
  __syncthreads();
  uint8_t *s_cyc = s_fdr;
  for (int j = 0; j < CPT; j++) {
    int c = threadIdx.x + 256 * j;
    s_nxt[c] = (uint16_t)nx[j];
    s_delta[c] = 0u;
    s_cyc[c] = 0;
  }
  __syncthreads();
  if (e != 0ull) {
    int ly, lx;
    dt_cell_of_slot(threadIdx.x, ly, lx);
    uint32_t c = (uint32_t)(ly * TW + lx);
    const bool cyc = (e & FA_CYCLE) != 0ull;  // fed by a D8 cycle spanning tiles: the path IS the cycle
    const uint32_t add = (uint32_t)e;         // real inflow < 2^31
    for (int it = 0; it < NT && c < NT; it++) {
      if (cyc) s_cyc[c] = 1;
      else atomicAdd(&s_delta[c], add);
      c = s_nxt[c];
    }
  }
  __syncthreads();
  for (int j = 0; j < CPT; j++) {
    int c = threadIdx.x + 256 * j;
    int gy = y0 + c / TW, gx = x0 + c % TW;
    if (gy >= H || gx >= W) continue;
    size_t o = (size_t)gy * W + gx;
    int32_t v = acc32[o];
    if (v != -100) v += (int32_t)s_delta[c];
    if (s_cyc[c]) v = -100;
    if (HAS_DEM && dem[o] <= DT_NODATA) v = -100;
    acc32[o] = v;
    if (W_RIVER) river[o] = v > river_thr ? 1 : 0;
  }
}

int dt_launch_flowacc_tiled(hipStream_t s, const uint8_t *fdr, const float *dem, int64_t H, int64_t W,
                            void *scratch, size_t scratch_bytes, int32_t *acc32, int64_t river_thr,
                            int8_t *river) {
  if (H == 0 || W == 0) return DT_OK;
  int tiles_x = (int)((W + TW - 1) / TW), tiles_y = (int)((H + TH - 1) / TH);
  int64_t ntiles = (int64_t)tiles_x * tiles_y;
  int64_t nnodes = ntiles * PS;
  DT_REQUIRE(nnodes < 0xFFFFFFF0ll, "raster too large for one device tile");
  DT_REQUIRE(scratch_bytes >= dt_flowacc_tiled_scratch(H, W), "scratch too small");
  char *p = (char *)scratch;
  unsigned long long *rec = (unsigned long long *)p;  p += dt_align256((size_t)nnodes * 8);
  unsigned long long *state = (unsigned long long *)p;  p += dt_align256((size_t)nnodes * 8);
  unsigned long long *ext = (unsigned long long *)p;  p += dt_align256((size_t)nnodes * 8);
  uint32_t *entry_of = (uint32_t *)p;  p += dt_align256((size_t)nnodes * 4);
  uint32_t *parent = (uint32_t *)p;
  // state and ext are contiguous: one memset
  DT_HIP(hipMemsetAsync(state, 0, dt_align256((size_t)nnodes * 8) * 2, s));
  hipLaunchKernelGGL(k_fa_tile1, dim3((unsigned)ntiles), dim3(256), 0, s, fdr, (int)H, (int)W, tiles_x, rec, acc32);
  dim3 gn((unsigned)((nnodes + 255) / 256)), b(256);
  hipLaunchKernelGGL(k_fa_link, gn, b, 0, s, rec, nnodes, tiles_x, (int)H, (int)W, entry_of, parent, state);
  hipLaunchKernelGGL(k_fa_reduce, gn, b, 0, s, rec, nnodes, entry_of, parent, state, ext);
  hipLaunchKernelGGL(k_fa_poison, gn, b, 0, s, rec, nnodes, entry_of, state, ext);
  int32_t thr = river_thr > 2147483647ll ? 2147483647 : (river_thr < -2147483647ll ? -2147483647 : (int32_t)river_thr);
  dim3 gt((unsigned)ntiles);
  if (dem && river) hipLaunchKernelGGL((k_fa_tile3<true, true>), gt, b, 0, s, fdr, dem, (int)H, (int)W, tiles_x, ext, acc32, thr, river);
  else if (dem) hipLaunchKernelGGL((k_fa_tile3<true, false>), gt, b, 0, s, fdr, dem, (int)H, (int)W, tiles_x, ext, acc32, thr, river);
  else if (river) hipLaunchKernelGGL((k_fa_tile3<false, true>), gt, b, 0, s, fdr, dem, (int)H, (int)W, tiles_x, ext, acc32, thr, river);
  else hipLaunchKernelGGL((k_fa_tile3<false, false>), gt, b, 0, s, fdr, dem, (int)H, (int)W, tiles_x, ext, acc32, thr, river);
  return DT_OK;
}

size_t dt_flowacc_tiled_scratch(int64_t H, int64_t W) {
  int64_t ntiles = ((W + TW - 1) / TW) * ((H + TH - 1) / TH);
  size_t nn = (size_t)ntiles * PS;
  return dt_align256(nn * 8) * 3 + dt_align256(nn * 4) * 2 + 256;
}

// ===========================================================================================
// Flow distance / drained-to river index / HAND (F3, F4; flowhand.py:566-846, :414-442)
// ===========================================================================================
// Word format (LDS per cell, and global per perimeter node; identical to the v1 kernels'):
//   ptr:32 | n_diag:16 | done:1 n_card:15
// "the path from here to `ptr` takes n_card cardinal and n_diag diagonal moves".
//   in a tile : ptr = local cell (bits 0-11) | kind << 12, kind of a finished path's end:
//               1 river cell, 2 dead (-100), 3 exit (the end cell steps into another tile)
//   node      : ptr = perimeter node id while unresolved; when done, the GLOBAL flat index of
//               the river cell, or FH_DEAD
#define FHT_DEAD 0xFFFFFFFFu
#define FHT_DONE 0x8000u
#define FHT_CAP 20000u
#define K_RIVER 1u
#define K_DEAD 2u
#define K_EXIT 3u

__device__ __forceinline__ unsigned long long fht_pack(uint32_t ptr, uint32_t nd, uint32_t ncf) {
  return ((unsigned long long)ptr << 32) | ((unsigned long long)nd << 16) | (unsigned long long)ncf;
}

// stage fdr (with a one-cell halo ring) and the river mask; build and resolve the in-tile words.
struct FhTile {
  uint8_t *s_fdr;   // [NT]
  uint8_t *s_halo;  // [2 * (TW + 2) + 2 * TH]: top row, bottom row, left col, right col
  unsigned long long *s_st;  // [NT]
};

__device__ __forceinline__ uint32_t fht_fdr_at(const FhTile &T, int ly, int lx) {
  if (ly >= 0 && ly < TH && lx >= 0 && lx < TW) return T.s_fdr[ly * TW + lx];
  if (ly < 0) return T.s_halo[lx + 1];
  if (ly >= TH) return T.s_halo[(TW + 2) + lx + 1];
  if (lx < 0) return T.s_halo[2 * (TW + 2) + ly];
  return T.s_halo[2 * (TW + 2) + TH + ly];
}

__device__ __forceinline__ void fht_solve_tile(const FhTile &T, const uint8_t *__restrict__ fdr,
                                               const int8_t *__restrict__ river, int H, int W, int y0,
                                               int x0) {
  dt_tile_load_fdr(fdr, H, W, y0, x0, T.s_fdr);
  for (int i = threadIdx.x; i < 2 * (TW + 2) + 2 * TH; i += 256) {
    int gy, gx;
    if (i < TW + 2) { gy = y0 - 1; gx = x0 - 1 + i; }
    else if (i < 2 * (TW + 2)) { gy = y0 + TH; gx = x0 - 1 + (i - (TW + 2)); }
    else if (i < 2 * (TW + 2) + TH) { gy = y0 + (i - 2 * (TW + 2)); gx = x0 - 1; }
    else { gy = y0 + (i - 2 * (TW + 2) - TH); gx = x0 + TW; }
    uint8_t v = 0;
    if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = fdr[(size_t)gy * W + gx];
    T.s_halo[i] = v;
  }
  // river mask of my 16 cells (row-contiguous per wave)
  uint32_t riv = 0;
  for (int j = 0; j < CPT; j++) {
    int c = threadIdx.x + 256 * j;
    int gy = y0 + c / TW, gx = x0 + c % TW;
    if (gy < H && gx < W && river[(size_t)gy * W + gx] == 1) riv |= 1u << j;
  }
  __syncthreads();
  for (int j = 0; j < CPT; j++) {
    int c = threadIdx.x + 256 * j;
    int ly = c / TW, lx = c % TW;
    int gy = y0 + ly, gx = x0 + lx;
    uint32_t code = T.s_fdr[c];
    unsigned long long s;
    if (gy >= H || gx >= W || code == 0u) {
      s = fht_pack((uint32_t)c | (K_DEAD << 12), 0, FHT_DONE);  // flowhand.py:601
    } else if ((riv >> j) & 1u) {
      s = fht_pack((uint32_t)c | (K_RIVER << 12), 0, FHT_DONE);  // flowhand.py:609-612
    } else if (!dt_d8_valid(code)) {
      s = fht_pack((uint32_t)c | (K_DEAD << 12), 0, FHT_DONE);  // non-D8 code: revisit test :830
    } else {
      int dy, dx;
      dt_d8_delta(code, dy, dx);
      int ty = gy + dy, tx = gx + dx;
      if (ty < 0 || ty >= H || tx < 0 || tx >= W || fht_fdr_at(T, ly + dy, lx + dx) == 0u) {
        s = fht_pack((uint32_t)c | (K_DEAD << 12), 0, FHT_DONE);  // raster exit / arrival on fdr==0
      } else if (ly + dy < 0 || ly + dy >= TH || lx + dx < 0 || lx + dx >= TW) {
        s = fht_pack((uint32_t)c | (K_EXIT << 12), 0, FHT_DONE);  // the step itself is added by the user
      } else {
        bool diag = dy != 0 && dx != 0;
        s = fht_pack((uint32_t)((ly + dy) * TW + lx + dx), diag ? 1u : 0u, diag ? 0u : 1u);
      }
    }
    T.s_st[c] = s;
  }
  __syncthreads();
  // pointer doubling in place; an in-tile cycle doubles its counts until they exceed the cap
  for (int round = 0; round < 16; round++) {
    int changed = 0;
    for (int j = 0; j < CPT; j++) {
      int c = threadIdx.x + 256 * j;
      unsigned long long s = T.s_st[c];
      uint32_t ncf = (uint32_t)(s & 0xFFFFu);
      if (ncf & FHT_DONE) continue;
      uint32_t ptr = (uint32_t)(s >> 32), nd = (uint32_t)((s >> 16) & 0xFFFFu);
      unsigned long long t = T.s_st[ptr & 0xFFFu];
      uint32_t tncf = (uint32_t)(t & 0xFFFFu);
      uint32_t nnc = ncf + (tncf & 0x7FFFu), nnd = nd + (uint32_t)((t >> 16) & 0xFFFFu);
      unsigned long long o;
      if (nnc + nnd > FHT_CAP) o = fht_pack((uint32_t)c | (K_DEAD << 12), 0, FHT_DONE);
      else o = fht_pack((uint32_t)(t >> 32), nnd, nnc | (tncf & FHT_DONE));
      T.s_st[c] = o;
      changed = 1;
    }
    if (!__syncthreads_or(changed)) break;
  }
  __syncthreads();
}

// pass 1: perimeter node words
__global__ __launch_bounds__(256) void k_fh_tile1(const uint8_t *__restrict__ fdr,
                                                 const int8_t *__restrict__ river, int H, int W,
                                                 int tiles_x, unsigned long long *__restrict__ nodes) {
  __shared__ __attribute__((aligned(16))) uint8_t s_fdr[NT];
  __shared__ uint8_t s_halo[2 * (TW + 2) + 2 * TH];
  __shared__ unsigned long long s_st[NT];
  const int tile = blockIdx.x;
  const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
  const int y0 = ty * TH, x0 = tx * TW;
  FhTile T{s_fdr, s_halo, s_st};
  fht_solve_tile(T, fdr, river, H, W, y0, x0);
  if (threadIdx.x < PS) {
    int ly, lx;
    dt_cell_of_slot(threadIdx.x, ly, lx);
    unsigned long long s = s_st[ly * TW + lx];
    uint32_t ptr = (uint32_t)(s >> 32), nd = (uint32_t)((s >> 16) & 0xFFFFu);
    uint32_t ncf = (uint32_t)(s & 0xFFFFu), nc = ncf & 0x7FFFu;
    uint32_t kind = (ptr >> 12) & 3u, f = ptr & 0xFFFu;
    unsigned long long o = fht_pack(FHT_DEAD, 0, FHT_DONE);
    if ((ncf & FHT_DONE) && kind == K_RIVER) {
      o = fht_pack((uint32_t)((y0 + (int)f / TW) * W + x0 + (int)f % TW), nd, nc | FHT_DONE);
    } else if ((ncf & FHT_DONE) && kind == K_EXIT) {
      int fy = (int)f / TW, fx = (int)f % TW, dy, dx;
      dt_d8_delta(s_fdr[f], dy, dx);
      int gy = y0 + fy + dy, gx = x0 + fx + dx;
      int t2y = gy / TH, t2x = gx / TW;
      uint32_t node = (uint32_t)((t2y * tiles_x + t2x) * PS + dt_slot_of(gy - t2y * TH, gx - t2x * TW));
      bool diag = dy != 0 && dx != 0;
      o = fht_pack(node, nd + (diag ? 1u : 0u), nc + (diag ? 0u : 1u));
    }
    nodes[(size_t)tile * PS + threadIdx.x] = o;
  }
}

// pass 3: resolved node words -> rasters
__global__ __launch_bounds__(256) void k_fh_tile3(const uint8_t *__restrict__ fdr,
                                                 const int8_t *__restrict__ river,
                                                 const float *__restrict__ dem,
                                                 const int32_t *__restrict__ acc32, int H, int W,
                                                 int tiles_x, const unsigned long long *__restrict__ nodes,
                                                 double px, float *__restrict__ fdist,
                                                 int32_t *__restrict__ idx32, float *__restrict__ hand,
                                                 int32_t *__restrict__ a_river) {
  __shared__ __attribute__((aligned(16))) uint8_t s_fdr[NT];
  __shared__ uint8_t s_halo[2 * (TW + 2) + 2 * TH];
  __shared__ unsigned long long s_st[NT];
  __shared__ unsigned long long s_x[PS];  // resolved word of the node each exit cell steps onto (+ the step)
  const int tile = blockIdx.x;
  const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
  const int y0 = ty * TH, x0 = tx * TW;
  FhTile T{s_fdr, s_halo, s_st};
  fht_solve_tile(T, fdr, river, H, W, y0, x0);
  if (threadIdx.x < PS) {
    int ly, lx;
    dt_cell_of_slot(threadIdx.x, ly, lx);
    int f = ly * TW + lx;
    unsigned long long s = s_st[f];
    unsigned long long o = fht_pack(FHT_DEAD, 0, FHT_DONE);
    // only exit cells (a finished word pointing at itself with kind EXIT) are looked up
    if ((uint32_t)(s >> 32) == ((uint32_t)f | (K_EXIT << 12))) {
      int dy, dx;
      dt_d8_delta(s_fdr[f], dy, dx);
      int gy = y0 + ly + dy, gx = x0 + lx + dx;
      int t2y = gy / TH, t2x = gx / TW;
      size_t node = (size_t)(t2y * tiles_x + t2x) * PS + dt_slot_of(gy - t2y * TH, gx - t2x * TW);
      unsigned long long ns = nodes[node];
      uint32_t nptr = (uint32_t)(ns >> 32), nnd = (uint32_t)((ns >> 16) & 0xFFFFu);
      uint32_t nncf = (uint32_t)(ns & 0xFFFFu);
      bool diag = dy != 0 && dx != 0;
      // not done after all rounds == longer than the cap
      if ((nncf & FHT_DONE) && nptr != FHT_DEAD)
        o = fht_pack(nptr, nnd + (diag ? 1u : 0u), ((nncf & 0x7FFFu) + (diag ? 0u : 1u)) | FHT_DONE);
    }
    s_x[threadIdx.x] = o;
  }
  __syncthreads();
  const double dcard = px, ddiag = px * sqrt(2.0);
  for (int j = 0; j < CPT; j++) {
    int c = threadIdx.x + 256 * j;
    int gy = y0 + c / TW, gx = x0 + c % TW;
    if (gy >= H || gx >= W) continue;
    unsigned long long s = s_st[c];
    uint32_t ptr = (uint32_t)(s >> 32), nd = (uint32_t)((s >> 16) & 0xFFFFu);
    uint32_t ncf = (uint32_t)(s & 0xFFFFu), nc = ncf & 0x7FFFu;
    uint32_t kind = (ptr >> 12) & 3u, f = ptr & 0xFFFu;
    bool ok = false;
    uint32_t ridx = 0;
    if (ncf & FHT_DONE) {
      if (kind == K_RIVER) {
        ok = true;
        ridx = (uint32_t)((y0 + (int)f / TW) * W + x0 + (int)f % TW);
      } else if (kind == K_EXIT) {
        unsigned long long xs = s_x[dt_slot_of((int)f / TW, (int)f % TW)];
        uint32_t xptr = (uint32_t)(xs >> 32);
        if (xptr != FHT_DEAD) {
          nc += (uint32_t)(xs & 0x7FFFu);
          nd += (uint32_t)((xs >> 16) & 0xFFFFu);
          ok = nc + nd <= FHT_CAP;  // flowhand.py:834-837
          ridx = xptr;
        }
      }
    }
    size_t o = (size_t)gy * W + gx;
    if (fdist) fdist[o] = ok ? (float)(dcard * (double)nc + ddiag * (double)nd) : DT_NODATA;
    if (idx32) idx32[o] = ok ? (int32_t)ridx : -100;
    if (hand) {
      float h = DT_NODATA, z = dem[o];
      if (z != DT_NODATA && ok) {  // flowhand.py:436
        h = z - dem[ridx];
        if (h < 0.0f && h != DT_NODATA) h = 0.0f;  // flowhand.py:438
      }
      hand[o] = h;
    }
    if (a_river) a_river[o] = ok ? acc32[ridx] : acc32[0];  // gfi.py:141-143
  }
}

// pass 2: pointer doubling over the perimeter nodes (same word format as the v1 raster kernel)
__global__ __launch_bounds__(256) void k_fh_node_jump(unsigned long long *__restrict__ state, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  unsigned long long s = __hip_atomic_load(&state[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  uint32_t ncf = (uint32_t)(s & 0xFFFFu);
  if (ncf & FHT_DONE) return;
  uint32_t ptr = (uint32_t)(s >> 32), nd = (uint32_t)((s >> 16) & 0xFFFFu);
  unsigned long long t = __hip_atomic_load(&state[ptr], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  uint32_t tptr = (uint32_t)(t >> 32), tnd = (uint32_t)((t >> 16) & 0xFFFFu);
  uint32_t tncf = (uint32_t)(t & 0xFFFFu);
  uint32_t nnc = ncf + (tncf & 0x7FFFu), nnd = nd + tnd;
  unsigned long long o;
  if (tptr == FHT_DEAD || nnc + nnd > FHT_CAP) o = fht_pack(FHT_DEAD, 0, FHT_DONE);
  else o = fht_pack(tptr, nnd, nnc | (tncf & FHT_DONE));
  __hip_atomic_store(&state[i], o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

size_t dt_flowhand_tiled_scratch(int64_t H, int64_t W) {
  int64_t ntiles = ((W + TW - 1) / TW) * ((H + TH - 1) / TH);
  return dt_align256((size_t)ntiles * PS * 8) + 256;
}

int dt_launch_flowhand_tiled(hipStream_t s, const float *dem, const uint8_t *fdr, const int8_t *river,
                             const int32_t *acc32, int64_t H, int64_t W, double px, void *scratch,
                             size_t scratch_bytes, float *fdist, int32_t *idx32, float *hand,
                             int32_t *a_river) {
  if (H == 0 || W == 0) return DT_OK;
  int tiles_x = (int)((W + TW - 1) / TW), tiles_y = (int)((H + TH - 1) / TH);
  int64_t ntiles = (int64_t)tiles_x * tiles_y;
  int64_t nnodes = ntiles * PS;
  DT_REQUIRE(nnodes < 0xFFFFFFF0ll, "raster too large for one device tile");
  DT_REQUIRE(scratch_bytes >= dt_flowhand_tiled_scratch(H, W), "scratch too small");
  unsigned long long *nodes = (unsigned long long *)scratch;
  dim3 gt((unsigned)ntiles), b(256), gn((unsigned)((nnodes + 255) / 256));
  hipLaunchKernelGGL(k_fh_tile1, gt, b, 0, s, fdr, river, (int)H, (int)W, tiles_x, nodes);
  // 15 rounds resolve every chain of <= 20000 moves (each node hop is >= 1 move; 2^15 > 20000)
  for (int r = 0; r < 15; r++) hipLaunchKernelGGL(k_fh_node_jump, gn, b, 0, s, nodes, nnodes);
  hipLaunchKernelGGL(k_fh_tile3, gt, b, 0, s, fdr, river, dem, acc32, (int)H, (int)W, tiles_x, nodes, px,
                     fdist, idx32, hand, a_river);
  return DT_OK;
}
