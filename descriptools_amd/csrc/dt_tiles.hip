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
//
// The same machinery is the multi-GPU exchange format: every kernel works on a DtWin -- a core
// window inside a larger (global) raster -- and a path that leaves the core into another rank's
// window ends on a "rank exit".  Pass 2 then has a stage between ranks (descriptools_amd/tiling.py):
// each rank publishes one summary row per cell of its core ring, the small rank-level graph is
// resolved redundantly by everybody, and the result is injected back (flow accumulation: external
// inflow per ring cell; HAND: the resolved river cell of every rank exit).
//
// Integer accumulation is order-independent and path counts are integers, so results are
// bit-identical to the v1 kernels and to the oracle for any tiling.
#include "dt_common.h"
#include "dt_kernels.h"
#include "dt_math.h"

#define TW 64
#define TH 64
#define NT (TW * TH)
#define PS (2 * TW + 2 * (TH - 2)) /* perimeter slots per tile: 252 */
#define CPT (NT / 256)             /* cells per thread: 16 */

#define NX_SINK 0xFFFFu  /* no D8 successor inside the global raster */
#define NX_EXIT 0xFFFEu  /* successor is in the core but in another tile */
#define NX_REXIT 0xFFFDu /* successor is outside the core, inside the global raster (another rank) */
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
// node id of the perimeter cell at core coordinates (y, x)
__device__ __forceinline__ uint32_t dt_node_of(int y, int x, int tiles_x) {
  int ty = y / TH, tx = x / TW;
  return (uint32_t)((ty * tiles_x + tx) * PS + dt_slot_of(y - ty * TH, x - tx * TW));
}

// Workgroup -> tile.  Workgroup ids congruent mod 8 share an XCD (and its L2): consecutive workgroups of one XCD
// take horizontally adjacent tiles, two by two.  A tile row of a BYTE raster is 64 bytes, half a 128-byte line:
// with the identity map the two halves went to two different XCDs and the line was fetched from HBM twice
// (PMC: 4.0 B/cell fetched by the passes that read 1-2 B/cell of direction codes / river mask).
__device__ __forceinline__ int dt_tile_of_block(int b, int ntiles) {
  const int full = ntiles & ~15;  // whole groups of 16 tiles; the tail keeps the identity
  if (b >= full) return b;
  const int xcd = b & 7, j = b >> 3;
  return ((j >> 1) * 8 + xcd) * 2 + (j & 1);
}

// ---- tile staging -----------------------------------------------------------------------------
// direction codes of the tile's core cells into LDS (0 outside the core): 256 threads x 16 bytes, one 64-byte row
// per 4 threads.  Split into the load and the LDS store so that a kernel can have several tile loads in flight
// before it waits for the first (dt_tile_load_fdr = both at once).
__device__ __forceinline__ uint4 dt_tile_fetch16(const uint8_t *__restrict__ fdr, const DtWin &w, int y0, int x0) {
  int t = threadIdx.x;
  int r = t >> 2, c = (t & 3) * 16;
  int gy = y0 + r, gx = x0 + c;
  uint4 v = make_uint4(0, 0, 0, 0);
  if (gy < w.H) {
    const uint8_t *p = fdr + (long long)gy * w.ld + gx;
    if (gx + 15 < w.W && (((uintptr_t)p) & 15) == 0) {
      v = *reinterpret_cast<const uint4 *>(p);
    } else {
      uint32_t w4[4] = {0, 0, 0, 0};
#pragma unroll
      for (int k = 0; k < 16; k++)
        if (gx + k < w.W) w4[k >> 2] |= (uint32_t)p[k] << (8 * (k & 3));
      v = make_uint4(w4[0], w4[1], w4[2], w4[3]);
    }
  }
  return v;
}
__device__ __forceinline__ void dt_tile_put16(uint8_t *s_fdr, uint4 v) {
  int t = threadIdx.x;
  *reinterpret_cast<uint4 *>(&s_fdr[(t >> 2) * TW + (t & 3) * 16]) = v;
}
__device__ __forceinline__ void dt_tile_load_fdr(const uint8_t *__restrict__ fdr, const DtWin &w, int y0,
                                                 int x0, uint8_t *s_fdr) {
  dt_tile_put16(s_fdr, dt_tile_fetch16(fdr, w, y0, x0));
}

// local successor of tile cell (ly, lx): local index, NX_EXIT, NX_REXIT or NX_SINK
__device__ __forceinline__ uint32_t dt_tile_next(uint32_t code, int ly, int lx, int y0, int x0,
                                                const DtWin &w) {
  int y = y0 + ly, x = x0 + lx;
  if (y >= w.H || x >= w.W || !dt_d8_valid(code)) return NX_SINK;
  int dy, dx;
  dt_d8_delta(code, dy, dx);
  int ty = y + dy, tx = x + dx;
  if (!dt_in_core(w, ty, tx)) return dt_in_global(w, ty, tx) ? NX_REXIT : NX_SINK;
  int ny = ly + dy, nx = lx + dx;
  if (ny < 0 || ny >= TH || nx < 0 || nx >= TW) return NX_EXIT;
  return (uint32_t)(ny * TW + nx);
}

// the same for a tile whose cells and their successors all lie inside the core (block-uniform test
// dt_tile_interior): validity, the D8 delta and the tile bounds are all that is left
__device__ __forceinline__ bool dt_tile_interior(const DtWin &w, int y0, int x0) {
  return y0 >= 1 && x0 >= 1 && y0 + TH + 1 <= w.H && x0 + TW + 1 <= w.W;
}
__device__ __forceinline__ uint32_t dt_tile_next_interior(uint32_t code, int ly, int lx) {
  if (!dt_d8_valid(code)) return NX_SINK;
  int dy, dx;
  dt_d8_delta(code, dy, dx);
  uint32_t ny = (uint32_t)(ly + dy), nx = (uint32_t)(lx + dx);
  return (ny < (uint32_t)TH && nx < (uint32_t)TW) ? ny * TW + nx : NX_EXIT;
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

// LDS holds what OTHER lanes need, one 32-bit word per cell: the cell's pointer word (idx:12 | PT_EXIT | PT_ALIVE) in
// the high half, what it receives in a round in the low half.  The scatter is a RETURNING atomic add whose old value
// carries the target's pointer word -- the add and the gather of the jump are one LDS instruction (round 3; with
// separate pointer / receive arrays and two random accesses per cell and round the pass took 3-5 % longer).  A cell's
// running sum is read and written by its owner only and lives in a register; each lane owns PAIRS of adjacent cells
// (2 (t + 256 j), + 1): its two words move as one 64-bit access.  On an in-tile D8 cycle the sums are garbage (such
// cells end up in s_cyc).  The receive field cannot carry into the
// pointer: off cycles a round delivers <= 4096 to a cell (the senders' subtrees are disjoint), and what a cell
// sends is clamped to 4096 (only cells ON an in-tile cycle ever exceed it, one such sender per target and round),
// so a round delivers <= 8192 < 2^16.  The owner reads and rewrites its pair of words (one conflict-free 64-bit
// access each way) between the barriers.
__device__ __forceinline__ void dt_tile_sums_packed(uint32_t *s_word, uint8_t *s_cyc, uint32_t (&va)[CPT / 2],
                                                    uint32_t (&vb)[CPT / 2]) {
  uint32_t P[CPT / 2];  // my two pointer words: cell a | cell b << 16
  unsigned long long *s_word2 = reinterpret_cast<unsigned long long *>(s_word);
#pragma unroll
  for (int j = 0; j < CPT / 2; j++) {
    unsigned long long r = s_word2[threadIdx.x + 256 * j];
    P[j] = ((uint32_t)r >> 16) | ((uint32_t)(r >> 32) & 0xFFFF0000u);
    va[j] = vb[j] = 1u;
  }
  for (int round = 0; round < 12; round++) {
    uint32_t jumped = 0;  // bit j: a cell of pair j jumped this round (its pointer word changed)
#pragma unroll
    for (int j = 0; j < CPT / 2; j++) {
      const uint32_t p = P[j];
      if (p & PT_ALIVE) {
        P[j] = (P[j] & 0xFFFF0000u) | (atomicAdd(&s_word[p & PT_IDX], min(va[j], (uint32_t)NT)) >> 16);
        jumped |= 1u << j;
      }
      if (p & (PT_ALIVE << 16)) {
        P[j] = (P[j] & 0xFFFFu) | (atomicAdd(&s_word[(p >> 16) & PT_IDX], min(vb[j], (uint32_t)NT)) & 0xFFFF0000u);
        jumped |= 1u << j;
      }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < CPT / 2; j++) {
      int c2 = threadIdx.x + 256 * j;
      unsigned long long r = s_word2[c2];
      const uint32_t ra = (uint32_t)r & 0xFFFFu, rb = (uint32_t)(r >> 32) & 0xFFFFu;
      va[j] += ra;
      vb[j] += rb;
      // a pair that neither jumped nor received keeps its words as they are (most pairs, in the late rounds)
      if (((jumped >> j) & 1u) | ra | rb)
        s_word2[c2] = ((unsigned long long)(P[j] & 0xFFFF0000u) << 32) | (unsigned long long)(P[j] << 16);
    }
    // Done when no cell is alive AFTER this round's jumps (what arrived in this round has just been added).  The test
    // was "nobody jumped in this round": one more round of reads, writes and two barriers that found nothing to do --
    // an eighth of the rounds of the benchmark terrain (7.9 per tile, the cells all at their ends after 7).
    uint32_t alive = 0;
#pragma unroll
    for (int j = 0; j < CPT / 2; j++) alive |= P[j] & (PT_ALIVE | (PT_ALIVE << 16));
    if (!__syncthreads_or((int)(alive != 0u))) break;
  }
  // still alive after 2^12 moves: the path never ends inside the tile -> in-tile cycle
#pragma unroll
  for (int j = 0; j < CPT / 2; j++) {
    if (P[j] & PT_ALIVE) s_cyc[P[j] & PT_IDX] = 1;
    if (P[j] & (PT_ALIVE << 16)) s_cyc[(P[j] >> 16) & PT_IDX] = 1;
  }
  // the final sums beside the final pointers, where the perimeter lanes read both with one load
#pragma unroll
  for (int j = 0; j < CPT / 2; j++)
    s_word2[threadIdx.x + 256 * j] = ((unsigned long long)((P[j] & 0xFFFF0000u) | (vb[j] & 0xFFFFu)) << 32) |
                                     (unsigned long long)((P[j] << 16) | (va[j] & 0xFFFFu));
  __syncthreads();
}

// countdown word of an exit node in the perimeter graph: pending feeders << 54 | sum of what arrived so far
#define FA_NONE 0xFFFFFFFFu
#define FA2_SH 54 /* pending count in bits 54-63: a tile exit has at most 260 feeders (the cells around the tile) */
#define FA2_MASK ((1ull << FA2_SH) - 1ull)
#define FA_CYCLE (1ull << 63) /* ext flag: this entry cell is fed by a cross-tile D8 cycle */
#define FA_VALUE(e) ((e) & ~FA_CYCLE)
// perimeter record (8 bytes):  W:32 | xslot:16 | code:8 | flags:8
//   code  = the cell's D8 code when its successor is in another tile or rank (an "exit" cell)
//   flags = bit 0: rank exit (the successor is outside the core window)
//   W     = cells draining through the exit cell inside the tile, itself included
//   xslot = perimeter slot of the exit cell reached by a path entering the tile at this cell
//           (X_NONE when that path ends inside the tile)
#define REC_RANK_EXIT 1ull
__device__ __forceinline__ unsigned long long fa_rec(uint32_t W_, uint32_t xslot, uint32_t code,
                                                    uint32_t flags) {
  return ((unsigned long long)W_ << 32) | ((unsigned long long)(xslot & 0xFFFFu) << 16) |
         ((unsigned long long)(code & 0xFFu) << 8) | (unsigned long long)(flags & 0xFFu);
}
#define REC_W(r) ((uint32_t)((r) >> 32))
#define REC_XSLOT(r) ((uint32_t)(((r) >> 16) & 0xFFFFu))
#define REC_CODE(r) ((uint32_t)(((r) >> 8) & 0xFFu))

__global__ __launch_bounds__(256, 6) void k_fa_tile1(const uint8_t *__restrict__ fdr, DtWin w, int tiles_x,
                                                    unsigned long long *__restrict__ rec,
                                                    uint16_t *__restrict__ loc16,
                                                    unsigned long long *__restrict__ state,
                                                    unsigned long long *__restrict__ ext,
                                                    uint32_t *__restrict__ entry_of,
                                                    uint32_t *__restrict__ parent) {
  // 21.5 KiB of LDS: six tiles per CU.  s_word: one 32-bit word per cell (dt_tile_sums_packed); s_aux stages the
  // direction codes, then holds the cycle mask (4 KiB) and the pending counts (252 words); s_lut: code -> successor.
  __shared__ __attribute__((aligned(16))) uint32_t s_word[NT];
  __shared__ __attribute__((aligned(16))) uint32_t s_aux[NT / 4 + 256];
  __shared__ uint16_t s_lut[256];  // D8 code -> successor's offset + 128 (0: not a D8 code)
  uint8_t *s_fdr = reinterpret_cast<uint8_t *>(s_aux);
  const int tile = dt_tile_of_block((int)blockIdx.x, (int)gridDim.x);
  const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
  const int y0 = ty * TH, x0 = tx * TW;
  // The tile's codes and, for the perimeter lanes, the codes of the <= 5 neighbours outside the tile: all loads in
  // flight together, before the first barrier (fetched one by one, each behind the test of the previous, the
  // neighbour codes were five dependent memory round trips on the workgroup's path to its barrier).
  const uint4 v_fdr = dt_tile_fetch16(fdr, w, y0, x0);
  const int8_t qdy[8] = {-1, -1, -1, 0, 0, 1, 1, 1}, qdx[8] = {-1, 0, 1, -1, 1, -1, 0, 1};  // NW N NE W E SW S SE
  uint32_t c2[8];
  int ply = 0, plx = 0;
  if (threadIdx.x < PS) dt_cell_of_slot(threadIdx.x, ply, plx);
  {
    const int y = y0 + ply, x = x0 + plx;
    const bool live = threadIdx.x < PS && y < w.H && x < w.W;
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const int dy = qdy[q], dx = qdx[q];
      const int ny = ply + dy, nx = plx + dx;
      const bool inside = ny >= 0 && ny < TH && nx >= 0 && nx < TW;  // my own tile
      // other ranks' cells arrive as injected inflow
      c2[q] = (live && !inside && dt_in_core(w, y + dy, x + dx)) ? (uint32_t)fdr[(long long)(y + dy) * w.ld + x + dx] : 0u;
    }
  }
  {
    uint32_t code = threadIdx.x, e = 0u;
    if (dt_d8_valid(code)) {
      int dy, dx;
      dt_d8_delta(code, dy, dx);
      e = (uint32_t)(dy * TW + dx + 128);
    }
    s_lut[code] = (uint16_t)e;
  }
  dt_tile_put16(s_fdr, v_fdr);
  __syncthreads();
  // Every cell's pointer word: terminals point at themselves; an exit terminal carries PT_EXIT, which every cell
  // whose in-tile path ends there inherits through the jumps.  A lane sets up the 4 x 4 cells 4 (t + 256 u) + k:
  // codes and words move as 32 / 128-bit LDS accesses.
  if (dt_tile_interior(w, y0, x0)) {
    // all cells and successors inside the core: the successor left the tile iff its index left [0, 4096) or its
    // column wrapped (0 <-> 63: it differs from mine by 63 instead of <= 1)
    const uint32_t lx0 = (4u * threadIdx.x) & 63u;
#pragma unroll
    for (int u = 0; u < NT / 4 / 256; u++) {
      const int c0 = 4 * (threadIdx.x + 256 * u);
      const uint32_t codes = *reinterpret_cast<const uint32_t *>(&s_fdr[c0]);
      uint32_t pw[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const uint32_t c = (uint32_t)(c0 + k), e = s_lut[(codes >> (8 * k)) & 0xFFu];
        const uint32_t n = c + e - 128u;
        const bool out = n > (uint32_t)(NT - 1) || ((n & 63u) - (lx0 + (uint32_t)k - 1u)) > 2u;
        pw[k] = (e == 0u ? c : (out ? (c | PT_EXIT) : (n | PT_ALIVE))) << 16;
      }
      *reinterpret_cast<uint4 *>(&s_word[c0]) = make_uint4(pw[0], pw[1], pw[2], pw[3]);
    }
  } else {
    for (int u = 0; u < NT / 4 / 256; u++) {
      const int c0 = 4 * (threadIdx.x + 256 * u);
      for (int k = 0; k < 4; k++) {
        const int c = c0 + k;
        const uint32_t n = dt_tile_next(s_fdr[c], c / TW, c % TW, y0, x0, w);
        const bool ex = (n == NX_EXIT || n == NX_REXIT);
        s_word[c] = (n < NT ? (n | PT_ALIVE) : ((uint32_t)c | (ex ? PT_EXIT : 0u))) << 16;
      }
    }
  }
  uint32_t my_code = 0, my_flags = 0;  // D8 code of my perimeter cell when it is an exit cell
  uint32_t feeders = 0;                // exits of neighbouring tiles (inside the core) that step onto my cell
  uint32_t fnode[8];                   // their node ids, by neighbour (a perimeter cell has <= 5 outside the tile)
  if (threadIdx.x < PS) {
    const int ly = ply, lx = plx;
    uint32_t code = s_fdr[ly * TW + lx];
    uint32_t n = dt_tile_next(code, ly, lx, y0, x0, w);
    if (n == NX_EXIT || n == NX_REXIT) my_code = code;
    if (n == NX_REXIT) my_flags = (uint32_t)REC_RANK_EXIT;
    // The links of the perimeter graph, without a link pass and its global atomics: every exit that feeds this
    // tile enters at a perimeter cell, from one of its <= 5 neighbours outside the tile.  The tile that is
    // ENTERED writes the feeder's entry and parent (each exit has one successor: no two writers) and counts
    // its own exits' pending feeders in LDS.
    const int y = y0 + ly, x = x0 + lx;
#pragma unroll
    for (int q = 0; q < 8; q++) fnode[q] = FA_NONE;
    if (y < w.H && x < w.W) {
      // neighbour q drains into me iff its code points back at me
      const uint8_t back[8] = {2, 4, 8, 1, 16, 128, 64, 32};
#pragma unroll
      for (int q = 0; q < 8; q++)
        if (c2[q] == (uint32_t)back[q]) {
          fnode[q] = dt_node_of(y + qdy[q], x + qdx[q], tiles_x);
          feeders++;
        }
    }
  }
  __syncthreads();
  // the staged codes have been consumed: cycle mask and pending counts start at zero (the rounds' first barrier comes
  // before anything reads them)
  uint8_t *s_cyc = reinterpret_cast<uint8_t *>(s_aux);
  uint32_t *s_pend = s_aux + NT / 4;  // 252 words above the cycle mask
  reinterpret_cast<uint4 *>(s_aux)[threadIdx.x] = make_uint4(0, 0, 0, 0);
  s_aux[NT / 4 + threadIdx.x] = 0u;
  uint32_t va[CPT / 2], vb[CPT / 2];  // running sums of my cells 2 (t + 256 j) and + 1
  dt_tile_sums_packed(s_word, s_cyc, va, vb);
  if (threadIdx.x < PS) {
    int ly, lx;
    dt_cell_of_slot(threadIdx.x, ly, lx);
    int c = ly * TW + lx;
    const uint32_t pv = s_word[c];  // final pointer word << 16 | final sum (<= 4096 off cycles)
    uint32_t p = pv >> 16;
    uint32_t xs = X_NONE;
    if (!(p & PT_ALIVE) && (p & PT_EXIT)) {
      uint32_t f = p & PT_IDX;
      xs = (uint32_t)dt_slot_of((int)f / TW, (int)f % TW);
    }
    rec[(size_t)tile * PS + threadIdx.x] = fa_rec(my_code ? (pv & 0xFFFFu) : 0u, xs, my_code, my_flags);
    // what enters at my cell waits at the exit its in-tile path leads to
    if (xs != X_NONE && feeders) atomicAdd(&s_pend[xs], feeders);
    const uint32_t me = (uint32_t)tile * PS + threadIdx.x, par = xs != X_NONE ? (uint32_t)tile * PS + xs : FA_NONE;
#pragma unroll
    for (int k = 0; k < 8; k++)
      if (fnode[k] != FA_NONE) {
        entry_of[fnode[k]] = me;
        parent[fnode[k]] = par;
      }
    if (my_flags & (uint32_t)REC_RANK_EXIT) {  // my exit leaves the rank: nobody else writes its links
      entry_of[me] = FA_NONE;
      parent[me] = FA_NONE;
    }
  }
  __syncthreads();
  if (threadIdx.x < PS) {  // pending count of my exit (0 for the rest): k_fa_reduce's countdown word
    state[(size_t)tile * PS + threadIdx.x] = (unsigned long long)s_pend[threadIdx.x] << FA2_SH;
    ext[(size_t)tile * PS + threadIdx.x] = 0ull;  // inflow accumulator of my entry
  }
  // in-tile accumulation (upstream cells of this tile only, <= 4095: 2 bytes per cell, tile-major; 0xFFFF =
  // on an in-tile cycle); pass 3 adds what enters from outside
  // (a lane's pairs, straight from its registers: one 2-byte mask read and one 4-byte store per pair)
#pragma unroll
  for (int j = 0; j < CPT / 2; j++) {
    int c2 = threadIdx.x + 256 * j;
    uint32_t cy = *reinterpret_cast<const uint16_t *>(&s_cyc[2 * c2]);
    uint32_t a = (cy & 0xFFu) ? 0xFFFFu : (va[j] - 1u) & 0xFFFFu;
    uint32_t b = (cy & 0xFF00u) ? 0xFFFFu : (vb[j] - 1u) & 0xFFFFu;
    *reinterpret_cast<uint32_t *>(loc16 + (size_t)tile * NT + 2 * c2) = a | (b << 16);
  }
}

// perimeter graph: node id = tile * PS + slot.  entry_of[q] = the entry node exit q feeds (the neighbouring
// tile's perimeter cell its D8 step lands on), parent[q] = the exit node that entry's in-tile path leads to;
// rank exits have neither.  Both are written by pass 1 of the tile that is entered.
// countdown over the reduced forest; A(q) = W(q) + sum of A over the exit nodes feeding
// q's tile through entry cells whose in-tile path leads to q.  ext[entry] accumulates the inflow
// arriving at an entry cell from other tiles.
__global__ __launch_bounds__(256) void k_fa_reduce(const unsigned long long *__restrict__ rec, int64_t nnodes,
                                                  const uint32_t *__restrict__ entry_of,
                                                  const uint32_t *__restrict__ parent,
                                                  unsigned long long *__restrict__ state,
                                                  unsigned long long *__restrict__ ext) {
  int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (n >= nnodes) return;
  unsigned long long r = rec[n];
  if (REC_CODE(r) == 0) return;  // not an exit node
  if (state[n] != 0ull) return;  // not a source of the reduced forest (a source's word stays 0)
  unsigned long long A = REC_W(r);
  uint32_t e = entry_of[n], p = parent[n];
  for (int64_t it = 0; it < nnodes; it++) {
    if (e == FA_NONE) break;  // rank exit: its total stays in (rec, state)
    atomicAdd(&ext[e], A);
    if (p == FA_NONE) break;
    // the next hop's operands are fetched while the countdown atomic is in flight
    unsigned long long rp = rec[p];
    uint32_t ep = entry_of[p], pp = parent[p];
    unsigned long long old = atomicAdd(&state[p], A - (1ull << FA2_SH));
    if ((old >> FA2_SH) != 1ull) break;
    A = REC_W(rp) + (old & FA2_MASK) + A;
    e = ep;
    p = pp;
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
  // the countdown word first: it is 0-pending for every node that is not an exit and for every resolved exit, i.e.
  // for all but the handful this kernel is about -- one array streamed instead of three
  if ((state[n] >> FA2_SH) == 0ull) return;
  if (REC_CODE(rec[n]) == 0 || entry_of[n] == FA_NONE) return;
  atomicOr(&ext[entry_of[n]], FA_CYCLE);
}

// pass 3: the inflow that enters the tile at a perimeter cell p (ext[p], resolved by pass 2) drains
// through every cell of p's in-tile path: one lane per entry cell walks that path adding ext[p] to
// an LDS delta raster (integer adds: order-free), then delta is added to pass 1's in-tile counts.
// AccT = int32_t: the accumulation raster of a raster below 2^31 cells (a value that could reach 2^31 raises the
// context's overflow status instead of wrapping); AccT = int64_t: the reference's own dtype (Example/example.py:39),
// selected for multi-rank rasters of >= 2^31 cells, where a basin can exceed 32 bits.
template <bool HAS_DEM, bool W_RIVER, typename AccT>
__global__ __launch_bounds__(256, sizeof(AccT) == 8 ? 4 : 6) void k_fa_tile3(const uint8_t *__restrict__ fdr,
                                                    const float *__restrict__ dem, DtWin w, int tiles_x,
                                                    const unsigned long long *__restrict__ ext,
                                                    const uint16_t *__restrict__ loc16,
                                                    AccT *__restrict__ acc32, AccT river_thr,
                                                    int8_t *__restrict__ river, int *__restrict__ status) {
  constexpr bool WIDE = sizeof(AccT) == 8;
  // 25.5 KiB of LDS: six tiles per CU (the kernel is a latency chain of LDS walks).  The direction codes are
  // staged through the delta array; bit 31 of a delta (real inflow < 2^31) marks the cells of a cycle
  // spanning tiles.  Both arrays are indexed with rows 68 cells apart (P3 below): with 64, a step north or south
  // keeps the LDS bank, so the entry walks that trail one another down a north-south stem -- the common case on
  // tilted terrain -- serialised on one bank (76 % of the kernel's LDS cycles were bank conflicts).  The padded
  // index of the successor is what s_nxt holds, so a step is still one read; rows stay 16-byte aligned.
#define P3(c) ((uint32_t)(c) + (((uint32_t)(c) >> 6) << 2))
#define NT3 (TH * (TW + 4))
  __shared__ uint16_t s_nxt[NT3];
  __shared__ __attribute__((aligned(16))) uint32_t s_delta[NT3];
  uint8_t *s_fdr = reinterpret_cast<uint8_t *>(s_delta);
  const int tile = dt_tile_of_block((int)blockIdx.x, (int)gridDim.x);
  const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
  const int y0 = ty * TH, x0 = tx * TW;
  // every global load of the kernel is issued here, before the first wait: the tile's codes, the resolved inflow
  // of its entries, pass 1's counts and the heights
  const uint4 v_fdr = dt_tile_fetch16(fdr, w, y0, x0);
  unsigned long long e = 0ull;
  if (threadIdx.x < PS) e = ext[(size_t)tile * PS + threadIdx.x];
  // pass 1's counts and the heights are fetched now, so that their latency hides behind the serial walks.
  // Block-uniform fast form: whole 64-cell rows inside the core, 16-byte aligned rasters -> each lane owns 4
  // groups of 4 consecutive cells (8- and 16-byte loads, 16- and 4-byte stores)
  const bool vec = x0 + TW <= w.W && (w.ld & 3) == 0 && (((uintptr_t)acc32 & 15) == 0) &&
                   (!HAS_DEM || ((uintptr_t)dem & 15) == 0) && (!W_RIVER || ((uintptr_t)river & 3) == 0);
  constexpr int VPT = NT / 4 / 256;
  uint2 l4[VPT];
  float4 z4[VPT];
  int32_t av[CPT];
  float zv[CPT];
  if (vec) {
#pragma unroll
    for (int u = 0; u < VPT; u++) {
      int c = 4 * (threadIdx.x + 256 * u);
      int y = y0 + c / TW;
      l4[u] = *reinterpret_cast<const uint2 *>(loc16 + (size_t)tile * NT + c);
      z4[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (HAS_DEM && y < w.H) z4[u] = *reinterpret_cast<const float4 *>(dem + (long long)y * w.ld + x0 + c % TW);
    }
  } else {
#pragma unroll
    for (int j = 0; j < CPT; j++) {
      int c = threadIdx.x + 256 * j;
      int y = y0 + c / TW, x = x0 + c % TW;
      bool in = y < w.H && x < w.W;
      long long o = (long long)y * w.ld + x;
      uint32_t l16 = loc16[(size_t)tile * NT + c];
      av[j] = l16 == 0xFFFFu ? -100 : (int32_t)l16;
      zv[j] = (HAS_DEM && in) ? dem[o] : 0.0f;
    }
  }
  dt_tile_put16(s_fdr, v_fdr);
  // No value of this tile can exceed the inflow entering it plus its own 4096 cells.  A tile whose entries bring in
  // >= 2^31 - 4096 in total is "big" (only tiles with an entry of >= 2^22 pay for the 64-bit sum that decides it):
  // with the int32 raster it raises the context's overflow status (dt_ctx_status) instead of wrapping silently;
  // with the int64 raster its inflow is carried through the 32-bit LDS delta raster in two limbs (below).
  bool big = false;
  if (__syncthreads_or(!(e & FA_CYCLE) && FA_VALUE(e) >= (1ull << 22))) {
    __shared__ unsigned long long s_in;
    if (threadIdx.x == 0) s_in = 0ull;
    __syncthreads();
    if (e != 0ull && !(e & FA_CYCLE)) atomicAdd(&s_in, FA_VALUE(e));
    __syncthreads();
    const bool over = s_in >= (1ull << 31) - (unsigned long long)NT;
    if (WIDE) big = over;
    else if (threadIdx.x == 0 && over && status) atomicOr(status, DT_STATUS_ACC_OVERFLOW);
  }
  uint32_t nx[CPT];
  if (dt_tile_interior(w, y0, x0)) {
#pragma unroll
    for (int j = 0; j < CPT; j++) {
      int c = threadIdx.x + 256 * j;
      nx[j] = dt_tile_next_interior(s_fdr[c], c / TW, c % TW);
    }
  } else {
    for (int j = 0; j < CPT; j++) {
      int c = threadIdx.x + 256 * j;
      nx[j] = dt_tile_next(s_fdr[c], c / TW, c % TW, y0, x0, w);
    }
  }
  __syncthreads();
  for (int j = 0; j < CPT; j++) {
    int c = threadIdx.x + 256 * j;
    s_nxt[P3(c)] = (uint16_t)(nx[j] < NT ? P3(nx[j]) : nx[j]);  // NX_* sentinels stay (>= 0xFFFD > NT3)
    s_delta[P3(c)] = 0u;
  }
  __syncthreads();
  // one lane per entry cell walks its in-tile path.  Limbs of a big tile (int64 raster only): the low 23 bits of
  // every inflow in a first sweep, the rest in a second one over the same (cleared) delta raster -- a path collects
  // at most 252 entries, so neither sweep can reach bit 31, which stays the cycle mark.
  constexpr uint32_t LIMB = 23;
  auto walk = [&](uint32_t add, bool cyc) {
    int ly, lx;
    dt_cell_of_slot(threadIdx.x, ly, lx);
    uint32_t c = P3(ly * TW + lx);
    if (cyc) {  // fed by a D8 cycle spanning tiles: the path IS the cycle
      for (int it = 0; it < NT && c < NT3; it++) {
        atomicOr(&s_delta[c], 0x80000000u);
        c = s_nxt[c];
      }
    } else {
      for (int it = 0; it < NT && c < NT3; it++) {
        atomicAdd(&s_delta[c], add);
        c = s_nxt[c];
      }
    }
  };
  // the lane's 16 cells in the order of the stores below
  auto cell_of = [&](int i) -> int { return vec ? 4 * ((int)threadIdx.x + 256 * (i >> 2)) + (i & 3) : (int)threadIdx.x + 256 * i; };
  uint32_t dlo[WIDE ? CPT : 1];
  if (e != 0ull) {
    const bool cyc = (e & FA_CYCLE) != 0ull;
    const uint32_t add = (WIDE && big) ? (uint32_t)(FA_VALUE(e) & ((1ull << LIMB) - 1ull)) : (uint32_t)e;
    if (cyc || add) walk(add, cyc);
  }
  __syncthreads();
  if (WIDE && big) {  // block-uniform
#pragma unroll
    for (int i = 0; i < CPT; i++) dlo[WIDE ? i : 0] = s_delta[P3(cell_of(i))];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < CPT; i++) s_delta[P3(cell_of(i))] = 0u;
    __syncthreads();
    if (e != 0ull && !(e & FA_CYCLE) && (FA_VALUE(e) >> LIMB) != 0ull) walk((uint32_t)(FA_VALUE(e) >> LIMB), false);
    __syncthreads();
  }
  // inflow of cell i of this lane (and its cycle mark) from the delta raster word d
  auto inflow = [&](int i, uint32_t d, bool &cyc) -> unsigned long long {
    if (WIDE && big) {
      const uint32_t lo = dlo[WIDE ? i : 0];
      cyc = (lo & 0x80000000u) != 0u;
      return ((unsigned long long)d << LIMB) + (unsigned long long)(lo & 0x7FFFFFFFu);
    }
    cyc = (d & 0x80000000u) != 0u;
    return (unsigned long long)(d & 0x7FFFFFFFu);
  };
  auto finish = [&](uint32_t l16, unsigned long long d, bool cyc, float z) -> AccT {
    AccT v = l16 == 0xFFFFu ? (AccT)-100 : (AccT)l16;
    if (v != (AccT)-100) v += (AccT)d;
    if (cyc) v = (AccT)-100;
    if (HAS_DEM && z <= DT_NODATA) v = (AccT)-100;
    return v;
  };
  if (vec) {
#pragma unroll
    for (int u = 0; u < VPT; u++) {
      int c = 4 * (threadIdx.x + 256 * u);
      int y = y0 + c / TW;
      if (y >= w.H) continue;
      long long o = (long long)y * w.ld + x0 + c % TW;
      uint4 d = *reinterpret_cast<const uint4 *>(&s_delta[P3(c)]);
      bool c0, c1, c2, c3;
      const unsigned long long i0 = inflow(4 * u, d.x, c0), i1 = inflow(4 * u + 1, d.y, c1),
                               i2 = inflow(4 * u + 2, d.z, c2), i3 = inflow(4 * u + 3, d.w, c3);
      const AccT vx = finish(l4[u].x & 0xFFFFu, i0, c0, z4[u].x), vy = finish(l4[u].x >> 16, i1, c1, z4[u].y),
                 vz = finish(l4[u].y & 0xFFFFu, i2, c2, z4[u].z), vw = finish(l4[u].y >> 16, i3, c3, z4[u].w);
      if (WIDE) {
        *reinterpret_cast<longlong2 *>(acc32 + o) = make_longlong2((long long)vx, (long long)vy);
        *reinterpret_cast<longlong2 *>(acc32 + o + 2) = make_longlong2((long long)vz, (long long)vw);
      } else {
        *reinterpret_cast<int4 *>(acc32 + o) = make_int4((int)vx, (int)vy, (int)vz, (int)vw);
      }
      if (W_RIVER) {
        uint32_t r = (vx > river_thr ? 1u : 0u) | (vy > river_thr ? 0x100u : 0u) |
                     (vz > river_thr ? 0x10000u : 0u) | (vw > river_thr ? 0x1000000u : 0u);
        *reinterpret_cast<uint32_t *>(river + o) = r;
      }
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < CPT; j++) {
    int c = threadIdx.x + 256 * j;
    int y = y0 + c / TW, x = x0 + c % TW;
    if (y >= w.H || x >= w.W) continue;
    long long o = (long long)y * w.ld + x;
    bool cy;
    const unsigned long long in = inflow(j, s_delta[P3(c)], cy);
    AccT v = finish(av[j] == -100 ? 0xFFFFu : (uint32_t)av[j], in, cy, zv[j]);
    acc32[o] = v;
    if (W_RIVER) river[o] = v > river_thr ? 1 : 0;
  }
}
#undef P3
#undef NT3

// ---- rank level (multi-GPU) ------------------------------------------------------------------------
// nxt[n]: the entry node that follows perimeter node n on its path through the rank -- the node its tile's
// exit steps onto -- or that exit itself when it leaves the rank (bit 31: terminal), or FA_NONE (path ends).
#define J_TERM 0x80000000u
__global__ __launch_bounds__(256) void k_fa_nxt_init(const unsigned long long *__restrict__ rec, int64_t nnodes,
                                                    const uint32_t *__restrict__ entry_of,
                                                    const unsigned long long *__restrict__ state,
                                                    uint32_t *__restrict__ nxt) {
  int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (n >= nnodes) return;
  uint32_t xs = REC_XSLOT(rec[n]);
  uint32_t j = FA_NONE;
  if (xs != X_NONE) {
    uint32_t q = (uint32_t)(n / PS) * PS + xs;
    // an exit the local countdown never resolved sits on a D8 cycle spanning tiles of this rank: whatever
    // enters it dies there (its cells are -100, k_fa_poison)
    if (rec[q] & REC_RANK_EXIT) j = q | J_TERM;
    else if ((state[q] >> FA2_SH) == 0ull) j = entry_of[q];
  }
  nxt[n] = j;
}

// one row per cell of the core ring: A = cells of this rank draining out through the cell (0 unless
// it is a rank exit), code = its D8 code if rank exit, xr = ring index of the rank exit reached by a
// path ENTERING the rank at this cell (-1: ends inside), found by walking the successor list: 65 k lanes,
// each at most as many hops as its path crosses tiles (a pointer doubling over all 16.6 M perimeter nodes
// costs more than these few latency chains)
__global__ __launch_bounds__(256) void k_fa_rank_summary(DtWin w, int tiles_x,
                                                        const unsigned long long *__restrict__ rec,
                                                        const unsigned long long *__restrict__ state,
                                                        const uint32_t *__restrict__ nxt, int64_t nnodes,
                                                        int64_t P, int64_t *__restrict__ A,
                                                        int32_t *__restrict__ xr, uint8_t *__restrict__ code) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= P) return;
  int y, x;
  dt_perim_cell(w.H, w.W, i, y, x);
  // a ring cell of a ragged last tile row / column (H or W not a multiple of 64) is not on its TILE's perimeter:
  // it lies on the edge of the global raster (Layout), nothing enters or leaves the rank there
  if (dt_slot_of(y % TH, x % TW) < 0) {
    A[i] = 0;
    xr[i] = -1;
    code[i] = 0;
    return;
  }
  uint32_t n = dt_node_of(y, x, tiles_x);
  unsigned long long r = rec[n];
  bool rex = (r & REC_RANK_EXIT) != 0ull;
  A[i] = rex ? (int64_t)(REC_W(r) + (state[n] & FA2_MASK)) : 0;
  code[i] = rex ? (uint8_t)REC_CODE(r) : 0;
  uint32_t j = nxt[n];
  int32_t out = -1;
  for (int64_t it = 0; it < nnodes && j != FA_NONE; it++) {
    if (j & J_TERM) {
      uint32_t q = j & ~J_TERM;
      int tile = (int)(q / PS), slot = (int)(q - (uint32_t)tile * PS);
      int ly, lx;
      dt_cell_of_slot(slot, ly, lx);
      out = (int32_t)dt_perim_index(w.H, w.W, (tile / tiles_x) * TH + ly, (tile % tiles_x) * TW + lx);
      break;
    }
    j = nxt[j];
  }
  xr[i] = out;
}

// the inflow other ranks deliver at ring cell i drains through every tile entry on its way to the rank
// exit (or to where the path ends): one lane per ring cell walks the entry -> entry successor list adding
// it to ext[].  The local countdown's results stay valid (accumulation is linear in the inflow), so the
// perimeter graph is not solved a second time.
__global__ __launch_bounds__(256) void k_fa_propagate(DtWin w, int tiles_x, int64_t nnodes,
                                                     const uint32_t *__restrict__ nxt,
                                                     const unsigned long long *__restrict__ ext_perim, int64_t P,
                                                     unsigned long long *__restrict__ ext) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= P) return;
  unsigned long long v = ext_perim[i];
  if (v == 0ull) return;
  int y, x;
  dt_perim_cell(w.H, w.W, i, y, x);
  if (dt_slot_of(y % TH, x % TW) < 0) return;  // ragged raster edge: no perimeter node, nothing can arrive here
  uint32_t n = dt_node_of(y, x, tiles_x);
  const unsigned long long val = FA_VALUE(v), cyc = v & FA_CYCLE;
  for (int64_t it = 0; it < nnodes; it++) {
    if (val) atomicAdd(&ext[n], val);
    if (cyc) atomicOr(&ext[n], FA_CYCLE);
    n = nxt[n];
    if (n == FA_NONE || (n & J_TERM)) break;
  }
}

struct FaScratch {
  unsigned long long *rec, *state, *ext;
  uint32_t *entry_of, *parent, *nxt;
  uint16_t *loc16;  // pass 1's in-tile counts, tile-major
  int64_t nnodes, ntiles;
  int tiles_x;
};
static FaScratch fa_layout(const DtWin &w, void *scratch) {
  FaScratch f;
  f.tiles_x = (w.W + TW - 1) / TW;
  f.ntiles = (int64_t)f.tiles_x * ((w.H + TH - 1) / TH);
  f.nnodes = f.ntiles * PS;
  char *p = (char *)scratch;
  size_t n8 = dt_align256((size_t)f.nnodes * 8), n4 = dt_align256((size_t)f.nnodes * 4);
  f.rec = (unsigned long long *)p;  p += n8;
  f.state = (unsigned long long *)p;  p += n8;  // state, ext contiguous: one memset
  f.ext = (unsigned long long *)p;  p += n8;
  f.entry_of = (uint32_t *)p;  p += n4;
  f.parent = (uint32_t *)p;  p += n4;
  f.nxt = (uint32_t *)p;  p += n4;
  f.loc16 = (uint16_t *)p;
  return f;
}
size_t dt_flowacc_tiled_scratch(int64_t H, int64_t W) {
  int64_t ntiles = ((W + TW - 1) / TW) * ((H + TH - 1) / TH);
  size_t nn = (size_t)ntiles * PS;
  return dt_align256(nn * 8) * 3 + dt_align256(nn * 4) * 3 + 256 + dt_align256((size_t)ntiles * NT * 2);
}

// phase 1: tile pass + local perimeter graph.  With `rank_level` the entry successor lists are built too
// (needed for dt_launch_fa_summary and the inflow of dt_launch_fa_finish).
int dt_launch_fa_local(hipStream_t s, const DtWin &w, const uint8_t *fdr, void *scratch, size_t scratch_bytes,
                       int32_t *acc32, int rank_level) {
  if (w.H == 0 || w.W == 0) return DT_OK;
  DT_REQUIRE(scratch_bytes >= dt_flowacc_tiled_scratch(w.H, w.W), "scratch too small");
  FaScratch f = fa_layout(w, scratch);
  DT_REQUIRE(f.nnodes < 0x7FFFFFF0ll, "raster too large for one device tile");
  dim3 gt((unsigned)f.ntiles), b(256), gn((unsigned)((f.nnodes + 255) / 256));
  (void)acc32;  // written by pass 3 only
  hipLaunchKernelGGL(k_fa_tile1, gt, b, 0, s, fdr, w, f.tiles_x, f.rec, f.loc16, f.state, f.ext, f.entry_of,
                     f.parent);
  hipLaunchKernelGGL(k_fa_reduce, gn, b, 0, s, f.rec, f.nnodes, f.entry_of, f.parent, f.state, f.ext);
  if (rank_level) {
    hipLaunchKernelGGL(k_fa_nxt_init, gn, b, 0, s, f.rec, f.nnodes, f.entry_of, f.state, f.nxt);
  }
  return DT_OK;
}

int dt_launch_fa_summary(hipStream_t s, const DtWin &w, void *scratch, int64_t *A, int32_t *xr,
                         uint8_t *code) {
  int64_t P = dt_perim_count(w.H, w.W);
  if (P == 0) return DT_OK;
  FaScratch f = fa_layout(w, scratch);
  hipLaunchKernelGGL(k_fa_rank_summary, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, s, w, f.tiles_x,
                     f.rec, f.state, f.nxt, f.nnodes, P, A, xr, code);
  return DT_OK;
}

// phase 2: optional inflow from other ranks, carried along the entry successor lists of phase 1
// (dt_launch_fa_local with rank_level), then the final tile pass.
template <typename AccT>
static void fa_launch_tile3(hipStream_t s, dim3 gt, const DtWin &w, const uint8_t *fdr, const float *dem,
                            const FaScratch &f, AccT *acc, AccT thr, int8_t *river, int *status) {
  dim3 b(256);
  if (dem && river) hipLaunchKernelGGL((k_fa_tile3<true, true, AccT>), gt, b, 0, s, fdr, dem, w, f.tiles_x, f.ext, f.loc16, acc, thr, river, status);
  else if (dem) hipLaunchKernelGGL((k_fa_tile3<true, false, AccT>), gt, b, 0, s, fdr, dem, w, f.tiles_x, f.ext, f.loc16, acc, thr, river, status);
  else if (river) hipLaunchKernelGGL((k_fa_tile3<false, true, AccT>), gt, b, 0, s, fdr, dem, w, f.tiles_x, f.ext, f.loc16, acc, thr, river, status);
  else hipLaunchKernelGGL((k_fa_tile3<false, false, AccT>), gt, b, 0, s, fdr, dem, w, f.tiles_x, f.ext, f.loc16, acc, thr, river, status);
}

// `acc` is int32_t* (acc64 == 0) or int64_t* (acc64 != 0: rasters of >= 2^31 cells split over ranks)
int dt_launch_fa_finish(hipStream_t s, const DtWin &w, const uint8_t *fdr, const float *dem, void *scratch,
                        const unsigned long long *ext_perim, int64_t river_thr, void *acc, int acc64,
                        int8_t *river, int *status) {
  if (w.H == 0 || w.W == 0) return DT_OK;
  FaScratch f = fa_layout(w, scratch);
  dim3 gt((unsigned)f.ntiles), b(256), gn((unsigned)((f.nnodes + 255) / 256));
  if (ext_perim) {
    int64_t P = dt_perim_count(w.H, w.W);
    hipLaunchKernelGGL(k_fa_propagate, dim3((unsigned)((P + 255) / 256)), b, 0, s, w, f.tiles_x, f.nnodes, f.nxt,
                       ext_perim, P, f.ext);
  }
  hipLaunchKernelGGL(k_fa_poison, gn, b, 0, s, f.rec, f.nnodes, f.entry_of, f.state, f.ext);
  if (acc64) {
    fa_launch_tile3<long long>(s, gt, w, fdr, dem, f, (long long *)acc, (long long)river_thr, river, status);
  } else {
    int32_t thr = river_thr > 2147483647ll ? 2147483647 : (river_thr < -2147483647ll ? -2147483647 : (int32_t)river_thr);
    fa_launch_tile3<int32_t>(s, gt, w, fdr, dem, f, (int32_t *)acc, thr, river, status);
  }
  return DT_OK;
}

// ===========================================================================================
// Flow distance / drained-to river index / HAND (F3, F4; flowhand.py:566-846, :414-442)
// ===========================================================================================
// Word format (LDS per cell, and global per perimeter node; identical to the v1 kernels'):
//   ptr:32 | n_diag:16 | done:1 n_card:15
// "the path from here to `ptr` takes n_card cardinal and n_diag diagonal moves".
//   in a tile : ptr = local cell (bits 0-11) | kind << 12, kind of a finished path's end:
//               1 river cell, 2 dead (-100), 3 exit (steps into another tile), 4 rank exit
//   node      : ptr = node id while unresolved; when done: the core-local flat index y*W+x of the
//               river cell, or (bit 31) a key into the rank-exit payload table, or FHT_DEAD.
//               Nodes [nnodes, nnodes + P) are GHOSTS, one per core-ring cell: a rank exit points at
//               its ghost, which is a fixed point until the rank-level result is written into it.
#define FHT_DEAD 0xFFFFFFFFu
#define FHT_REMOTE 0x80000000u
#define FHT_DONE 0x8000u
#define FHT_CAP 20000u
#define K_RIVER 1u
#define K_DEAD 2u
#define K_EXIT 3u
#define K_REXIT 4u

__device__ __forceinline__ unsigned long long fht_pack(uint32_t ptr, uint32_t nd, uint32_t ncf) {
  return ((unsigned long long)ptr << 32) | ((unsigned long long)nd << 16) | (unsigned long long)ncf;
}

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

// stage fdr (with a one-cell halo ring, read from the neighbouring rank's halo where the tile touches
// the core border) and the river mask; build and resolve the in-tile words.
__device__ __forceinline__ void fht_solve_tile(const FhTile &T, const uint8_t *__restrict__ fdr,
                                               const int8_t *__restrict__ river, const DtWin &w, int y0,
                                               int x0) {
  dt_tile_load_fdr(fdr, w, y0, x0, T.s_fdr);
  for (int i = threadIdx.x; i < 2 * (TW + 2) + 2 * TH; i += 256) {
    int y, x;
    if (i < TW + 2) { y = y0 - 1; x = x0 - 1 + i; }
    else if (i < 2 * (TW + 2)) { y = y0 + TH; x = x0 - 1 + (i - (TW + 2)); }
    else if (i < 2 * (TW + 2) + TH) { y = y0 + (i - 2 * (TW + 2)); x = x0 - 1; }
    else { y = y0 + (i - 2 * (TW + 2) - TH); x = x0 + TW; }
    uint8_t v = 0;
    if (dt_readable(w, y, x)) v = fdr[(long long)y * w.ld + x];
    T.s_halo[i] = v;
  }
  // river mask: staged through the word array with 16-byte row loads (it is free until the words are
  // built), then picked up as one bit per owned cell
  uint8_t *s_riv = reinterpret_cast<uint8_t *>(T.s_st);
  dt_tile_load_fdr(reinterpret_cast<const uint8_t *>(river), w, y0, x0, s_riv);
  __syncthreads();
  uint32_t riv = 0;
  for (int j = 0; j < CPT; j++) {
    int c = threadIdx.x + 256 * j;
    if (s_riv[c] == 1) riv |= 1u << j;
  }
  __syncthreads();
  if (dt_tile_interior(w, y0, x0)) {
    // every cell and every successor is inside the core: no window tests
#pragma unroll
    for (int j = 0; j < CPT; j++) {
      int c = threadIdx.x + 256 * j;
      int ly = c / TW, lx = c % TW;
      uint32_t code = T.s_fdr[c];
      uint32_t ptr = (uint32_t)c | (K_DEAD << 12), lo = FHT_DONE;  // flowhand.py:601 / non-D8 code :830
      if (code != 0u && ((riv >> j) & 1u)) {
        ptr = (uint32_t)c | (K_RIVER << 12);  // flowhand.py:609-612
      } else if (dt_d8_valid(code)) {
        int dy, dx;
        dt_d8_delta(code, dy, dx);
        uint32_t ny = (uint32_t)(ly + dy), nx = (uint32_t)(lx + dx);
        bool in_tile = ny < (uint32_t)TH && nx < (uint32_t)TW;
        uint32_t tcode = in_tile ? (uint32_t)T.s_fdr[ny * TW + nx] : fht_fdr_at(T, ly + dy, lx + dx);
        if (tcode != 0u) {  // else: arrival on fdr == 0 (flowhand.py:826)
          if (in_tile) {
            bool diag = dy != 0 && dx != 0;
            ptr = ny * TW + nx;
            lo = diag ? (1u << 16) : 1u;
          } else {
            ptr = (uint32_t)c | (K_EXIT << 12);  // the step itself is added by the user
          }
        }
      }
      T.s_st[c] = ((unsigned long long)ptr << 32) | (unsigned long long)lo;
    }
  } else {
    for (int j = 0; j < CPT; j++) {
      int c = threadIdx.x + 256 * j;
      int ly = c / TW, lx = c % TW;
      int y = y0 + ly, x = x0 + lx;
      uint32_t code = T.s_fdr[c];
      unsigned long long s;
      if (y >= w.H || x >= w.W || code == 0u) {
        s = fht_pack((uint32_t)c | (K_DEAD << 12), 0, FHT_DONE);  // flowhand.py:601
      } else if ((riv >> j) & 1u) {
        s = fht_pack((uint32_t)c | (K_RIVER << 12), 0, FHT_DONE);  // flowhand.py:609-612
      } else if (!dt_d8_valid(code)) {
        s = fht_pack((uint32_t)c | (K_DEAD << 12), 0, FHT_DONE);  // non-D8 code: revisit test :830
      } else {
        int dy, dx;
        dt_d8_delta(code, dy, dx);
        int ty = y + dy, tx = x + dx;
        if (!dt_in_global(w, ty, tx) || fht_fdr_at(T, ly + dy, lx + dx) == 0u) {
          s = fht_pack((uint32_t)c | (K_DEAD << 12), 0, FHT_DONE);  // raster exit / arrival on fdr==0
        } else if (!dt_in_core(w, ty, tx)) {
          s = fht_pack((uint32_t)c | (K_REXIT << 12), 0, FHT_DONE);
        } else if (ly + dy < 0 || ly + dy >= TH || lx + dx < 0 || lx + dx >= TW) {
          s = fht_pack((uint32_t)c | (K_EXIT << 12), 0, FHT_DONE);  // the step itself is added by the user
        } else {
          bool diag = dy != 0 && dx != 0;
          s = fht_pack((uint32_t)((ly + dy) * TW + lx + dx), diag ? 1u : 0u, diag ? 0u : 1u);
        }
      }
      T.s_st[c] = s;
    }
  }
  __syncthreads();
  // Pointer doubling in place.  The low 32 bits hold n_diag:16 | done:1 | n_card:15, so ONE integer add
  // of the two low words adds both move counts and inherits the target's done bit (in-tile counts stay
  // below 2^13: no carry into the done bit or between the fields).  An acyclic in-tile path has < 4096
  // moves and is finished after 12 rounds; whatever is still unfinished after 13 runs into an in-tile D8
  // cycle and is dead (the reference's revisit test / move cap, flowhand.py:830-837).
  ulonglong2 *s_st2 = reinterpret_cast<ulonglong2 *>(T.s_st);  // lane owns adjacent cell pairs
  for (int round = 0; round < 13; round++) {
    int changed = 0;
#pragma unroll
    for (int j = 0; j < CPT / 2; j++) {
      int c2 = threadIdx.x + 256 * j;
      ulonglong2 s = s_st2[c2];
      bool dx = ((uint32_t)s.x & FHT_DONE) != 0u, dy = ((uint32_t)s.y & FHT_DONE) != 0u;
      if (dx && dy) continue;
      if (!dx) {
        unsigned long long t = T.s_st[(uint32_t)(s.x >> 32) & 0xFFFu];
        s.x = (t & 0xFFFFFFFF00000000ull) | (unsigned long long)((uint32_t)s.x + (uint32_t)t);
      }
      if (!dy) {
        unsigned long long t = T.s_st[(uint32_t)(s.y >> 32) & 0xFFFu];
        s.y = (t & 0xFFFFFFFF00000000ull) | (unsigned long long)((uint32_t)s.y + (uint32_t)t);
      }
      s_st2[c2] = s;
      // (a round's work is over when every cell is done AFTER it: "somebody was not done before it" cost one more
      // round of reads and a barrier that found nothing to do)
      changed |= (((uint32_t)s.x & (uint32_t)s.y & FHT_DONE) == 0u) ? 1 : 0;
    }
    if (!__syncthreads_or(changed)) break;
  }
#pragma unroll
  for (int j = 0; j < CPT; j++) {
    int c = threadIdx.x + 256 * j;
    if (!((uint32_t)T.s_st[c] & FHT_DONE)) T.s_st[c] = fht_pack((uint32_t)c | (K_DEAD << 12), 0, FHT_DONE);
  }
  __syncthreads();
}

// The in-tile word cache: 4 bytes per cell (f:12 | kind:3 | n_diag:8 | n_card:8, every cached word is
// "done") when no path of the tile has more than 255 moves of a kind, the full 8-byte words otherwise; one
// flag byte per tile.  The narrow words sit in the first half of the tile's 8-byte slot.
__device__ __forceinline__ uint32_t fh_cache_pack(unsigned long long s) {
  uint32_t ptr = (uint32_t)(s >> 32);
  return (ptr & 0x7FFFu) | ((uint32_t)((s >> 16) & 0xFFu) << 15) | ((uint32_t)(s & 0xFFu) << 23);
}
__device__ __forceinline__ unsigned long long fh_cache_unpack(uint32_t v) {
  return fht_pack(v & 0x7FFFu, (v >> 15) & 0xFFu, ((v >> 23) & 0xFFu) | FHT_DONE);
}
__device__ __forceinline__ unsigned long long fh_cache_get(const unsigned long long *__restrict__ cache, bool wide,
                                                            int tile, int c) {
  if (wide) return cache[(size_t)tile * NT + c];
  return fh_cache_unpack(reinterpret_cast<const uint32_t *>(cache + (size_t)tile * NT)[c]);
}

// pass 1: perimeter node words
// `cache` keeps every cell's resolved in-tile word (tile-major: slot cache[tile * NT ...]) so that
// pass 3 does not have to solve the tile again: 16 bytes / cell of streaming HBM traffic instead of a
// second LDS-bound pointer doubling.
__global__ __launch_bounds__(256) void k_fh_tile1(const uint8_t *__restrict__ fdr,
                                                 const int8_t *__restrict__ river, DtWin w, int tiles_x,
                                                 uint32_t nnodes, unsigned long long *__restrict__ nodes,
                                                 unsigned long long *__restrict__ cache,
                                                 uint8_t *__restrict__ cache_wide, int only_marked, int ntiles) {
  __shared__ __attribute__((aligned(16))) uint8_t s_fdr[NT];
  __shared__ uint8_t s_halo[2 * (TW + 2) + 2 * TH];
  __shared__ __attribute__((aligned(16))) unsigned long long s_st[NT];
  // grid-stride over the tiles: launched over all of them as the 64-bit solve, over a small grid as the redo of
  // the tiles the narrow kernel gave up (usually none: 2048 workgroups look at 32 flags each and leave)
  for (int tile = (int)blockIdx.x; tile < ntiles; tile += (int)gridDim.x) {
  if (only_marked && cache_wide[tile] != 2) continue;  // block-uniform: k_fh_tile1n did this tile
  const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
  const int y0 = ty * TH, x0 = tx * TW;
  FhTile T{s_fdr, s_halo, s_st};
  fht_solve_tile(T, fdr, river, w, y0, x0);
  if (threadIdx.x < PS) {
    int ly, lx;
    dt_cell_of_slot(threadIdx.x, ly, lx);
    unsigned long long s = s_st[ly * TW + lx];
    uint32_t ptr = (uint32_t)(s >> 32), nd = (uint32_t)((s >> 16) & 0xFFFFu);
    uint32_t ncf = (uint32_t)(s & 0xFFFFu), nc = ncf & 0x7FFFu;
    uint32_t kind = (ptr >> 12) & 7u, f = ptr & 0xFFFu;
    int fy = (int)f / TW, fx = (int)f % TW;
    unsigned long long o = fht_pack(FHT_DEAD, 0, FHT_DONE);
    if (!(ncf & FHT_DONE)) {
      // unreachable: every in-tile path is finished after the doubling rounds
    } else if (kind == K_RIVER) {
      o = fht_pack((uint32_t)((y0 + fy) * w.W + x0 + fx), nd, nc | FHT_DONE);
    } else if (kind == K_EXIT || kind == K_REXIT) {
      int dy, dx;
      dt_d8_delta(s_fdr[f], dy, dx);
      bool diag = dy != 0 && dx != 0;
      uint32_t node = kind == K_EXIT ? dt_node_of(y0 + fy + dy, x0 + fx + dx, tiles_x)
                                     : nnodes + (uint32_t)dt_perim_index(w.H, w.W, y0 + fy, x0 + fx);
      o = fht_pack(node, nd + (diag ? 1u : 0u), nc + (diag ? 0u : 1u));
    }
    nodes[(size_t)tile * PS + threadIdx.x] = o;
  }
  unsigned long long wv[CPT];
  int wide = 0;
#pragma unroll
  for (int j = 0; j < CPT; j++) {
    wv[j] = s_st[threadIdx.x + 256 * j];
    wide |= (((uint32_t)wv[j] & 0x7FFFu) > 255u || ((uint32_t)(wv[j] >> 16) & 0xFFFFu) > 255u) ? 1 : 0;
  }
  wide = __syncthreads_or(wide);
  if (threadIdx.x == 0) cache_wide[tile] = (uint8_t)wide;
  if (wide) {
#pragma unroll
    for (int j = 0; j < CPT; j++) cache[(size_t)tile * NT + threadIdx.x + 256 * j] = wv[j];
  } else {
    uint32_t *c32 = reinterpret_cast<uint32_t *>(cache + (size_t)tile * NT);
#pragma unroll
    for (int j = 0; j < CPT; j++) c32[threadIdx.x + 256 * j] = fh_cache_pack(wv[j]);
  }
  __syncthreads();  // the LDS images are reused by the next tile
  }
}

// ---- pass 1, narrow form -----------------------------------------------------------------------------
// The same tile solve with one 32-bit LDS word per cell,  n_card:9 | n_diag:9 | done:1 | ptr:12  (bit 8 of a
// count is a guard: counts are <= 255 before an addition, so a sum cannot carry into the next field), and the
// kind of a path's end in a byte array indexed by the end cell.  21 KiB of LDS less than the 64-bit form:
// six tiles per CU instead of four, and half the LDS bytes per doubling step.  A tile in which some path
// makes more than 255 moves of a kind cannot be represented: its workgroup stops, marks the tile
// (cache_wide = 2) and the 64-bit kernel, launched afterwards over all tiles, redoes exactly the marked ones.
#define FN_CNT 0x3FFFFu  /* both counts */
#define FN_DONE 0x40000u
#define FN_LOW 0x7FFFFu  /* counts + done: what ONE addition propagates */
#define FN_OVF 0x20100u  /* a count above 255 */
#define FN_PTR_SH 19
// The narrow pass 1 from the point where the tile's codes (s_fdr) and the code table (s_lut) are in LDS, every lane
// holds the river mask of its 4 x 4 cells (riv4), *s_ovf is 0 and a barrier has been passed: shared by k_fh_tile1n (which stages both from HBM) and by
// k_fa3fh1 (flow accumulation's last pass, which has the codes in registers and has just computed the mask).
// (No halo ring of codes: "the successor's code is 0" need not be tested at the predecessor -- a cell whose code is 0
// is a dead end in its own right (flowhand.py:601, :826), inside the tile through s_kind of the path's end, across a
// tile or rank border through that cell's perimeter node / ring summary.)
// s_lut (256 words, fh_lut_entry): what a D8 code adds to a cell's word  c << FN_PTR_SH  -- the successor's offset in
// the pointer field and the move in its count field (0 in the count field: not a D8 code).
__device__ __forceinline__ uint32_t fh_lut_entry(uint32_t code) {
  if (!dt_d8_valid(code)) return 0u;
  int dy, dx;
  dt_d8_delta(code, dy, dx);
  return ((uint32_t)(dy * TW + dx) << FN_PTR_SH) + ((dy != 0 && dx != 0) ? (1u << 9) : 1u);
}
__device__ __forceinline__ void fh_tile1n_body(uint8_t *s_fdr, uint32_t *s_w, uint8_t *s_kind, int *s_ovf_p,
                                               const uint32_t *s_lut, const uint32_t (&riv4)[NT / 4 / 256],
                                               const DtWin &w, int tile, int tiles_x, int y0, int x0, uint32_t nnodes,
                                               unsigned long long *__restrict__ nodes,
                                               unsigned long long *__restrict__ cache,
                                               uint8_t *__restrict__ cache_wide) {
#define s_ovf (*s_ovf_p)
  // A lane sets up the 4 x 4 cells  4 (t + 256 u) + k  whose river mask it holds (riv4, one byte per cell): codes,
  // words and kinds move as 32 / 128-bit LDS accesses, and nobody else touches these cells before the barrier.
  const bool interior = dt_tile_interior(w, y0, x0);
  const uint32_t lx0 = (4u * threadIdx.x) & 63u;  // column of my cells' first (1024 u keeps it)
#pragma unroll
  for (int u = 0; u < NT / 4 / 256; u++) {
    const int c0 = 4 * (threadIdx.x + 256 * u);
    const uint32_t codes = *reinterpret_cast<const uint32_t *>(&s_fdr[c0]);
    uint32_t wd[4], kinds = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int c = c0 + k;
      const uint32_t code = (codes >> (8 * k)) & 0xFFu, cw = (uint32_t)c << FN_PTR_SH;
      const bool is_river = ((riv4[u] >> (8 * k)) & 0xFFu) == 1u;
      uint32_t word = cw | FN_DONE, kind = K_DEAD;  // flowhand.py:601 / non-D8 code :830
      if (interior) {
        // every cell and every successor is in the core: the table gives the successor's word in one addition; it
        // left the tile iff the 13-bit pointer field went negative / past 4095 (bit 31) or its column wrapped
        // (0 <-> 63: the new column differs from mine by 63 instead of <= 1)
        const uint32_t wn = cw + s_lut[code];
        if (code != 0u) {
          if (is_river) {
            kind = K_RIVER;  // flowhand.py:609-612
          } else if (wn & FN_CNT) {
            const bool out = (int32_t)wn < 0 || (((wn >> FN_PTR_SH) & 63u) - (lx0 + (uint32_t)k - 1u)) > 2u;
            if (out) kind = K_EXIT;  // the step itself is added by the user
            else word = wn;
          }
        }
      } else {
        const int ly = c / TW, lx = c % TW;
        const int y = y0 + ly, x = x0 + lx;
        if (y < w.H && x < w.W && code != 0u) {
          if (is_river) {
            kind = K_RIVER;
          } else if (dt_d8_valid(code)) {
            int dy, dx;
            dt_d8_delta(code, dy, dx);
            uint32_t ny = (uint32_t)(ly + dy), nx = (uint32_t)(lx + dx);
            bool in_tile = ny < (uint32_t)TH && nx < (uint32_t)TW;
            // a step off the raster stays dead (flowhand.py:623-628)
            if (dt_in_global(w, y + dy, x + dx)) {
              if (!dt_in_core(w, y + dy, x + dx)) kind = K_REXIT;
              else if (!in_tile) kind = K_EXIT;
              else word = ((ny * TW + nx) << FN_PTR_SH) | ((dy != 0 && dx != 0) ? (1u << 9) : 1u);
            }
          }
        }
      }
      wd[k] = word;
      kinds |= kind << (8 * k);
    }
    *reinterpret_cast<uint4 *>(&s_w[c0]) = make_uint4(wd[0], wd[1], wd[2], wd[3]);
    *reinterpret_cast<uint32_t *>(&s_kind[c0]) = kinds;
  }
  __syncthreads();
  // pointer doubling in place; ONE addition of the low 19 bits adds both counts and inherits the done bit
  uint2 *s_w2 = reinterpret_cast<uint2 *>(s_w);  // lane owns adjacent cell pairs
  // a lane's own words change only by its own hand: they live in registers, LDS gets the copies others gather
  uint2 own[CPT / 2];
#pragma unroll
  for (int j = 0; j < CPT / 2; j++) own[j] = s_w2[threadIdx.x + 256 * j];
  for (int round = 0; round < 13; round++) {
    int changed = 0;
    uint32_t ovf = 0;
#pragma unroll
    for (int j = 0; j < CPT / 2; j++) {
      int c2 = threadIdx.x + 256 * j;
      uint2 v = own[j];
      bool dx = (v.x & FN_DONE) != 0u, dy = (v.y & FN_DONE) != 0u;
      if (dx && dy) continue;
      // two jumps per round (in place: a target read here may or may not have jumped already this round; every
      // value it ever held is true): half the rounds, barriers and done-tests for the same gathers
      // a jump: pointer and done bit of the target, counts added -- ONE addition, target word + my counts (my done
      // bit is clear, and counts <= 255 cannot carry out of their 9-bit fields)
      if (!dx) {
        v.x = s_w[v.x >> FN_PTR_SH] + (v.x & FN_CNT);
        if (!(v.x & (FN_DONE | FN_OVF))) v.x = s_w[v.x >> FN_PTR_SH] + (v.x & FN_CNT);
        ovf |= v.x & FN_OVF;
      }
      if (!dy) {
        v.y = s_w[v.y >> FN_PTR_SH] + (v.y & FN_CNT);
        if (!(v.y & (FN_DONE | FN_OVF))) v.y = s_w[v.y >> FN_PTR_SH] + (v.y & FN_CNT);
        ovf |= v.y & FN_OVF;
      }
      own[j] = v;
      s_w2[c2] = v;
      changed |= ((v.x & v.y & FN_DONE) == 0u) ? 1 : 0;  // (done after this round: see dt_tile_sums_packed)
    }
    if (ovf) s_ovf = 1;
    if (!__syncthreads_or(changed)) break;
    if (s_ovf) break;  // block-uniform: written before the barrier
  }
  if (s_ovf) {
    if (threadIdx.x == 0) cache_wide[tile] = 2;  // for k_fh_tile1
    return;
  }
  // cells unfinished after 13 rounds run into an in-tile D8 cycle: dead (flowhand.py:830-837).  The final words of a
  // lane's pairs are in its registers and the last round ended with a barrier: the cache's narrow words go straight
  // into s_w (cache store, perimeter lanes), the kinds of the paths' ends are the only gathers left.
  auto narrow = [&](uint32_t v, uint32_t c) -> uint32_t {
    uint32_t f = v >> FN_PTR_SH, kind = s_kind[f];
    if (!(v & FN_DONE)) { f = c; kind = K_DEAD; v = 0; }
    return f | (kind << 12) | (((v >> 9) & 0xFFu) << 15) | ((v & 0xFFu) << 23);
  };
#pragma unroll
  for (int j = 0; j < CPT / 2; j++) {
    const int c2 = threadIdx.x + 256 * j;
    s_w2[c2] = make_uint2(narrow(own[j].x, 2u * (uint32_t)c2), narrow(own[j].y, 2u * (uint32_t)c2 + 1u));
  }
  if (threadIdx.x == 0) cache_wide[tile] = 0;
  __syncthreads();
  uint32_t *c32 = reinterpret_cast<uint32_t *>(cache + (size_t)tile * NT);
#pragma unroll
  for (int u = 0; u < NT / 4 / 256; u++) {
    int c = 4 * (threadIdx.x + 256 * u);
    *reinterpret_cast<uint4 *>(c32 + c) = *reinterpret_cast<const uint4 *>(&s_w[c]);
  }
  if (threadIdx.x < PS) {
    int ly, lx;
    dt_cell_of_slot(threadIdx.x, ly, lx);
    unsigned long long sw = fh_cache_unpack(s_w[ly * TW + lx]);
    uint32_t ptr = (uint32_t)(sw >> 32), nd = (uint32_t)((sw >> 16) & 0xFFFFu), nc = (uint32_t)sw & 0x7FFFu;
    uint32_t kind = (ptr >> 12) & 7u, f = ptr & 0xFFFu;
    int fy = (int)f / TW, fx = (int)f % TW;
    unsigned long long o = fht_pack(FHT_DEAD, 0, FHT_DONE);
    if (kind == K_RIVER) {
      o = fht_pack((uint32_t)((y0 + fy) * w.W + x0 + fx), nd, nc | FHT_DONE);
    } else if (kind == K_EXIT || kind == K_REXIT) {
      int dy, dx;
      dt_d8_delta(s_fdr[f], dy, dx);
      bool diag = dy != 0 && dx != 0;
      uint32_t node = kind == K_EXIT ? dt_node_of(y0 + fy + dy, x0 + fx + dx, tiles_x)
                                     : nnodes + (uint32_t)dt_perim_index(w.H, w.W, y0 + fy, x0 + fx);
      o = fht_pack(node, nd + (diag ? 1u : 0u), nc + (diag ? 0u : 1u));
    }
    nodes[(size_t)tile * PS + threadIdx.x] = o;
  }
#undef s_ovf
}

__global__ __launch_bounds__(256, 6) void k_fh_tile1n(const uint8_t *__restrict__ fdr,
                                                     const int8_t *__restrict__ river, DtWin w, int tiles_x,
                                                     uint32_t nnodes, unsigned long long *__restrict__ nodes,
                                                     unsigned long long *__restrict__ cache,
                                                     uint8_t *__restrict__ cache_wide) {
  __shared__ __attribute__((aligned(16))) uint8_t s_fdr[NT];
  __shared__ __attribute__((aligned(16))) uint32_t s_w[NT];
  __shared__ __attribute__((aligned(16))) uint8_t s_kind[NT];  // first the river mask, then the end kinds
  __shared__ uint32_t s_lut[256];
  __shared__ int s_ovf;
  const int tile = dt_tile_of_block((int)blockIdx.x, (int)gridDim.x);
  const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
  const int y0 = ty * TH, x0 = tx * TW;
  // the tile's codes and its river mask: both loads issued before the first is waited for
  const uint4 v_fdr = dt_tile_fetch16(fdr, w, y0, x0);
  const uint4 v_riv = dt_tile_fetch16(reinterpret_cast<const uint8_t *>(river), w, y0, x0);
  dt_tile_put16(s_fdr, v_fdr);
  dt_tile_put16(s_kind, v_riv);
  s_lut[threadIdx.x] = fh_lut_entry(threadIdx.x);
  if (threadIdx.x == 0) s_ovf = 0;
  __syncthreads();
  uint32_t riv4[NT / 4 / 256];
#pragma unroll
  for (int u = 0; u < NT / 4 / 256; u++) riv4[u] = *reinterpret_cast<const uint32_t *>(&s_kind[4 * (threadIdx.x + 256 * u)]);
  fh_tile1n_body(s_fdr, s_w, s_kind, &s_ovf, s_lut, riv4, w, tile, tiles_x, y0, x0, nnodes, nodes, cache, cache_wide);
}

// ---- flow accumulation's last pass and HAND's first pass in one kernel ----------------------------------------
// Both stage the same 64 x 64 tile of direction codes, and HAND's river mask is what this pass has just computed
// (acc > threshold): one read of the codes instead of two, no read of the mask, one launch and one tile staging less.
// The common form only: int32 accumulation, tiles of whole 64-cell rows on 16-byte aligned rasters (the launcher
// falls back to the two separate kernels otherwise).  The accumulation half is k_fa_tile3's vector path, the HAND
// half fh_tile1n_body; they run one after the other in the same 25.5 KiB of LDS.
// ND: where "this cell is nodata" (accumulation -100) comes from -- 0 nowhere, 1 the DEM (4 B/cell read for one bit),
// 2 the D8 kernel's mask (a 16-bit word per 4 x 4 patch, ldm words per row of patches: round 4, the chain's form)
template <int ND>
__global__ __launch_bounds__(256, 6) void k_fa3fh1(const uint8_t *__restrict__ fdr, const float *__restrict__ dem,
                                                  const uint8_t *__restrict__ nod4, int ldm, DtWin w, int tiles_x,
                                                  const unsigned long long *__restrict__ ext,
                                                  const uint16_t *__restrict__ loc16, int32_t *__restrict__ acc32,
                                                  int32_t river_thr, int8_t *__restrict__ river,
                                                  int *__restrict__ status, uint32_t nnodes,
                                                  unsigned long long *__restrict__ nodes,
                                                  unsigned long long *__restrict__ cache,
                                                  uint8_t *__restrict__ cache_wide) {
#define P3(c) ((uint32_t)(c) + (((uint32_t)(c) >> 6) << 2))
#define NT3 (TH * (TW + 4))
  __shared__ __attribute__((aligned(16))) unsigned char smem[NT3 * 6];  // 26112 bytes
  __shared__ unsigned long long s_in;
  __shared__ int s_ovf;
  // accumulation half: delta raster + padded successor indices
  uint32_t *s_delta = reinterpret_cast<uint32_t *>(smem);
  uint16_t *s_nxt = reinterpret_cast<uint16_t *>(smem + NT3 * 4);
  const int tile = dt_tile_of_block((int)blockIdx.x, (int)gridDim.x);
  const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
  const int y0 = ty * TH, x0 = tx * TW;
  const uint4 v_fdr = dt_tile_fetch16(fdr, w, y0, x0);
  unsigned long long e = 0ull;
  if (threadIdx.x < PS) e = ext[(size_t)tile * PS + threadIdx.x];
  constexpr int VPT = NT / 4 / 256;
  uint2 l4[VPT];
  float4 z4[VPT];
#pragma unroll
  for (int u = 0; u < VPT; u++) {
    int c = 4 * (threadIdx.x + 256 * u);
    int y = y0 + c / TW;
    l4[u] = *reinterpret_cast<const uint2 *>(loc16 + (size_t)tile * NT + c);
    z4[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ND == 1 && y < w.H) z4[u] = *reinterpret_cast<const float4 *>(dem + (long long)y * w.ld + x0 + c % TW);
    if (ND == 2 && y < w.H) {  // the four cells' bits, turned into the sentinel where set: finish() tests z <= -100
      const uint32_t m = (uint32_t)reinterpret_cast<const uint16_t *>(nod4)[(long long)(y >> 2) * ldm + ((x0 + c % TW) >> 2)] >>
                         (4 * (y & 3));
      z4[u] = make_float4((m & 1u) ? DT_NODATA : 0.f, (m & 2u) ? DT_NODATA : 0.f, (m & 4u) ? DT_NODATA : 0.f,
                          (m & 8u) ? DT_NODATA : 0.f);
    }
  }
  if (threadIdx.x == 0) s_ovf = 0;
  if (__syncthreads_or(!(e & FA_CYCLE) && FA_VALUE(e) >= (1ull << 22))) {  // see k_fa_tile3
    if (threadIdx.x == 0) s_in = 0ull;
    __syncthreads();
    if (e != 0ull && !(e & FA_CYCLE)) atomicAdd(&s_in, FA_VALUE(e));
    __syncthreads();
    if (threadIdx.x == 0 && s_in >= (1ull << 31) - (unsigned long long)NT && status) atomicOr(status, DT_STATUS_ACC_OVERFLOW);
  }
  // successor indices of the 16 cells whose codes this lane fetched (row t / 4, columns 16 (t % 4) ..): straight from
  // its registers, the padded row is contiguous -- four 8-byte stores of indices, four 16-byte stores of zeros
  {
    const int ly = (int)threadIdx.x >> 2, lxb = ((int)threadIdx.x & 3) * 16;
    const uint32_t cv[4] = {v_fdr.x, v_fdr.y, v_fdr.z, v_fdr.w};
    const bool interior = dt_tile_interior(w, y0, x0);
    const int pbase = ly * (TW + 4) + lxb;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      uint32_t nn[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const uint32_t code = (cv[q] >> (8 * k)) & 0xFFu;
        const int lx = lxb + 4 * q + k;
        const uint32_t n = interior ? dt_tile_next_interior(code, ly, lx) : dt_tile_next(code, ly, lx, y0, x0, w);
        nn[k] = n < NT ? P3(n) : n;
      }
      *reinterpret_cast<uint2 *>(&s_nxt[pbase + 4 * q]) = make_uint2(nn[0] | (nn[1] << 16), nn[2] | (nn[3] << 16));
      *reinterpret_cast<uint4 *>(&s_delta[pbase + 4 * q]) = make_uint4(0, 0, 0, 0);
    }
  }
  __syncthreads();
  if (e != 0ull) {
    int ly, lx;
    dt_cell_of_slot(threadIdx.x, ly, lx);
    uint32_t c = P3(ly * TW + lx);
    if (e & FA_CYCLE) {
      for (int it = 0; it < NT && c < NT3; it++) {
        atomicOr(&s_delta[c], 0x80000000u);
        c = s_nxt[c];
      }
    } else {
      const uint32_t add = (uint32_t)e;
      for (int it = 0; it < NT && c < NT3; it++) {
        atomicAdd(&s_delta[c], add);
        c = s_nxt[c];
      }
    }
  }
  __syncthreads();
  auto finish = [&](uint32_t l16, uint32_t d, float z) -> int32_t {
    int32_t v = l16 == 0xFFFFu ? -100 : (int32_t)l16;
    if (v != -100) v += (int32_t)(d & 0x7FFFFFFFu);
    if (d & 0x80000000u) v = -100;
    if (ND != 0 && z <= DT_NODATA) v = -100;
    return v;
  };
  uint32_t riv4[VPT];  // the river mask of the lane's 4 x 4 cells, one byte per cell
#pragma unroll
  for (int u = 0; u < VPT; u++) {
    int c = 4 * (threadIdx.x + 256 * u);
    int y = y0 + c / TW;
    riv4[u] = 0u;
    if (y >= w.H) continue;
    long long o = (long long)y * w.ld + x0 + c % TW;
    uint4 d = *reinterpret_cast<const uint4 *>(&s_delta[P3(c)]);
    int4 v = make_int4(finish(l4[u].x & 0xFFFFu, d.x, z4[u].x), finish(l4[u].x >> 16, d.y, z4[u].y),
                       finish(l4[u].y & 0xFFFFu, d.z, z4[u].z), finish(l4[u].y >> 16, d.w, z4[u].w));
    *reinterpret_cast<int4 *>(acc32 + o) = v;
    riv4[u] = (v.x > river_thr ? 1u : 0u) | (v.y > river_thr ? 0x100u : 0u) | (v.z > river_thr ? 0x10000u : 0u) |
              (v.w > river_thr ? 0x1000000u : 0u);
    *reinterpret_cast<uint32_t *>(river + o) = riv4[u];
  }
  __syncthreads();  // everybody is done with the delta raster: the same LDS now holds HAND's arrays
#undef P3
#undef NT3
  uint32_t *s_w = reinterpret_cast<uint32_t *>(smem);             // 16 KiB
  uint8_t *s_fdr = smem + NT * 4;                                  // 4 KiB
  uint8_t *s_kind = smem + NT * 5;                                 // 4 KiB: first the river mask, then the end kinds
  uint32_t *s_lut = reinterpret_cast<uint32_t *>(smem + NT * 6);   // 1 KiB of the 1.5 KiB left
  s_lut[threadIdx.x] = fh_lut_entry(threadIdx.x);
  dt_tile_put16(s_fdr, v_fdr);
  __syncthreads();
  fh_tile1n_body(s_fdr, s_w, s_kind, &s_ovf, s_lut, riv4, w, tile, tiles_x, y0, x0, nnodes, nodes, cache, cache_wide);
}

// ghost g of ring cell i: a fixed point (ptr = itself, no moves, not done) until resolved
__global__ __launch_bounds__(256) void k_fh_ghost_init(unsigned long long *__restrict__ nodes, uint32_t nnodes,
                                                      int64_t P) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < P) nodes[nnodes + i] = fht_pack(nnodes + (uint32_t)i, 0, 0);
}

// pass 2: pointer doubling over the perimeter nodes (same word format as the v1 raster kernel)
__global__ __launch_bounds__(256) void k_fh_node_jump(unsigned long long *__restrict__ state, int64_t n,
                                                      int *__restrict__ flags, int round, int hops) {
  // flags[r] != 0: round r left a node that can still make progress.  Rounds after the first quiet one are
  // no-ops and return at once.  `hops` jumps per node and launch: any value a word ever held is a true statement,
  // so a later jump may use a target this launch has or has not updated yet -- either way the resolved distance
  // grows at least (hops + 1)-fold per launch (doubles with one hop), and one stream over the 16.6 M node words
  // serves all the jumps (the kernel is bound by that stream).
  if (flags && round > 0 && flags[round - 1] == 0) {
    if (blockIdx.x == 0 && threadIdx.x == 0) flags[round] = 0;
    return;
  }
  bool pending = false;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    unsigned long long s = __hip_atomic_load(&state[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    bool open = false, changed = false;
    for (int h = 0; h < hops; h++) {
      const uint32_t ncf = (uint32_t)(s & 0xFFFFu);
      const uint32_t ptr = (uint32_t)(s >> 32), nd = (uint32_t)((s >> 16) & 0xFFFFu);
      if ((ncf & FHT_DONE) || ptr == (uint32_t)i) break;  // finished, or an unresolved ghost
      const unsigned long long t = __hip_atomic_load(&state[ptr], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const uint32_t tptr = (uint32_t)(t >> 32), tnd = (uint32_t)((t >> 16) & 0xFFFFu);
      const uint32_t tncf = (uint32_t)(t & 0xFFFFu);
      const uint32_t nnc = ncf + (tncf & 0x7FFFu), nnd = nd + tnd;
      if (((tncf & FHT_DONE) && tptr == FHT_DEAD) || nnc + nnd > FHT_CAP) s = fht_pack(FHT_DEAD, 0, FHT_DONE);
      else s = fht_pack(tptr, nnd, nnc | (tncf & FHT_DONE));
      changed = true;
      // still open and not parked on an unresolved ghost (a fixed point pointing at itself)
      open = !(s & (unsigned long long)FHT_DONE) && tptr != ptr;
      if (!open) break;
    }
    if (changed) __hip_atomic_store(&state[i], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    pending = pending || open;
  }
  if (flags && __any(pending) && (threadIdx.x & 63) == 0) flags[round] = 1;
}

// rank level: one row per core-ring cell describing the path that ENTERS the rank there:
//   kind 1 river (ref = core-local flat index, zr / ar = its height / accumulation), 2 dead,
//   4 leaves the rank again through ring cell `ref` after (nc, nd) moves INCLUDING the crossing step
template <typename AccT>
__global__ __launch_bounds__(256) void k_fh_rank_summary(DtWin w, int tiles_x, uint32_t nnodes,
                                                        const unsigned long long *__restrict__ nodes,
                                                        const float *__restrict__ dem,
                                                        const AccT *__restrict__ acc32, int64_t P,
                                                        uint8_t *__restrict__ kind, int32_t *__restrict__ ref,
                                                        int32_t *__restrict__ nc, int32_t *__restrict__ nd,
                                                        float *__restrict__ zr, long long *__restrict__ ar) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= P) return;
  int y, x;
  dt_perim_cell(w.H, w.W, i, y, x);
  // ragged last tile row / column: see k_fa_rank_summary (the cell has no perimeter node)
  const bool has_node = dt_slot_of(y % TH, x % TW) >= 0;
  unsigned long long s = has_node ? nodes[dt_node_of(y, x, tiles_x)] : fht_pack(FHT_DEAD, 0, FHT_DONE);
  uint32_t ptr = (uint32_t)(s >> 32), ncf = (uint32_t)(s & 0xFFFFu);
  uint8_t k = (uint8_t)K_DEAD;
  int32_t r = -1;
  float z = DT_NODATA;
  long long a = 0;
  if (ncf & FHT_DONE) {
    if (ptr != FHT_DEAD) {
      k = (uint8_t)K_RIVER;
      r = (int32_t)ptr;
      long long o = (long long)(ptr / (uint32_t)w.W) * w.ld + (ptr % (uint32_t)w.W);
      if (dem) z = dem[o];
      if (acc32) a = (long long)acc32[o];
    }
  } else if (ptr >= nnodes) {  // parked on a ghost: leaves the rank
    k = (uint8_t)K_REXIT;
    r = (int32_t)(ptr - nnodes);
  }  // else: not resolved within the cap -> dead
  kind[i] = k;
  ref[i] = r;
  nc[i] = (int32_t)(ncf & 0x7FFFu);
  nd[i] = (int32_t)((s >> 16) & 0xFFFFu);
  zr[i] = z;
  ar[i] = a;
}

// write the rank-level result of every rank exit into its ghost: res_ok[i] != 0 -> the path leaving
// through ring cell i ends on a river cell after (res_nc, res_nd) further moves (payload key = i)
__global__ __launch_bounds__(256) void k_fh_ghost_set(unsigned long long *__restrict__ nodes, uint32_t nnodes,
                                                     int64_t P, const uint8_t *__restrict__ res_ok,
                                                     const int32_t *__restrict__ res_nc,
                                                     const int32_t *__restrict__ res_nd) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= P) return;
  unsigned long long o = fht_pack(FHT_DEAD, 0, FHT_DONE);
  if (res_ok[i]) o = fht_pack(FHT_REMOTE | (uint32_t)i, (uint32_t)res_nd[i], (uint32_t)res_nc[i] | FHT_DONE);
  nodes[nnodes + i] = o;
}

// pass 3: resolved node words -> rasters
// optional fused epilogue: GFI and ln(hl/H) (gfi.py:268-294, :404-440) from the HAND and river accumulation
// this kernel has in registers and the cell's own accumulation -- saves the separate pass over hand,
// A_river and fac (and, when a_river is not asked for, that raster altogether)
struct FhGfi {
  float *gfi, *lnhlh;  // NULL: not fused
  double expo, c0;     // n, ln b + n ln(size^2)
  const DtLogEntry *tab;
};
struct FhRemote {  // payload of rank exits (multi-GPU), all indexed by core-ring index; NULL when unused
  const long long *gidx;
  const float *zr;
  const long long *ar;  // 64 bits whatever the raster's width: a river of another rank may carry >= 2^31 cells
};
// per exit slot of a tile: height and accumulation of the river cell the exit resolves to.  One 64-bit LDS word with
// the int32 accumulation raster, two arrays with the int64 one.
template <typename AccT>
struct FhPay;
template <>
struct FhPay<int32_t> {
  unsigned long long v[PS];
  __device__ __forceinline__ void set(int i, float z, int32_t a) {
    v[i] = ((unsigned long long)(uint32_t)a << 32) | (unsigned long long)__float_as_uint(z);
  }
  __device__ __forceinline__ void get(int i, float &z, int32_t &a) const {
    const unsigned long long p = v[i];
    z = __uint_as_float((uint32_t)p);
    a = (int32_t)(uint32_t)(p >> 32);
  }
};
template <>
struct FhPay<long long> {
  long long av[PS];
  float zv[PS];
  __device__ __forceinline__ void set(int i, float z, long long a) {
    av[i] = a;
    zv[i] = z;
  }
  __device__ __forceinline__ void get(int i, float &z, long long &a) const {
    z = zv[i];
    a = av[i];
  }
};

typedef float fh_v4f __attribute__((ext_vector_type(4)));
typedef int fh_v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void fh_store4(float *p, float a, float b, float c, float d) {
  fh_v4f v = {a, b, c, d};
  __builtin_nontemporal_store(v, reinterpret_cast<fh_v4f *>(p));
}
__device__ __forceinline__ void fh_store4(int32_t *p, int32_t a, int32_t b, int32_t c, int32_t d) {
  fh_v4i v = {a, b, c, d};
  __builtin_nontemporal_store(v, reinterpret_cast<fh_v4i *>(p));
}
typedef long long fh_v2l __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void fh_store4(long long *p, long long a, long long b, long long c, long long d) {
  fh_v2l v0 = {a, b}, v1 = {c, d};
  __builtin_nontemporal_store(v0, reinterpret_cast<fh_v2l *>(p));
  __builtin_nontemporal_store(v1, reinterpret_cast<fh_v2l *>(p + 2));
}

// RANKED: the multi-GPU form (rank-exit payloads `rem`, ghost nodes, global int64 river indices); the
// single-raster form leaves all of that out of the register budget
template <bool RANKED, int VH, int MINW, typename AccT>
__global__ __launch_bounds__(256, MINW) void k_fh_tile3(const uint8_t *__restrict__ fdr,
                                                 const float *__restrict__ dem,
                                                 const AccT *__restrict__ acc32, DtWin w, int tiles_x,
                                                 uint32_t nnodes, const unsigned long long *__restrict__ nodes,
                                                 const unsigned long long *__restrict__ cache,
                                                 const uint8_t *__restrict__ cache_wide,
                                                 FhRemote rem, double px, float *__restrict__ fdist,
                                                 int32_t *__restrict__ idx32, long long *__restrict__ idx64,
                                                 float *__restrict__ hand, AccT *__restrict__ a_river,
                                                 FhGfi G) {
  // per exit slot: the resolved word of the node the exit cell steps onto (+ the step), and the payload
  // {river height, river accumulation} of the river cell it resolves to.  4 KiB of LDS: the kernel streams
  // 28 B/cell and needs the occupancy, not a 32 KiB per-cell table.
  __shared__ unsigned long long s_x[PS];
  __shared__ FhPay<AccT> s_pay;
  __shared__ DtLogEntry s_tab[DT_LOGTAB_N];  // 2 KiB: the GFI epilogue's logarithm table
  if (G.gfi) dt_math_stage(G.tab, s_tab);  // before the barrier every path below passes
  const int tile = dt_tile_of_block((int)blockIdx.x, (int)gridDim.x);
  const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
  const int y0 = ty * TH, x0 = tx * TW;
  const bool wide = cache_wide[tile] != 0;  // block-uniform
  if (threadIdx.x < PS) {
    int ly, lx;
    dt_cell_of_slot(threadIdx.x, ly, lx);
    int f = ly * TW + lx;
    unsigned long long s = fh_cache_get(cache, wide, tile, f);
    uint32_t sp = (uint32_t)(s >> 32);
    unsigned long long o = fht_pack(FHT_DEAD, 0, FHT_DONE);
    float zr = DT_NODATA;
    AccT ar = -100;
    // only exit cells (a finished word pointing at itself with kind EXIT / REXIT) are looked up
    bool ex = sp == ((uint32_t)f | (K_EXIT << 12)), rex = sp == ((uint32_t)f | (K_REXIT << 12));
    if (ex || rex) {
      int dy, dx;
      dt_d8_delta(fdr[(long long)(y0 + ly) * w.ld + x0 + lx], dy, dx);
      size_t node = ex ? (size_t)dt_node_of(y0 + ly + dy, x0 + lx + dx, tiles_x)
                       : (size_t)nnodes + (size_t)dt_perim_index(w.H, w.W, y0 + ly, x0 + lx);
      unsigned long long ns = nodes[node];
      uint32_t nptr = (uint32_t)(ns >> 32), nnd = (uint32_t)((ns >> 16) & 0xFFFFu);
      uint32_t nncf = (uint32_t)(ns & 0xFFFFu);
      if (RANKED && !(nncf & FHT_DONE) && nptr >= nnodes && (size_t)nptr != node) {
        // parked on the ghost of a rank exit (the local doubling left it pointing straight at it): one more
        // hop picks up what the rank-level solve wrote there
        unsigned long long gs = nodes[nptr];
        uint32_t gncf = (uint32_t)(gs & 0xFFFFu);
        nnd += (uint32_t)((gs >> 16) & 0xFFFFu);
        // the two cardinal counts are <= 20000 each: their sum can exceed the word's 15-bit field, so the cap is
        // tested on the full sum before anything is packed again (a wrapped count once passed the test: a path of
        // 32800 moves across two ranks came out as 32 moves, tests/test_gpu_acc64.py)
        const uint32_t cnt = (nncf & 0x7FFFu) + (gncf & 0x7FFFu);
        nptr = (uint32_t)(gs >> 32);
        if ((gncf & FHT_DONE) && nptr != FHT_DEAD && cnt + nnd <= FHT_CAP) nncf = cnt | FHT_DONE;
        else nptr = FHT_DEAD;
      }
      bool diag = dy != 0 && dx != 0;
      // not done after all rounds == longer than the cap (or an unresolved rank exit)
      if ((nncf & FHT_DONE) && nptr != FHT_DEAD) {
        o = fht_pack(nptr, nnd + (diag ? 1u : 0u), ((nncf & 0x7FFFu) + (diag ? 0u : 1u)) | FHT_DONE);
        if (RANKED && (nptr & FHT_REMOTE)) {
          zr = rem.zr[nptr & ~FHT_REMOTE];
          ar = (AccT)rem.ar[nptr & ~FHT_REMOTE];
        } else {
          long long ro = (long long)(nptr / (uint32_t)w.W) * w.ld + (nptr % (uint32_t)w.W);
          if (dem) zr = dem[ro];
          if (acc32) ar = acc32[ro];
        }
      }
    }
    s_x[threadIdx.x] = o;
    s_pay.set((int)threadIdx.x, zr, ar);
  }
  // block-uniform: whole 64-cell rows of this tile are inside the core and every raster row is 16-byte
  // aligned -> each lane handles 4 consecutive cells with 16-byte loads and stores
  const bool vec = x0 + TW <= w.W && (w.ld & 3) == 0 && (!dem || ((uintptr_t)dem & 15) == 0) &&
                   (!fdist || ((uintptr_t)fdist & 15) == 0) && (!idx32 || ((uintptr_t)idx32 & 15) == 0) &&
                   (!idx64 || ((uintptr_t)idx64 & 15) == 0) && (!hand || ((uintptr_t)hand & 15) == 0) &&
                   (!a_river || ((uintptr_t)a_river & 15) == 0) &&
                   (!G.gfi || ((((uintptr_t)G.gfi | (uintptr_t)G.lnhlh | (uintptr_t)acc32) & 15) == 0));
  const double dcard = px, ddiag = px * sqrt(2.0);
  struct CellOut {
    float fd, h;
    int32_t i32;
    AccT ar;
    long long i64;
  };
  // one cell: its pass-1 word s, own height z -> outputs (needs s_x / s_pay: call after the barrier)
  auto solve = [&](unsigned long long s, float z) -> CellOut {
    uint32_t ptr = (uint32_t)(s >> 32), nd = (uint32_t)((s >> 16) & 0xFFFFu);
    uint32_t ncf = (uint32_t)(s & 0xFFFFu), nc = ncf & 0x7FFFu;
    uint32_t kind = (ptr >> 12) & 7u, f = ptr & 0xFFFu;
    bool ok = false;
    uint32_t ridx = 0;  // core-local flat index of the river cell, or FHT_REMOTE | ring index
    float pz = 0.0f;  // the river cell's height and accumulation
    AccT pa = 0;
    if (ncf & FHT_DONE) {
      if (kind == K_RIVER) {
        // the path ends on a river cell of this tile: its own height / accumulation (a gather inside the
        // tile's own rows)
        ok = true;
        ridx = (uint32_t)((y0 + (int)f / TW) * w.W + x0 + (int)f % TW);
        long long ro = (long long)(y0 + (int)f / TW) * w.ld + x0 + (int)f % TW;
        pz = dem ? dem[ro] : DT_NODATA;
        pa = acc32 ? acc32[ro] : (AccT)-100;
      } else if (kind == K_EXIT || kind == K_REXIT) {
        int slot = dt_slot_of((int)f / TW, (int)f % TW);
        unsigned long long xs = s_x[slot];
        uint32_t xptr = (uint32_t)(xs >> 32);
        if (xptr != FHT_DEAD) {
          nc += (uint32_t)(xs & 0x7FFFu);
          nd += (uint32_t)((xs >> 16) & 0xFFFFu);
          ok = nc + nd <= FHT_CAP;  // flowhand.py:834-837
          ridx = xptr;
          s_pay.get(slot, pz, pa);
        }
      }
    }
    const bool remote = RANKED && ok && (ridx & FHT_REMOTE) != 0u;
    CellOut o;
    o.fd = ok ? (float)(dcard * (double)nc + ddiag * (double)nd) : DT_NODATA;
    o.i32 = (ok && !remote) ? (int32_t)ridx : -100;
    o.i64 = -100;
    if (RANKED && ok) {
      // rank mode: the river index is the GLOBAL raster's, in both widths (the 32-bit one is meaningful while the
      // global raster has <= 2^31 cells: 8 ranks of 16384^2; half the bytes of the widest output raster)
      o.i64 = remote ? rem.gidx[ridx & ~FHT_REMOTE]
                     : (long long)(w.gy0 + (int)(ridx / (uint32_t)w.W)) * w.Wg + w.gx0 + (int)(ridx % (uint32_t)w.W);
      o.i32 = (int32_t)o.i64;
    }
    o.h = DT_NODATA;
    if (z != DT_NODATA && ok) {  // flowhand.py:436
      o.h = z - pz;
      if (o.h < 0.0f && o.h != DT_NODATA) o.h = 0.0f;  // flowhand.py:438
    }
    // A_river = fac[idx] carried as payload; cells without a river cell get -100 (GFI is -100 there
    // anyway: their hand is -100, gfi.py:289)
    o.ar = ok ? pa : (AccT)-100;
    return o;
  };
  if (vec) {
    // 4 groups of 4 cells per lane, two at a time: 24 registers of loads in flight per lane and twice the
    // waves per SIMD instead of 48 and half of them
    constexpr int VPT = NT / 4 / 256;
#pragma unroll 1
    for (int h = 0; h < VPT / VH; h++) {
      uint4 wa[VH], wb[VH];
      float4 z4[VH];
#pragma unroll
      for (int u = 0; u < VH; u++) {
        int c = 4 * (threadIdx.x + 256 * (h * VH + u));
        if (wide) {
          const uint4 *cp = reinterpret_cast<const uint4 *>(cache + (size_t)tile * NT + c);
          wa[u] = cp[0];
          wb[u] = cp[1];
        } else {  // four narrow words in one 16-byte load
          uint4 v = *reinterpret_cast<const uint4 *>(reinterpret_cast<const uint32_t *>(cache + (size_t)tile * NT) + c);
          unsigned long long a = fh_cache_unpack(v.x), b2 = fh_cache_unpack(v.y), c2 = fh_cache_unpack(v.z),
                             d2 = fh_cache_unpack(v.w);
          wa[u] = make_uint4((uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b2, (uint32_t)(b2 >> 32));
          wb[u] = make_uint4((uint32_t)c2, (uint32_t)(c2 >> 32), (uint32_t)d2, (uint32_t)(d2 >> 32));
        }
        int y = y0 + c / TW;
        z4[u] = make_float4(DT_NODATA, DT_NODATA, DT_NODATA, DT_NODATA);
        if (dem && y < w.H) z4[u] = *reinterpret_cast<const float4 *>(dem + (long long)y * w.ld + x0 + c % TW);
      }
      if (h == 0) __syncthreads();  // s_x / s_pay
#pragma unroll
      for (int u = 0; u < VH; u++) {
        int c = 4 * (threadIdx.x + 256 * (h * VH + u));
        int y = y0 + c / TW;
        if (y >= w.H) continue;
        long long o = (long long)y * w.ld + x0 + c % TW;
        CellOut r0 = solve(((unsigned long long)wa[u].y << 32) | wa[u].x, z4[u].x);
        CellOut r1 = solve(((unsigned long long)wa[u].w << 32) | wa[u].z, z4[u].y);
        CellOut r2 = solve(((unsigned long long)wb[u].y << 32) | wb[u].x, z4[u].z);
        CellOut r3 = solve(((unsigned long long)wb[u].w << 32) | wb[u].z, z4[u].w);
        // every output byte is written once and not read again by this kernel: non-temporal stores
        if (fdist) fh_store4(fdist + o, r0.fd, r1.fd, r2.fd, r3.fd);
        if (idx32) fh_store4(idx32 + o, r0.i32, r1.i32, r2.i32, r3.i32);
        if (RANKED && idx64) {
          *reinterpret_cast<longlong2 *>(idx64 + o) = make_longlong2(r0.i64, r1.i64);
          *reinterpret_cast<longlong2 *>(idx64 + o + 2) = make_longlong2(r2.i64, r3.i64);
        }
        if (hand) fh_store4(hand + o, r0.h, r1.h, r2.h, r3.h);
        if (a_river) fh_store4(a_river + o, r0.ar, r1.ar, r2.ar, r3.ar);
        if (G.gfi) {
          // own accumulation (aligned like the outputs)
          AccT fx, fy, fz, fw;
          if (sizeof(AccT) == 8) {
            const longlong2 fa = *reinterpret_cast<const longlong2 *>(acc32 + o);
            const longlong2 fb = *reinterpret_cast<const longlong2 *>(acc32 + o + 2);
            fx = (AccT)fa.x; fy = (AccT)fa.y; fz = (AccT)fb.x; fw = (AccT)fb.y;
          } else {
            const int4 f = *reinterpret_cast<const int4 *>(acc32 + o);
            fx = (AccT)f.x; fy = (AccT)f.y; fz = (AccT)f.z; fw = (AccT)f.w;
          }
          float4 g, l;
          // one cell after the other (the scheduler would interleave the four float64 chains: 105 VGPRs)
          dt_gfi_both_cell(r0.h, r0.ar, fx, G.expo, G.c0, s_tab, g.x, l.x);
          __builtin_amdgcn_sched_barrier(0);
          dt_gfi_both_cell(r1.h, r1.ar, fy, G.expo, G.c0, s_tab, g.y, l.y);
          __builtin_amdgcn_sched_barrier(0);
          dt_gfi_both_cell(r2.h, r2.ar, fz, G.expo, G.c0, s_tab, g.z, l.z);
          __builtin_amdgcn_sched_barrier(0);
          dt_gfi_both_cell(r3.h, r3.ar, fw, G.expo, G.c0, s_tab, g.w, l.w);
          fh_store4(G.gfi + o, g.x, g.y, g.z, g.w);
          fh_store4(G.lnhlh + o, l.x, l.y, l.z, l.w);
        }
        __builtin_amdgcn_sched_barrier(0);  // finish this group of 4 cells before the next one starts
      }
    }
    return;
  }
  // ragged tiles and unaligned rasters: one cell at a time (rare: keep it small, not fast)
  __syncthreads();
#pragma unroll 1
  for (int j = 0; j < CPT; j++) {
    int c = threadIdx.x + 256 * j;
    int y = y0 + c / TW, x = x0 + c % TW;
    if (y >= w.H || x >= w.W) continue;
    long long o = (long long)y * w.ld + x;
    CellOut r = solve(fh_cache_get(cache, wide, tile, c), dem ? dem[o] : DT_NODATA);
    if (fdist) fdist[o] = r.fd;
    if (idx32) idx32[o] = r.i32;
    if (RANKED && idx64) idx64[o] = r.i64;
    if (hand) hand[o] = r.h;
    if (a_river) a_river[o] = r.ar;
    if (G.gfi) {
      float g, l;
      dt_gfi_both_cell(r.h, r.ar, acc32[o], G.expo, G.c0, s_tab, g, l);
      G.gfi[o] = g;
      G.lnhlh[o] = l;
    }
  }
}

struct FhScratch {
  unsigned long long *nodes, *cache;
  uint8_t *cache_wide;  // one flag per tile
  int64_t nnodes, ntiles, P;
  int tiles_x;
};
static FhScratch fh_layout(const DtWin &w, void *scratch) {
  FhScratch f;
  f.tiles_x = (w.W + TW - 1) / TW;
  f.ntiles = (int64_t)f.tiles_x * ((w.H + TH - 1) / TH);
  f.nnodes = f.ntiles * PS;
  f.P = dt_perim_count(w.H, w.W);
  f.nodes = (unsigned long long *)scratch;
  f.cache = (unsigned long long *)((char *)scratch + dt_align256(((size_t)f.nnodes + (size_t)f.P) * 8));
  f.cache_wide = (uint8_t *)f.cache + dt_align256((size_t)f.ntiles * NT * 8) + 256;  // after the round flags
  return f;
}
size_t dt_flowhand_tiled_scratch(int64_t H, int64_t W) {
  int64_t ntiles = ((W + TW - 1) / TW) * ((H + TH - 1) / TH);
  return dt_align256(((size_t)ntiles * PS + (size_t)dt_perim_count((int)H, (int)W)) * 8) +
         dt_align256((size_t)ntiles * NT * 8) + 256 + dt_align256((size_t)ntiles);
}

// what follows the narrow tile pass in phase 1: the 64-bit solve for the tiles it had to give up (usually none: the
// launch is 2048 workgroups that look at 32 flags each and leave), the ghosts, the perimeter node doubling
static int fh_local_tail(hipStream_t s, const DtWin &w, const uint8_t *fdr, const int8_t *river, const FhScratch &f) {
  dim3 gt((unsigned)f.ntiles), b(256), gn((unsigned)((f.nnodes + f.P + 255) / 256));
  hipLaunchKernelGGL(k_fh_tile1, dim3(gt.x < 2048u ? gt.x : 2048u), b, 0, s, fdr, river, w, f.tiles_x,
                     (uint32_t)f.nnodes, f.nodes, f.cache, f.cache_wide, 1, (int)f.ntiles);
  hipLaunchKernelGGL(k_fh_ghost_init, dim3((unsigned)((f.P + 255) / 256)), b, 0, s, f.nodes, (uint32_t)f.nnodes, f.P);
  // 8 launches of 3 jumps resolve every chain of <= 20000 moves (each node hop is >= 1 move; the resolved
  // distance at least quadruples per launch: 4^8 > 20000).  HAND's first phase at 16384^2: 1.40 ms with 15 x 1,
  // 1.33 with 10 x 2, 1.30 with 8 x 3 or 7 x 4.
  int *flags = (int *)((char *)f.cache + dt_align256((size_t)f.ntiles * NT * 8));  // the layout's spare 256 bytes
  DT_HIP(hipMemsetAsync(flags, 0, 64, s));
  dim3 gj(gn.x < 4096u ? gn.x : 4096u);  // grid-stride: a quiet round costs a few microseconds
  for (int r = 0; r < 8; r++)
    hipLaunchKernelGGL(k_fh_node_jump, gj, b, 0, s, f.nodes, f.nnodes + f.P, flags, r, 3);
  return DT_OK;
}

// phase 1: tile pass + perimeter node doubling (rank exits park on their ghosts)
int dt_launch_fh_local(hipStream_t s, const DtWin &w, const uint8_t *fdr, const int8_t *river, void *scratch,
                       size_t scratch_bytes) {
  if (w.H == 0 || w.W == 0) return DT_OK;
  DT_REQUIRE(scratch_bytes >= dt_flowhand_tiled_scratch(w.H, w.W), "scratch too small");
  FhScratch f = fh_layout(w, scratch);
  DT_REQUIRE(f.nnodes + f.P < 0x7FFFFFF0ll, "raster too large for one device tile");
  hipLaunchKernelGGL(k_fh_tile1n, dim3((unsigned)f.ntiles), dim3(256), 0, s, fdr, river, w, f.tiles_x,
                     (uint32_t)f.nnodes, f.nodes, f.cache, f.cache_wide);
  return fh_local_tail(s, w, fdr, river, f);
}

// Flow accumulation's phase 2 (inflow from other ranks, poison, last tile pass with the river mask) and HAND's phase 1
// in one go.  `fa_scratch` holds the state dt_launch_fa_local left, `fh_scratch` receives HAND's.  In the common form
// (int32 accumulation, rows of whole 64-cell tiles, 16-byte aligned rasters) the two tile passes are ONE kernel
// (k_fa3fh1); otherwise they run one after the other.  Same results either way.
int dt_launch_fa_finish_fh_local(hipStream_t s, const DtWin &w, const uint8_t *fdr, const float *dem, void *fa_scratch,
                                 void *fh_scratch, size_t fh_bytes, const unsigned long long *ext_perim,
                                 int64_t river_thr, void *acc, int acc64, int8_t *river, int *status,
                                 const uint8_t *nod4, int ldm) {
  if (w.H == 0 || w.W == 0) return DT_OK;
  DT_REQUIRE(fh_bytes >= dt_flowhand_tiled_scratch(w.H, w.W), "scratch too small");
  DT_REQUIRE(river != nullptr, "HAND needs the river mask");
  FaScratch f = fa_layout(w, fa_scratch);
  FhScratch h = fh_layout(w, fh_scratch);
  DT_REQUIRE(h.nnodes + h.P < 0x7FFFFFF0ll, "raster too large for one device tile");
  dim3 gt((unsigned)f.ntiles), b(256), gn((unsigned)((f.nnodes + 255) / 256));
  if (ext_perim) {
    int64_t P = dt_perim_count(w.H, w.W);
    hipLaunchKernelGGL(k_fa_propagate, dim3((unsigned)((P + 255) / 256)), b, 0, s, w, f.tiles_x, f.nnodes, f.nxt,
                       ext_perim, P, f.ext);
  }
  hipLaunchKernelGGL(k_fa_poison, gn, b, 0, s, f.rec, f.nnodes, f.entry_of, f.state, f.ext);
  const bool fused = !acc64 && dt_debug_get(DT_DBG_NO_FUSED_FA_FH) == 0 && w.W % TW == 0 && (w.ld & 3) == 0 &&
                     (((uintptr_t)acc | (uintptr_t)dem) & 15) == 0 && ((uintptr_t)river & 3) == 0;
  if (fused) {
    int32_t thr = river_thr > 2147483647ll ? 2147483647 : (river_thr < -2147483647ll ? -2147483647 : (int32_t)river_thr);
    // the single raster's window starts on the mask's 4-cell grid; a rank's window need not: the DEM there
    const bool use_mask = nod4 != nullptr && w.halo == 0 && w.gx0 == 0 && w.gy0 == 0;
    if (use_mask)
      hipLaunchKernelGGL(k_fa3fh1<2>, gt, b, 0, s, fdr, dem, nod4, ldm, w, f.tiles_x, f.ext, f.loc16, (int32_t *)acc, thr,
                         river, status, (uint32_t)h.nnodes, h.nodes, h.cache, h.cache_wide);
    else if (dem)
      hipLaunchKernelGGL(k_fa3fh1<1>, gt, b, 0, s, fdr, dem, nod4, ldm, w, f.tiles_x, f.ext, f.loc16, (int32_t *)acc, thr,
                         river, status, (uint32_t)h.nnodes, h.nodes, h.cache, h.cache_wide);
    else
      hipLaunchKernelGGL(k_fa3fh1<0>, gt, b, 0, s, fdr, dem, nod4, ldm, w, f.tiles_x, f.ext, f.loc16, (int32_t *)acc, thr,
                         river, status, (uint32_t)h.nnodes, h.nodes, h.cache, h.cache_wide);
  } else {
    if (acc64) {
      fa_launch_tile3<long long>(s, gt, w, fdr, dem, f, (long long *)acc, (long long)river_thr, river, status);
    } else {
      int32_t thr = river_thr > 2147483647ll ? 2147483647 : (river_thr < -2147483647ll ? -2147483647 : (int32_t)river_thr);
      fa_launch_tile3<int32_t>(s, gt, w, fdr, dem, f, (int32_t *)acc, thr, river, status);
    }
    hipLaunchKernelGGL(k_fh_tile1n, gt, b, 0, s, fdr, river, w, h.tiles_x, (uint32_t)h.nnodes, h.nodes, h.cache,
                       h.cache_wide);
  }
  return fh_local_tail(s, w, fdr, river, h);
}

int dt_launch_fh_summary(hipStream_t s, const DtWin &w, void *scratch, const float *dem, const void *acc, int acc64,
                         uint8_t *kind, int32_t *ref, int32_t *nc, int32_t *nd, float *zr, long long *ar) {
  FhScratch f = fh_layout(w, scratch);
  if (f.P == 0) return DT_OK;
  dim3 g((unsigned)((f.P + 255) / 256)), b(256);
  if (acc64)
    hipLaunchKernelGGL(k_fh_rank_summary<long long>, g, b, 0, s, w, f.tiles_x, (uint32_t)f.nnodes, f.nodes, dem,
                       (const long long *)acc, f.P, kind, ref, nc, nd, zr, ar);
  else
    hipLaunchKernelGGL(k_fh_rank_summary<int32_t>, g, b, 0, s, w, f.tiles_x, (uint32_t)f.nnodes, f.nodes, dem,
                       (const int32_t *)acc, f.P, kind, ref, nc, nd, zr, ar);
  return DT_OK;
}

// phase 2: optional rank-exit results (res_* and rem_* indexed by core-ring index), final tile pass
// `acc` / `a_river` are int32_t* (acc64 == 0) or int64_t* rasters; rem_ar is 64-bit either way
int dt_launch_fh_finish(hipStream_t s, const DtWin &w, const float *dem, const uint8_t *fdr,
                        const int8_t *river, const void *acc, int acc64, double px, void *scratch,
                        const uint8_t *res_ok, const int32_t *res_nc, const int32_t *res_nd,
                        const long long *rem_gidx, const float *rem_zr, const long long *rem_ar, float *fdist,
                        int32_t *idx32, long long *idx64, float *hand, void *a_river, float *gfi,
                        float *lnhlh, double n_gfi, double b_gfi, double size) {
  if (w.H == 0 || w.W == 0) return DT_OK;
  FhScratch f = fh_layout(w, scratch);
  dim3 gt((unsigned)f.ntiles), b(256), gn((unsigned)((f.nnodes + f.P + 255) / 256));
  if (res_ok) {
    hipLaunchKernelGGL(k_fh_ghost_set, dim3((unsigned)((f.P + 255) / 256)), b, 0, s, f.nodes, (uint32_t)f.nnodes,
                       f.P, res_ok, res_nc, res_nd);
    // nodes parked on a ghost point at it directly: pass 3 takes that one hop itself
  }
  FhRemote rem{rem_gidx, rem_zr, rem_ar};
  FhGfi G{nullptr, nullptr, 0.0, 0.0, nullptr};
  if (gfi && lnhlh) {
    DT_REQUIRE(dem && acc, "fused GFI needs dem and the accumulation raster");
    G = FhGfi{gfi, lnhlh, n_gfi, log(b_gfi) + n_gfi * log(size * size), dt_math_device_table(s)};
  }
  (void)river;
  const bool ranked = res_ok || idx64;
#define FH_GO(R, MW, T)                                                                                                \
  hipLaunchKernelGGL((k_fh_tile3<R, 1, MW, T>), gt, b, 0, s, fdr, dem, (const T *)acc, w, f.tiles_x, (uint32_t)f.nnodes, \
                     f.nodes, f.cache, f.cache_wide, rem, px, fdist, idx32, idx64, hand, (T *)a_river, G)
  if (acc64) {
    if (ranked) FH_GO(true, 4, long long);
    else FH_GO(false, 4, long long);
  } else {
    if (ranked) FH_GO(true, 5, int32_t);
    else FH_GO(false, 5, int32_t);
  }
#undef FH_GO
  return DT_OK;
}

// ===========================================================================================
// Rank level (multi-GPU): the small graph over the core-ring cells of ALL ranks, solved redundantly
// on every GPU from the all-gathered summary rows.  Same algorithms as the tile level: in-degree
// countdown with packed 64-bit atomics for the flow-accumulation inflow, pointer doubling for HAND.
// Node id = rank * Pmax + ring index.
// ===========================================================================================
#define RK_MAX 64
struct RkLayout {
  int ty, tx, nranks;
  long long Pmax;
  int ys[RK_MAX + 1], xs[RK_MAX + 1];  // cumulative row / column origins of the rank grid
  int Hg, Wg;
};
__device__ __forceinline__ void rk_rank_geom(const RkLayout &L, int r, int &y0, int &x0, int &H, int &W) {
  int ry = r / L.tx, rx = r - ry * L.tx;
  y0 = L.ys[ry];
  x0 = L.xs[rx];
  H = L.ys[ry + 1] - y0;
  W = L.xs[rx + 1] - x0;
}
// node of the ring cell that the D8 step `code` from ring cell i of rank r lands on, or -1 when the step
// stays inside the rank / leaves the global raster / the code is not a D8 code
__device__ __forceinline__ long long rk_step_target(const RkLayout &L, int r, long long i, uint32_t code) {
  if (!dt_d8_valid(code)) return -1;
  int y0, x0, H, W;
  rk_rank_geom(L, r, y0, x0, H, W);
  if (i >= dt_perim_count(H, W)) return -1;
  int y, x, dy, dx;
  dt_perim_cell(H, W, i, y, x);
  dt_d8_delta(code, dy, dx);
  int ty = y + dy, tx = x + dx;
  if (ty >= 0 && ty < H && tx >= 0 && tx < W) return -1;
  int gy = y0 + ty, gx = x0 + tx;
  if (gy < 0 || gy >= L.Hg || gx < 0 || gx >= L.Wg) return -1;
  int ry = 0, rx = 0;
  while (gy >= L.ys[ry + 1]) ry++;
  while (gx >= L.xs[rx + 1]) rx++;
  int r2 = ry * L.tx + rx, y2, x2, H2, W2;
  rk_rank_geom(L, r2, y2, x2, H2, W2);
  return (long long)r2 * L.Pmax + dt_perim_index(H2, W2, gy - y2, gx - x2);
}

struct RkRows {  // one all-gathered byte row per rank: field k of rank r at buf + r * rowbytes + off[k]
  const unsigned char *buf;
  long long rowbytes;
  long long off[8];
};
template <typename T>
__device__ __forceinline__ T rk_get(const RkRows &R, int field, int r, long long i) {
  return reinterpret_cast<const T *>(R.buf + (long long)r * R.rowbytes + R.off[field])[i];
}

// ---- flow accumulation: rows = {A int64, xr int32, code uint8} --------------------------------------
// A rank exit can be fed by thousands of exits of the neighbouring ranks, so the pending count gets a
// 32-bit word of its own (the tile level packs count and sum into one 64-bit word: <= 260 feeders there).
__global__ __launch_bounds__(256) void k_rk_fa_link(RkLayout L, RkRows R, int32_t *__restrict__ entry_of,
                                                   int32_t *__restrict__ parent, uint32_t *__restrict__ pending) {
  long long n = (long long)blockIdx.x * 256 + threadIdx.x;
  if (n >= (long long)L.nranks * L.Pmax) return;
  int r = (int)(n / L.Pmax);
  long long i = n - (long long)r * L.Pmax;
  long long e = rk_step_target(L, r, i, rk_get<uint8_t>(R, 2, r, i));
  int32_t par = -1;
  if (e >= 0) {
    int r2 = (int)(e / L.Pmax);
    int32_t xr = rk_get<int32_t>(R, 1, r2, e - (long long)r2 * L.Pmax);
    if (xr >= 0) {
      par = (int32_t)((long long)r2 * L.Pmax + xr);
      atomicAdd(&pending[par], 1u);
    }
  }
  entry_of[n] = (int32_t)e;
  parent[n] = par;
}
// sources of the forest of rank exits: exits nothing feeds (decided before the countdown starts changing `pending`)
__global__ __launch_bounds__(256) void k_rk_fa_sources(long long nn, const int32_t *__restrict__ entry_of,
                                                      const uint32_t *__restrict__ pending,
                                                      uint8_t *__restrict__ is_src) {
  long long n = (long long)blockIdx.x * 256 + threadIdx.x;
  if (n >= nn) return;
  is_src[n] = (entry_of[n] >= 0 && pending[n] == 0u) ? 1 : 0;
}
__global__ __launch_bounds__(256) void k_rk_fa_reduce(RkLayout L, RkRows R, const int32_t *__restrict__ entry_of,
                                                     const int32_t *__restrict__ parent,
                                                     const uint8_t *__restrict__ is_src,
                                                     uint32_t *__restrict__ pending,
                                                     unsigned long long *__restrict__ sum,
                                                     unsigned long long *__restrict__ ext) {
  long long n = (long long)blockIdx.x * 256 + threadIdx.x;
  long long nn = (long long)L.nranks * L.Pmax;
  if (n >= nn) return;
  if (!is_src[n]) return;
  long long q = n;
  int rq = (int)(q / L.Pmax);
  unsigned long long A = (unsigned long long)rk_get<long long>(R, 0, rq, q - (long long)rq * L.Pmax);
  for (long long it = 0; it < nn; it++) {
    atomicAdd(&ext[entry_of[q]], A);
    int32_t p = parent[q];
    if (p < 0) break;
    atomicAdd(&sum[p], A);
    __threadfence();  // the sum lands before the count drops: whoever takes the count to 0 sees every addend
    if (atomicSub(&pending[p], 1u) != 1u) break;
    __threadfence();
    int rp = (int)(p / L.Pmax);
    A = (unsigned long long)rk_get<long long>(R, 0, rp, p - (long long)rp * L.Pmax) +
        __hip_atomic_load(&sum[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    q = p;
  }
}
// exits whose count never reached 0 sit on a D8 cycle spanning ranks
__global__ __launch_bounds__(256) void k_rk_fa_poison(RkLayout L, const int32_t *__restrict__ entry_of,
                                                     const uint32_t *__restrict__ pending,
                                                     unsigned long long *__restrict__ ext) {
  long long n = (long long)blockIdx.x * 256 + threadIdx.x;
  if (n >= (long long)L.nranks * L.Pmax) return;
  if (entry_of[n] >= 0 && pending[n] != 0u) atomicOr(&ext[entry_of[n]], FA_CYCLE);
}

static int rk_make_layout(int ty, int tx, const int64_t *heights, const int64_t *widths, int64_t Pmax,
                          RkLayout *L) {
  DT_REQUIRE(ty >= 1 && tx >= 1 && ty <= RK_MAX && tx <= RK_MAX, "rank grid too large");
  L->ty = ty;
  L->tx = tx;
  L->nranks = ty * tx;
  L->Pmax = Pmax;
  L->ys[0] = L->xs[0] = 0;
  for (int i = 0; i < ty; i++) L->ys[i + 1] = L->ys[i] + (int)heights[i];
  for (int i = 0; i < tx; i++) L->xs[i + 1] = L->xs[i] + (int)widths[i];
  L->Hg = L->ys[ty];
  L->Wg = L->xs[tx];
  return DT_OK;
}

size_t dt_rank_solve_scratch(int nranks, int64_t Pmax) {
  size_t nn = (size_t)nranks * (size_t)Pmax;
  return dt_align256(nn * 8) * 2 + dt_align256(nn * 4) * 3 + dt_align256(nn) + 256;
}

// ext_out[i] (device, P_rank entries) = inflow arriving at ring cell i of `rank` (bit 63 = cycle)
int dt_launch_rank_solve_flowacc(hipStream_t s, int ty, int tx, const int64_t *heights, const int64_t *widths,
                                 int64_t Pmax, const void *rows, int64_t rowbytes, const int64_t *offs, int rank,
                                 int64_t P_rank, void *scratch, unsigned long long *ext_out) {
  RkLayout L;
  DT_TRY(rk_make_layout(ty, tx, heights, widths, Pmax, &L));
  RkRows R;
  R.buf = (const unsigned char *)rows;
  R.rowbytes = rowbytes;
  for (int k = 0; k < 8; k++) R.off[k] = k < 3 ? offs[k] : 0;
  size_t nn = (size_t)L.nranks * (size_t)Pmax;
  if (nn == 0) return DT_OK;
  char *p = (char *)scratch;
  unsigned long long *sum = (unsigned long long *)p;  p += dt_align256(nn * 8);  // sum, ext, pending: one memset
  unsigned long long *ext = (unsigned long long *)p;  p += dt_align256(nn * 8);
  uint32_t *pending = (uint32_t *)p;  p += dt_align256(nn * 4);
  int32_t *entry_of = (int32_t *)p;  p += dt_align256(nn * 4);
  int32_t *parent = (int32_t *)p;  p += dt_align256(nn * 4);
  uint8_t *is_src = (uint8_t *)p;
  DT_HIP(hipMemsetAsync(sum, 0, dt_align256(nn * 8) * 2 + dt_align256(nn * 4), s));
  dim3 g((unsigned)((nn + 255) / 256)), b(256);
  hipLaunchKernelGGL(k_rk_fa_link, g, b, 0, s, L, R, entry_of, parent, pending);
  hipLaunchKernelGGL(k_rk_fa_sources, g, b, 0, s, (long long)nn, entry_of, pending, is_src);
  hipLaunchKernelGGL(k_rk_fa_reduce, g, b, 0, s, L, R, entry_of, parent, is_src, pending, sum, ext);
  hipLaunchKernelGGL(k_rk_fa_poison, g, b, 0, s, L, entry_of, pending, ext);
  DT_HIP(hipMemcpyAsync(ext_out, ext + (size_t)rank * Pmax, (size_t)P_rank * 8, hipMemcpyDeviceToDevice, s));
  return DT_OK;
}

// ---- HAND: rows = {ref int32, nc int32, nd int32, zr float, ar int64, kind uint8, ringcode uint8} ----
__global__ __launch_bounds__(256) void k_rk_fh_build(RkLayout L, RkRows R, unsigned long long *__restrict__ nodes) {
  long long n = (long long)blockIdx.x * 256 + threadIdx.x;
  if (n >= (long long)L.nranks * L.Pmax) return;
  int r = (int)(n / L.Pmax);
  long long i = n - (long long)r * L.Pmax;
  uint32_t kind = rk_get<uint8_t>(R, 5, r, i);
  unsigned long long o = fht_pack(FHT_DEAD, 0, FHT_DONE);
  if (kind == K_RIVER) {
    o = fht_pack((uint32_t)n, (uint32_t)rk_get<int32_t>(R, 2, r, i), (uint32_t)rk_get<int32_t>(R, 1, r, i) | FHT_DONE);
  } else if (kind == K_REXIT) {
    long long ex = rk_get<int32_t>(R, 0, r, i);  // ring index of the exit cell in the same rank
    long long e = rk_step_target(L, r, ex, rk_get<uint8_t>(R, 6, r, ex));
    if (e >= 0) o = fht_pack((uint32_t)e, (uint32_t)rk_get<int32_t>(R, 2, r, i), (uint32_t)rk_get<int32_t>(R, 1, r, i));
  }
  nodes[n] = o;
}
__global__ __launch_bounds__(256) void k_rk_fh_result(RkLayout L, RkRows R, const unsigned long long *__restrict__ nodes,
                                                     int rank, long long P_rank, uint8_t *__restrict__ res_ok,
                                                     int32_t *__restrict__ res_nc, int32_t *__restrict__ res_nd,
                                                     long long *__restrict__ gidx, float *__restrict__ zr,
                                                     long long *__restrict__ ar) {
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= P_rank) return;
  uint8_t ok = 0;
  int32_t c = 0, d = 0;
  long long g = -100, a = 0;
  float z = DT_NODATA;
  long long e = rk_step_target(L, rank, i, rk_get<uint8_t>(R, 6, rank, i));
  if (e >= 0) {
    unsigned long long wd = nodes[e];
    uint32_t ptr = (uint32_t)(wd >> 32), ncf = (uint32_t)(wd & 0xFFFFu);
    if ((ncf & FHT_DONE) && ptr != FHT_DEAD) {
      ok = 1;
      c = (int32_t)(ncf & 0x7FFFu);
      d = (int32_t)((wd >> 16) & 0xFFFFu);
      int rt = (int)(ptr / (uint32_t)L.Pmax);
      long long it = (long long)ptr - (long long)rt * L.Pmax;
      int y0, x0, H, W;
      rk_rank_geom(L, rt, y0, x0, H, W);
      int ref = rk_get<int32_t>(R, 0, rt, it);  // core-local flat index of the river cell
      g = (long long)(y0 + ref / W) * L.Wg + x0 + ref % W;
      z = rk_get<float>(R, 3, rt, it);
      a = rk_get<long long>(R, 4, rt, it);
    }
  }
  res_ok[i] = ok;
  res_nc[i] = c;
  res_nd[i] = d;
  gidx[i] = g;
  zr[i] = z;
  ar[i] = a;
}

int dt_launch_rank_solve_flowhand(hipStream_t s, int ty, int tx, const int64_t *heights, const int64_t *widths,
                                  int64_t Pmax, const void *rows, int64_t rowbytes, const int64_t *offs, int rank,
                                  int64_t P_rank, void *scratch, uint8_t *res_ok, int32_t *res_nc,
                                  int32_t *res_nd, long long *gidx, float *zr, long long *ar) {
  RkLayout L;
  DT_TRY(rk_make_layout(ty, tx, heights, widths, Pmax, &L));
  RkRows R;
  R.buf = (const unsigned char *)rows;
  R.rowbytes = rowbytes;
  for (int k = 0; k < 8; k++) R.off[k] = k < 7 ? offs[k] : 0;
  size_t nn = (size_t)L.nranks * (size_t)Pmax;
  if (nn == 0 || P_rank == 0) return DT_OK;
  DT_REQUIRE(nn < 0x7FFFFFF0ull, "too many ring cells");
  unsigned long long *nodes = (unsigned long long *)scratch;
  dim3 g((unsigned)((nn + 255) / 256)), b(256);
  hipLaunchKernelGGL(k_rk_fh_build, g, b, 0, s, L, R, nodes);
  // every hop between ranks is >= 1 move: 8 launches of three jumps (>= 4 x each) cover the 20000-move cap
  for (int r = 0; r < 8; r++) hipLaunchKernelGGL(k_fh_node_jump, g, b, 0, s, nodes, (int64_t)nn, (int *)nullptr, r, 3);
  hipLaunchKernelGGL(k_rk_fh_result, dim3((unsigned)((P_rank + 255) / 256)), b, 0, s, L, R, nodes, rank,
                     (long long)P_rank, res_ok, res_nc, res_nd, gidx, zr, ar);
  return DT_OK;
}
