/*
 * dt_oracle.c -- CPU ORACLE for the descriptools hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C, single-threaded restatement of the *normative* per-cell semantics of the
 * reference's Numba-CUDA kernels (NOT of its *_sequential_jit twins, which diverge --
 * SURVEY.md 2.2).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this library; the product (descriptools_amd/) never does.
 *
 * Parity status: PINNED.  Every function below is checked (tests/test_oracle_golden.py)
 * against golden vectors produced by running the reference's unmodified Python source
 * in the build container (oracle/gen_golden.py; numba replaced by the interpretive
 * stand-in in oracle/numba_standin/), and the HAND -> calibration -> class-map chain
 * against the reference's own known-answer file Example/output/hand_class.tif.
 * D8 flow direction and flow accumulation have NO reference implementation
 * (SURVEY.md 8a rows N1/N2); for those two this file IS the definition, pinned only by
 * the consumers' encoding and the bundled 12_fdr/12_fac consistency check.
 *
 * Arithmetic follows Numba's typing of the reference kernels:
 *   - DEM differences are taken in the DEM's dtype (float32 here; int16 DEMs are
 *     converted exactly to float32 at the boundary, where the differences are exact);
 *   - every "/ px", "+ 0.01", log/tan/pow is float64, rounded ONCE to float32 on store.
 *
 * Reference citations are file:line relative to /root/reference/descriptools/.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define DT_NODATA (-100.0f)

/* ------------------------------------------------------------------------------------
 * Synthetic DEM: "tilted integer fBm" (SURVEY.md 8d).  Entirely integer, so the CPU and
 * the HIP generator are bit-identical.  Heights are multiples of 1/256 m below 2^16 m
 * (exact in float32, and all differences are exact in float32).  By construction every
 * cell's S neighbour is strictly lower (sum of per-octave y-increments < tilt), so the
 * terrain has no interior pits and no flats; water leaves through the bottom row.
 * ---------------------------------------------------------------------------------- */
#define DT_SYNTH_TILT 64 /* 1/256 m per cell along +y */
#define DT_SYNTH_KMIN 5 /* coarsest..finest octave: lambda_y = 2^O .. 2^KMIN */
#define DT_SYNTH_SX 4   /* lambda_x = lambda_y >> SX */
/* The per-cell loops are independent across cells (each cell's walk reads the rasters only): bench.py's
 * all-cores CPU baseline builds this file with -fopenmp -DDT_ORACLE_OMP (Makefile target bench); the checker build
 * used by the tests is the plain sequential one. */
#ifdef DT_ORACLE_OMP
#define DT_OMP_FOR _Pragma("omp parallel for schedule(dynamic, 4096)")
#else
#define DT_OMP_FOR
#endif

static const int32_t DT_SYNTH_AMP[15] = {/* floor(16 * 2^(0.8 k)), k = log2(lambda_y) */
                                         0,    0,    0,    0,    0,     256,   445,  776,
                                         1351, 2352, 4096, 7131, 12416, 21618, 37640};

static inline uint32_t dt_hash32(uint32_t seed, uint32_t o, uint32_t ix, uint32_t iy) {
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

int dt_oracle_synth_octaves(int64_t Hg, int64_t Wg) {
  int64_t m = Hg < Wg ? Hg : Wg;
  int lg = 0;
  while ((m >> (lg + 1)) > 0) lg++;
  int O = lg - 2;
  if (O < DT_SYNTH_KMIN) O = DT_SYNTH_KMIN;
  if (O > 14) O = 14;
  return O;
}

/* height in 1/256 m units of global cell (y, x) of an Hg x Wg DEM.  Tilt runs along +y
 * only; each octave's lattice is 16x finer along x than along y, so x-gradients are free
 * to exceed the tilt (valleys, hillslopes draining E/W) while the y-increment stays
 * below it: sum_{k=5..14} (A_k / 2^k + 1) <= 56.4 < 64. */
static inline int64_t dt_synth_units(uint32_t seed, int O, int64_t Hg, int64_t Wg, int64_t y,
                                     int64_t x) {
  (void)Wg;
  int64_t z = (int64_t)DT_SYNTH_TILT * (Hg - 1 - y);
  for (int o = 0; o < O; o++) {
    int sh = O - o; /* lambda_y = 2^sh */
    if (sh < DT_SYNTH_KMIN) break;
    int shx = sh - DT_SYNTH_SX; /* lambda_x = lambda_y / 16 (>= 2) */
    uint32_t lx = (uint32_t)(x >> shx), ly = (uint32_t)(y >> sh);
    uint64_t fx = (uint64_t)(x & ((1 << shx) - 1)) << (16 - shx);
    uint64_t fy = (uint64_t)(y & ((1 << sh) - 1)) << (16 - sh);
    uint64_t v00 = dt_hash32(seed, (uint32_t)o, lx, ly) >> 16;
    uint64_t v10 = dt_hash32(seed, (uint32_t)o, lx + 1, ly) >> 16;
    uint64_t v01 = dt_hash32(seed, (uint32_t)o, lx, ly + 1) >> 16;
    uint64_t v11 = dt_hash32(seed, (uint32_t)o, lx + 1, ly + 1) >> 16;
    uint64_t top = v00 * (65536 - fx) + v10 * fx;
    uint64_t bot = v01 * (65536 - fx) + v11 * fx;
    uint64_t val = (top * (65536 - fy) + bot * fy) >> 32; /* < 65536 */
    z += (int64_t)((val * (uint64_t)DT_SYNTH_AMP[sh]) >> 16);
  }
  return z;
}

/* window [y0, y0+h) x [x0, x0+w) of the global Hg x Wg DEM; nodata_pct > 0 punches
 * seeded square nodata blobs (-100) covering roughly that percentage. */
int dt_oracle_synth_dem(uint32_t seed, int64_t Hg, int64_t Wg, int64_t y0, int64_t x0, int64_t h,
                        int64_t w, int nodata_pct, float *out) {
  int O = dt_oracle_synth_octaves(Hg, Wg);
  for (int64_t r = 0; r < h; r++)
    for (int64_t c = 0; c < w; c++) {
      int64_t y = y0 + r, x = x0 + c;
      float z = (float)dt_synth_units(seed, O, Hg, Wg, y, x) * (1.0f / 256.0f);
      if (nodata_pct > 0) {
        /* 32x32 blocks, each nodata with probability nodata_pct % */
        uint32_t hb = dt_hash32(seed ^ 0xA5A5A5A5u, 77u, (uint32_t)(x >> 5), (uint32_t)(y >> 5));
        if ((hb % 100u) < (uint32_t)nodata_pct) z = DT_NODATA;
      }
      out[r * w + c] = z;
    }
  return 0;
}

/* ------------------------------------------------------------------------------------
 * S3 slope -- slope.py:210-259 (kernel slope_gpu) with the -100 ring of slope.py:175-182
 * folded in as "neighbour outside the raster == nodata neighbour".
 *   centre <= -100 -> -100 (slope.py:231); neighbour == -100 skipped (slope.py:247);
 *   aux < (z_c - z_nb)/d, strict, scan NW,N,NE,W,(C),E,SW,S,SE (slope.py:244-258);
 *   result aux*100 stored float32 (slope.py:259, buffer dtype slope.py:193).
 * Also emits the D8 code of the neighbour that set the final maximum (row N1): the
 * first strict maximum in scan order; centre nodata -> 0; no lower neighbour -> 0 for an
 * interior cell, and for a cell on the raster border the code pointing OUT of the raster
 * (bottom row S, top row N, else left column W, right column E) -- the builder's
 * documented choice for N1, so that outlets stay distinguishable from nodata.
 * Codes (flowhand.py:801-824): 1=E 2=SE 4=S 8=SW 16=W 32=NW 64=N 128=NE.
 * ---------------------------------------------------------------------------------- */
static const int DT_DY[8] = {-1, -1, -1, 0, 0, 1, 1, 1};
static const int DT_DX[8] = {-1, 0, 1, -1, 1, -1, 0, 1};
static const uint8_t DT_CODE[8] = {32, 64, 128, 16, 1, 8, 4, 2};

int dt_oracle_slope_d8_f32(const float *dem, int64_t H, int64_t W, double px, float *slope,
                           uint8_t *fdr) {
  const double dcard = px, ddiag = px * sqrt(2.0);
  DT_OMP_FOR
  for (int64_t y = 0; y < H; y++)
    for (int64_t x = 0; x < W; x++) {
      int64_t i = y * W + x;
      float c = dem[i];
      if (c <= DT_NODATA) {
        if (slope) slope[i] = DT_NODATA;
        if (fdr) fdr[i] = 0;
        continue;
      }
      double aux = 0.0;
      uint8_t code = 0;
      for (int k = 0; k < 8; k++) {
        int64_t yy = y + DT_DY[k], xx = x + DT_DX[k];
        if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue; /* -100 ring */
        float nb = dem[yy * W + xx];
        if (nb == DT_NODATA) continue;
        volatile float diff = c - nb; /* DEM-dtype subtraction */
        double v = (double)diff / ((DT_DY[k] == 0 || DT_DX[k] == 0) ? dcard : ddiag);
        if (aux < v) {
          aux = v;
          code = DT_CODE[k];
        }
      }
      if (code == 0) {
        if (y == H - 1) code = 4;
        else if (y == 0) code = 64;
        else if (x == 0) code = 16;
        else if (x == W - 1) code = 1;
      }
      if (slope) slope[i] = (float)(aux * 100.0);
      if (fdr) fdr[i] = code;
    }
  return 0;
}

/* D8 step: returns the flat index reached from `pos` along `code`, or -1 when the code
 * is not one of the eight ESRI codes, or -2 when the move would leave the raster
 * (flowhand.py:623-798 exit tests; downslope.py:469-488). */
static inline int64_t dt_step(int64_t pos, uint8_t code, int64_t H, int64_t W, int *diag) {
  int64_t y = pos / W, x = pos - y * W;
  int dy, dx;
  switch (code) {
    case 1: dy = 0; dx = 1; break;
    case 2: dy = 1; dx = 1; break;
    case 4: dy = 1; dx = 0; break;
    case 8: dy = 1; dx = -1; break;
    case 16: dy = 0; dx = -1; break;
    case 32: dy = -1; dx = -1; break;
    case 64: dy = -1; dx = 0; break;
    case 128: dy = -1; dx = 1; break;
    default: return -1;
  }
  *diag = (dy != 0 && dx != 0);
  y += dy;
  x += dx;
  if (y < 0 || y >= H || x < 0 || x >= W) return -2;
  return y * W + x;
}

/* ------------------------------------------------------------------------------------
 * N2 flow accumulation (net-new): number of upstream cells EXCLUDING
 * self, by Kahn in-degree countdown.  Cells with fdr==0, a non-D8 code or an off-raster
 * target are sinks.  If `dem` is non-NULL, cells with dem <= -100 are set to -100.
 * ---------------------------------------------------------------------------------- */
int dt_oracle_flowacc(const uint8_t *fdr, const float *dem, int64_t H, int64_t W, int64_t *acc) {
  int64_t N = H * W;
  int32_t *indeg = (int32_t *)calloc((size_t)N, sizeof(int32_t));
  int64_t *stack = (int64_t *)malloc((size_t)N * sizeof(int64_t));
  if (!indeg || !stack) return -1;
  int diag;
  for (int64_t i = 0; i < N; i++) {
    acc[i] = 0;
    int64_t t = dt_step(i, fdr[i], H, W, &diag);
    if (t >= 0) indeg[t]++;
  }
  int64_t sp = 0;
  for (int64_t i = 0; i < N; i++)
    if (indeg[i] == 0) stack[sp++] = i;
  while (sp > 0) {
    int64_t c = stack[--sp];
    int64_t t = dt_step(c, fdr[c], H, W, &diag);
    if (t >= 0) {
      acc[t] += acc[c] + 1;
      if (--indeg[t] == 0) stack[sp++] = t;
    }
  }
  /* cells on a D8 cycle keep indeg > 0: their accumulation is undefined -> -100 */
  for (int64_t i = 0; i < N; i++)
    if (indeg[i] > 0 || (dem && dem[i] <= DT_NODATA)) acc[i] = -100;
  free(indeg);
  free(stack);
  return 0;
}

/* ------------------------------------------------------------------------------------
 * F3 flow distance + drained-to river-cell index -- flowhand.py:566-846, untiled case
 * (out[] all zero, row_start = col_start = 0, matrix_columns = W).  Literal per-cell walk.
 *   own fdr <= 0 -> -100 (:601); river cell -> 0 / own index (:609-612);
 *   leaving the raster -> -100 (:623-628, :674-677, :718-721, :760-764);
 *   dist += px | px*sqrt(2.0) in float64 (:801-824); a non-D8 code does not move, and is
 *   then caught by the revisit test;
 *   arrival on fdr == 0 -> -100 (:826-828); new pos equals any of the previous three
 *   positions -> -100 (:830-832); more than 20000 moves -> -100 (:834-837);
 *   distance stored float32 (:540), index emitted through a float64 buffer (:541).
 * F4 HAND -- flowhand.py:414-442: dem - dem[idx] in the DEM dtype where dem != -100 and
 *   idx != -100, negatives (other than exactly -100) clipped to 0.
 * ---------------------------------------------------------------------------------- */
int dt_oracle_flowhand(const float *dem, const uint8_t *fdr, const int8_t *river, int64_t H,
                       int64_t W, double px, float *fdist, int64_t *idx, float *hand) {
  const int64_t N = H * W;
  const double dcard = px, ddiag = px * sqrt(2.0);
  DT_OMP_FOR
  for (int64_t i = 0; i < N; i++) {
    int64_t res_idx = -100;
    float res_d = DT_NODATA;
    if (fdr[i] != 0) {
      if (river[i] == 1) {
        res_d = 0.0f;
        res_idx = i;
      } else {
        int64_t pos = i, l1 = -10, l2 = -20, l3 = -30;
        double dist = 0.0;
        int isnan_ = 0, loop = 0;
        while (river[pos] != 1) {
          int diag = 0;
          int64_t t = dt_step(pos, fdr[pos], H, W, &diag);
          if (t == -2) { isnan_ = 1; break; }
          l3 = l2; l2 = l1; l1 = pos;
          if (t >= 0) {
            pos = t;
            dist += diag ? ddiag : dcard;
          }
          if (fdr[pos] == 0) { isnan_ = 1; break; }
          if (pos == l1 || pos == l2 || pos == l3) { isnan_ = 1; break; }
          if (++loop > 20000) { isnan_ = 1; break; }
        }
        if (!isnan_) {
          res_d = (float)dist;
          res_idx = pos;
        }
      }
    }
    if (fdist) fdist[i] = res_d;
    idx[i] = res_idx;
  }
  if (hand && dem) {
    DT_OMP_FOR
    for (int64_t i = 0; i < N; i++) {
      float h = DT_NODATA;
      if (dem[i] != DT_NODATA && idx[i] != -100) {
        volatile float d = dem[i] - dem[idx[i]];
        h = d;
        if (h < 0.0f && h != DT_NODATA) h = 0.0f;
      }
      hand[i] = h;
    }
  }
  return 0;
}

/* Fast O(N) equivalent of dt_oracle_flowhand's walk, used ONLY to check large rasters in
 * tests (validated against the literal walk above at small sizes): resolves (moves,
 * diagonal moves, river index) by memoised path compression; distance is re-summed in
 * path order so it is bit-identical to the literal walk.  Outputs n_card/n_diag too. */
int dt_oracle_flowhand_fast(const uint8_t *fdr, const int8_t *river, int64_t H, int64_t W,
                            int64_t *idx, int32_t *ncard, int32_t *ndiag) {
  const int64_t N = H * W;
  /* state: 0 = unknown, 1 = on stack, 2 = done */
  uint8_t *st = (uint8_t *)calloc((size_t)N, 1);
  int64_t *stack = (int64_t *)malloc((size_t)N * sizeof(int64_t));
  if (!st || !stack) return -1;
  for (int64_t i = 0; i < N; i++) {
    if (st[i] == 2) continue;
    int64_t sp = 0, cur = i;
    /* descend until a resolved / terminal cell */
    for (;;) {
      if (st[cur] == 2) break;
      if (st[cur] == 1) { /* cycle: everything on it is dead */
        idx[cur] = -100; ncard[cur] = 0; ndiag[cur] = 0; st[cur] = 2;
        break;
      }
      if (fdr[cur] == 0) { idx[cur] = -100; ncard[cur] = 0; ndiag[cur] = 0; st[cur] = 2; break; }
      if (river[cur] == 1) { idx[cur] = cur; ncard[cur] = 0; ndiag[cur] = 0; st[cur] = 2; break; }
      int diag = 0;
      int64_t t = dt_step(cur, fdr[cur], H, W, &diag);
      if (t < 0) { idx[cur] = -100; ncard[cur] = 0; ndiag[cur] = 0; st[cur] = 2; break; }
      st[cur] = 1;
      stack[sp++] = cur;
      cur = t;
    }
    /* unwind */
    while (sp > 0) {
      int64_t c = stack[--sp];
      int diag = 0;
      int64_t t = dt_step(c, fdr[c], H, W, &diag);
      /* arrival on fdr==0 kills the path even if that cell is a river (flowhand.py:826) */
      if (idx[t] == -100 || fdr[t] == 0) {
        idx[c] = -100; ncard[c] = 0; ndiag[c] = 0;
      } else {
        int32_t nc = ncard[t] + (diag ? 0 : 1), nd = ndiag[t] + (diag ? 1 : 0);
        if (nc + nd > 20000) { idx[c] = -100; ncard[c] = 0; ndiag[c] = 0; }
        else { idx[c] = idx[t]; ncard[c] = nc; ndiag[c] = nd; }
      }
      st[c] = 2;
    }
  }
  free(st);
  free(stack);
  return 0;
}

/* ------------------------------------------------------------------------------------
 * T2/T3 topographic index + modified topographic index -- topoindexes.py:234-261,:265-295
 *   fac <= -100 -> -100 (:252); fac == 0 -> 1 (:255); "+0.01" INSIDE tan (:257,:261);
 *   float64 math, float32 store (:210-211).
 * ---------------------------------------------------------------------------------- */
/* ---------------------------------------------------------------------------------------------
 * Hydrological conditioning (net-new; no reference counterpart -- the definition the HIP kernels
 * of dt_hydro.hip are held to): sequential priority flood for the filled surface, D8 on it, breadth-first hop
 * distances over the flats.
 * --------------------------------------------------------------------------------------------- */
typedef struct { float w; int64_t i; } dt_heap_item;
static void dt_heap_push(dt_heap_item *h, int64_t *n, dt_heap_item v) {
  int64_t k = (*n)++;
  h[k] = v;
  while (k > 0) {
    int64_t p = (k - 1) / 2;
    if (h[p].w <= h[k].w) break;
    dt_heap_item t = h[p]; h[p] = h[k]; h[k] = t;
    k = p;
  }
}
static dt_heap_item dt_heap_pop(dt_heap_item *h, int64_t *n) {
  dt_heap_item top = h[0];
  h[0] = h[--(*n)];
  int64_t k = 0;
  for (;;) {
    int64_t l = 2 * k + 1, r = l + 1, m = k;
    if (l < *n && h[l].w < h[m].w) m = l;
    if (r < *n && h[r].w < h[m].w) m = r;
    if (m == k) break;
    dt_heap_item t = h[m]; h[m] = h[k]; h[k] = t;
    k = m;
  }
  return top;
}

/* filled[c] = min over paths from c to an outlet (raster edge / next to nodata) of the highest cell on the path */
int dt_oracle_fill(const float *dem, int64_t H, int64_t W, float *filled) {
  const int64_t N = H * W;
  uint8_t *done = (uint8_t *)calloc((size_t)N, 1);
  dt_heap_item *heap = (dt_heap_item *)malloc((size_t)(N + 1) * sizeof(dt_heap_item));
  if (!done || !heap) return -1;
  int64_t hn = 0;
  for (int64_t y = 0; y < H; y++)
    for (int64_t x = 0; x < W; x++) {
      int64_t i = y * W + x;
      if (dem[i] == DT_NODATA) { filled[i] = DT_NODATA; done[i] = 1; continue; }
      int outlet = (y == 0 || x == 0 || y == H - 1 || x == W - 1);
      for (int k = 0; k < 8 && !outlet; k++)
        if (dem[(y + DT_DY[k]) * W + x + DT_DX[k]] == DT_NODATA) outlet = 1;
      if (outlet) {
        filled[i] = dem[i];
        done[i] = 1;
        dt_heap_item it = {dem[i], i};
        dt_heap_push(heap, &hn, it);
      }
    }
  while (hn > 0) {
    dt_heap_item c = dt_heap_pop(heap, &hn);
    int64_t y = c.i / W, x = c.i % W;
    for (int k = 0; k < 8; k++) {
      int64_t yy = y + DT_DY[k], xx = x + DT_DX[k];
      if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
      int64_t n = yy * W + xx;
      if (done[n]) continue;
      done[n] = 1;
      filled[n] = dem[n] > c.w ? dem[n] : c.w;
      dt_heap_item it = {filled[n], n};
      dt_heap_push(heap, &hn, it);
    }
  }
  free(done);
  free(heap);
  return 0;
}

/* D8 on the filled surface with the flats resolved (see include/descriptools_hip.h, dt_d8_conditioned_f32);
 * returns the number of valid cells left without a code */
int64_t dt_oracle_condition_d8(const float *dem, int64_t H, int64_t W, double px, float *filled, uint8_t *fdr) {
  const int64_t N = H * W;
  if (dt_oracle_fill(dem, H, W, filled) != 0) return -1;
  float *sl = (float *)malloc((size_t)N * sizeof(float));
  int64_t *queue = (int64_t *)malloc((size_t)N * sizeof(int64_t));
  uint32_t *dist = (uint32_t *)malloc((size_t)N * sizeof(uint32_t));
  if (!sl || !queue || !dist) return -1;
  dt_oracle_slope_d8_f32(filled, H, W, px, sl, fdr);
  free(sl);
  int64_t qh = 0, qt = 0;
  for (int64_t i = 0; i < N; i++) {
    dist[i] = 0u;
    if (filled[i] != DT_NODATA && fdr[i] == 0) {
      dist[i] = 0x7FFFFFFFu;
      int64_t y = i / W, x = i % W;
      for (int k = 0; k < 8; k++) {
        int64_t yy = y + DT_DY[k], xx = x + DT_DX[k];
        if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
        if (filled[yy * W + xx] == DT_NODATA) { fdr[i] = DT_CODE[k]; dist[i] = 0u; break; }
      }
    }
  }
  for (int64_t i = 0; i < N; i++)
    if (filled[i] != DT_NODATA && dist[i] == 0u) queue[qt++] = i;
  while (qh < qt) {
    int64_t c = queue[qh++];
    int64_t y = c / W, x = c % W;
    for (int k = 0; k < 8; k++) {
      int64_t yy = y + DT_DY[k], xx = x + DT_DX[k];
      if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
      int64_t n = yy * W + xx;
      if (dist[n] != 0x7FFFFFFFu || filled[n] != filled[c]) continue;
      dist[n] = dist[c] + 1u;
      queue[qt++] = n;
    }
  }
  int64_t unresolved = 0;
  for (int64_t i = 0; i < N; i++) {
    if (dist[i] == 0u) continue;
    uint8_t code = 0;
    if (dist[i] != 0x7FFFFFFFu) {
      int64_t y = i / W, x = i % W;
      /* one hop closer: the four cardinal neighbours first (N, W, E, S), then the diagonals (NW, NE, SW, SE) --
       * the order in which N1's steepest descent would rank equal drops (a cardinal step is the shorter one) */
      static const int pref[8] = {1, 3, 4, 6, 0, 2, 5, 7};
      for (int q = 0; q < 8 && !code; q++) {
        int k = pref[q];
        int64_t yy = y + DT_DY[k], xx = x + DT_DX[k];
        if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
        int64_t n = yy * W + xx;
        if (filled[n] == filled[i] && dist[n] == dist[i] - 1u) code = DT_CODE[k];
      }
    }
    if (!code) unresolved++;
    fdr[i] = code;
  }
  free(queue);
  free(dist);
  return unresolved;
}

int dt_oracle_twi(const int64_t *fac, const float *slope_rad, int64_t N, double px, double n,
                  float *ti, float *mti) {
  DT_OMP_FOR
  for (int64_t i = 0; i < N; i++) {
    if (fac[i] <= -100) {
      ti[i] = DT_NODATA;
      mti[i] = DT_NODATA;
      continue;
    }
    double a = fac[i] == 0 ? 1.0 * (px * px) : (double)fac[i] * (px * px);
    double t = tan((double)slope_rad[i] + 0.01);
    ti[i] = (float)log(a / t);
    mti[i] = (float)log(pow(a, n) / t);
  }
  return 0;
}

/* ------------------------------------------------------------------------------------
 * G1 river_accumulation -- gfi.py:119-147: A_r = fac.flat[idx] where idx != -100, else
 * fac.flat[0].   G2 GFI -- gfi.py:268-294: hand <= -100 -> -100 (:289), else
 * ln(b * (A_r*size^2)^n / (hand + 0.01)), no zero-area guard.
 * G3 ln(hl/H) -- gfi.py:404-440: own-cell fac, fac == 0 -> 1 (:432).
 * ---------------------------------------------------------------------------------- */
int dt_oracle_gfi(const float *hand, const int64_t *fac, const int64_t *idx, int64_t N, double n,
                  double b, double size, float *gfi) {
  DT_OMP_FOR
  for (int64_t i = 0; i < N; i++) {
    if (hand[i] <= DT_NODATA) {
      gfi[i] = DT_NODATA;
      continue;
    }
    int64_t ar = idx[i] != -100 ? fac[idx[i]] : fac[0];
    gfi[i] = (float)log(b * pow((double)ar * (size * size), n) / ((double)hand[i] + 0.01));
  }
  return 0;
}

int dt_oracle_lnhlh(const float *hand, const int64_t *fac, int64_t N, double n, double b,
                    double size, float *out) {
  DT_OMP_FOR
  for (int64_t i = 0; i < N; i++) {
    if (hand[i] <= DT_NODATA) {
      out[i] = DT_NODATA;
      continue;
    }
    double a = fac[i] == 0 ? 1.0 * (size * size) : (double)fac[i] * (size * size);
    out[i] = (float)log((b * pow(a, n)) / ((double)hand[i] + 0.01));
  }
  return 0;
}

/* ------------------------------------------------------------------------------------
 * D1-D3 downslope index -- downslope.py:435-532 (kernel, failures -> marker -50) followed
 * by the normative CPU repair of every -50 cell, downslope.py:161-314.  Untiled case.
 * One walk restates both, because a walk the kernel completes is identical in the repair:
 *   dem <= -100 -> -100 (downslope.py:460; repair :197 leaves it);
 *   while z0 - z[pos] < d (:468 / :208): raster-edge exit -> stop (:469-488 / :209-228);
 *   next cell == -100 -> stop WITHOUT moving (repair :231-281); a non-D8 code does not
 *   move; after 5000 loop iterations -> stop (:518-521 / :303-304);
 *   stopped by edge/nodata with dist == 0 -> 0 (:306-308) else (z0 - z[pos]) / dist
 *   (:310,:312), float64 quotient stored float32 (:352, :366).
 *   0/0 (a walk that never moves for 5000 iterations: a valid-DEM cell with a non-D8
 *   code) is undefined in the reference; the build returns 0.
 * ---------------------------------------------------------------------------------- */
int dt_oracle_downslope(const float *dem, const uint8_t *fdr, int64_t H, int64_t W, double px,
                        double dz, float *out) {
  const int64_t N = H * W;
  const double dcard = px, ddiag = px * sqrt(2.0);
  DT_OMP_FOR
  for (int64_t i = 0; i < N; i++) {
    float z0 = dem[i];
    if (z0 <= DT_NODATA) {
      out[i] = DT_NODATA;
      continue;
    }
    int64_t pos = i;
    double dist = 0.0;
    int loop = 0, stopped = 0;
    for (;;) {
      volatile float drop = z0 - dem[pos];
      if (!((double)drop < dz)) break;
      int diag = 0;
      int64_t t = dt_step(pos, fdr[pos], H, W, &diag);
      if (t == -2) { stopped = 1; break; }
      if (t >= 0) {
        if (dem[t] == DT_NODATA) { stopped = 1; break; }
        pos = t;
        dist += diag ? ddiag : dcard;
      }
      if (++loop == 5000) break;
    }
    volatile float drop = z0 - dem[pos];
    if (dist == 0.0)
      out[i] = 0.0f; /* stopped && dist==0 -> 0 (:307); 0/0 undefined -> 0 */
    else
      out[i] = (float)((double)drop / dist);
    (void)stopped;
  }
  return 0;
}

/* ------------------------------------------------------------------------------------
 * The same descriptors on a DEM that float32 cannot hold (a genuinely float64 raster, or integer heights
 * beyond 2^24 converted exactly to float64).  The reference takes every height difference in the raster's
 * OWN dtype -- slope.py:244-258 under Numba typing (float64 - float64), flowhand.py:436-438
 * `dem - dem[indices]`, downslope.py:468 -- and hands a float64 HAND to gfi.py:289-294 / :429-440, whose
 * `hand + 0.01` is then a float64 sum of a float64 value.  Pinned by tests/golden/f64.npz, the reference's own
 * run on such a raster (oracle/gen_golden.py f64).  D8 and flow accumulation take no part: they are inputs.
 * ---------------------------------------------------------------------------------- */
int dt_oracle_slope_f64(const double *dem, int64_t H, int64_t W, double px, float *slope) {
  const double dcard = px, ddiag = px * sqrt(2.0);
  DT_OMP_FOR
  for (int64_t y = 0; y < H; y++)
    for (int64_t x = 0; x < W; x++) {
      int64_t i = y * W + x;
      double c = dem[i];
      if (c <= -100.0) { /* slope.py:231 */
        slope[i] = DT_NODATA;
        continue;
      }
      double aux = 0.0;
      for (int k = 0; k < 8; k++) {
        int64_t yy = y + DT_DY[k], xx = x + DT_DX[k];
        if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue; /* -100 ring */
        double nb = dem[yy * W + xx];
        if (nb == -100.0) continue; /* slope.py:247 */
        volatile double diff = c - nb;
        double v = diff / ((DT_DY[k] == 0 || DT_DX[k] == 0) ? dcard : ddiag);
        if (aux < v) aux = v;
      }
      slope[i] = (float)(aux * 100.0);
    }
  return 0;
}

int dt_oracle_hand_f64(const double *dem, const int64_t *idx, int64_t N, double *hand) {
  DT_OMP_FOR
  for (int64_t i = 0; i < N; i++) {
    double h = -100.0;
    if (dem[i] != -100.0 && idx[i] != -100) {
      volatile double d = dem[i] - dem[idx[i]]; /* flowhand.py:436 */
      h = d;
      if (h < 0.0 && h != -100.0) h = 0.0; /* :438 */
    }
    hand[i] = h;
  }
  return 0;
}

int dt_oracle_downslope_f64(const double *dem, const uint8_t *fdr, int64_t H, int64_t W, double px, double dz,
                            float *out) {
  const int64_t N = H * W;
  const double dcard = px, ddiag = px * sqrt(2.0);
  DT_OMP_FOR
  for (int64_t i = 0; i < N; i++) {
    double z0 = dem[i];
    if (z0 <= -100.0) {
      out[i] = DT_NODATA;
      continue;
    }
    int64_t pos = i;
    double dist = 0.0;
    int loop = 0;
    for (;;) {
      volatile double drop = z0 - dem[pos]; /* downslope.py:468, the DEM's own dtype */
      if (!(drop < dz)) break;
      int diag = 0;
      int64_t t = dt_step(pos, fdr[pos], H, W, &diag);
      if (t == -2) break;
      if (t >= 0) {
        if (dem[t] == -100.0) break;
        pos = t;
        dist += diag ? ddiag : dcard;
      }
      if (++loop == 5000) break;
    }
    volatile double drop = z0 - dem[pos];
    out[i] = dist == 0.0 ? 0.0f : (float)(drop / dist);
  }
  return 0;
}

int dt_oracle_gfi_f64h(const double *hand, const int64_t *fac, const int64_t *idx, int64_t N, double n, double b,
                       double size, float *gfi) {
  DT_OMP_FOR
  for (int64_t i = 0; i < N; i++) {
    if (hand[i] <= -100.0) {
      gfi[i] = DT_NODATA;
      continue;
    }
    int64_t ar = idx[i] != -100 ? fac[idx[i]] : fac[0];
    gfi[i] = (float)log(b * pow((double)ar * (size * size), n) / (hand[i] + 0.01));
  }
  return 0;
}

int dt_oracle_lnhlh_f64h(const double *hand, const int64_t *fac, int64_t N, double n, double b, double size,
                         float *out) {
  DT_OMP_FOR
  for (int64_t i = 0; i < N; i++) {
    if (hand[i] <= -100.0) {
      out[i] = DT_NODATA;
      continue;
    }
    double a = fac[i] == 0 ? 1.0 * (size * size) : (double)fac[i] * (size * size);
    out[i] = (float)log((b * pow(a, n)) / (hand[i] + 0.01));
  }
  return 0;
}

/* ------------------------------------------------------------------------------------
 * E2+E3 confusion counts for many thresholds in one pass -- evaluation.py:90-123 and
 * :126-171.  desc is float64 (minMaxScale output, :5-9); cells equal to desc[0] or NaN
 * classify 0 (:111-121); 'under' -> desc <= th else desc >= th; benchmark map remap
 * 1 -> 2, -100 -> 0 (:149-150) is applied on the fly; counts4[t*4 + v] = number of cells
 * whose (binary + benchmark) equals v, v in 0..3.
 * ---------------------------------------------------------------------------------- */
int dt_oracle_confusion_multi(const double *desc, const int8_t *flood, int64_t N, const double *th,
                              int nth, int under, int64_t *counts4) {
  const double nod = desc[0];
  memset(counts4, 0, (size_t)nth * 4 * sizeof(int64_t));
  for (int64_t i = 0; i < N; i++) {
    double v = desc[i];
    int f = flood[i];
    if (f == 1) f = 2;
    else if (f == -100) f = 0;
    int isn = (v == nod) || isnan(v);
    for (int t = 0; t < nth; t++) {
      int bin = isn ? 0 : (under ? (v <= th[t]) : (v >= th[t]));
      int r = bin + f;
      if (r >= 0 && r <= 3) counts4[t * 4 + r]++;
    }
  }
  return 0;
}
