/*
 * descriptools_hip.h -- C ABI of libdescriptools_hip.so (MI355X / gfx950 terrain-descriptor engine)
 *
 * Drop-in boundary for the descriptools hot path.  Each entry point replaces one host<->device
 * shim (`*_cpu`) + Numba-CUDA kernel (`*_gpu`) pair of the reference (citations are
 * file:line relative to /root/reference/descriptools/).  Plain pointers and sizes only.
 *
 * Conventions:
 *   - rasters are row-major C-contiguous, H rows x W columns; nodata sentinel is -100;
 *   - DEM / HAND are float32 (int16 rasters are converted exactly by the caller);
 *   - D8 codes are ESRI: 1=E 2=SE 4=S 8=SW 16=W 32=NW 64=N 128=NE, 0 = nodata / undefined;
 *   - every function returns 0 on success and a negative DT_E* code on failure;
 *     dt_last_error() gives the message (thread-local).  Nothing falls back to the CPU.
 *
 * Two tiers:
 *   dt_<op>(...)      host pointers in / host pointers out (what the reference's *_cpu do);
 *   dt_dev_<op>(ctx,) device pointers, asynchronous on the context's stream -- the resident
 *                     chained / multi-GPU path.  Device buffers are caller-owned (hipMalloc,
 *                     torch tensors, ...); scratch is owned by the context.
 */
#ifndef DESCRIPTOOLS_HIP_H
#define DESCRIPTOOLS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DT_OK 0
#define DT_EINVAL (-1) /* bad shape / null pointer / unsupported size */
#define DT_EHIP (-2)   /* HIP runtime error (message in dt_last_error) */
#define DT_ENOMEM (-3) /* device allocation failed */
#define DT_ENODEV (-4) /* no usable GPU */

typedef struct dt_ctx dt_ctx;

/* One rank's CORE window (H x W cells) of a global Hg x Wg raster, at global offset (gy0, gx0).
 * Every raster pointer passed with a window addresses the core origin with row stride ld; `halo`
 * cells beyond the core exist in memory on every side (>= 1 when the global raster is larger).
 * Single GPU: ld = W, gy0 = gx0 = 0, Hg = H, Wg = W, halo = 0. */
typedef struct dt_window {
  int64_t H, W, ld, gy0, gx0, Hg, Wg, halo;
} dt_window;

/* ---- runtime ------------------------------------------------------------------------- */
const char *dt_last_error(void);
int dt_device_count(void);
const char *dt_version(void);
/* A/B knob: 1 = first-generation global kernels for flow accumulation / HAND / downslope (raster-wide
 * countdown, raster-wide pointer doubling, one thread per cell walking global memory), 2 = tile-hierarchical /
 * windowed (default; also selectable with the environment variable DT_FLOW_IMPL=v1).  Same results: bench.py
 * cross-checks the timed step against impl 1. */
int dt_set_flow_impl(int impl);
/* Test knobs, all 0 by default.  key 0 (DT_DBG_TWI_FLAG_ALL): the fused slope + TI + MTI stencil sends every
 * cell through its exact (cold) path as well as the fast one; key 1 (DT_DBG_TWI_PLAIN): default cache policy
 * instead of non-temporal loads / stores in that stencil (A/B timing); key 2 (DT_DBG_TWI_WX): tile geometry of that
 * stencil, 1 / 2 / 4 = tiles of 256 x 16 / 512 x 8 / 1024 x 4 cells (0 = the default); key 3 (DT_DBG_TWI_MAP): experimental
 * workgroup -> tile maps of that stencil; key 4 (DT_DBG_DS_MARGIN): margin of the downslope kernel's LDS window (16 / 20;
 * default 24); key 5 (DT_DBG_NO_FUSED_FA_FH): the last accumulation pass and HAND's first as two kernels (A/B timing). */
int dt_debug_set(int key, int value);

/* Context = one device + one stream + grow-only scratch.  `stream` may be NULL (the context
 * creates its own non-blocking stream) or an existing hipStream_t (e.g. torch's). */
int dt_ctx_create(int device, void *stream, dt_ctx **out);
int dt_ctx_destroy(dt_ctx *ctx);
int dt_ctx_set_stream(dt_ctx *ctx, void *stream);
void *dt_ctx_stream(dt_ctx *ctx);
/* Two contexts of one device as concurrent branches of a pipeline: after dt_ctx_fork the child's stream waits
 * for everything enqueued so far on the parent's; after dt_ctx_join the parent's waits for the child's.
 * Device-side ordering only (events), the host never blocks.  The chain uses it to run downslope beside the
 * flow-accumulation / HAND kernels, whose latency chains leave most of the GPU idle. */
/* Re-create the context's own stream with a scheduling priority: -1 high, 0 normal, +1 low (clamped to the device's
 * range).  DT_EINVAL for a context that runs on a caller's stream. */
int dt_ctx_set_priority(dt_ctx *ctx, int priority);
int dt_ctx_fork(dt_ctx *parent, dt_ctx *child);
int dt_ctx_join(dt_ctx *parent, dt_ctx *child);
int dt_ctx_sync(dt_ctx *ctx);
/* HIP graphs.  Everything enqueued on the context's stream between dt_ctx_capture_begin and dt_ctx_capture_end
 * (dt_dev_* calls on this context, and on contexts forked from it and joined again) is recorded instead of run;
 * dt_graph_launch replays it with ONE launch on a context of the same device.  Capture a step whose buffers and
 * workspaces already exist (run it once first): allocation and synchronisation inside a capture fail.  The
 * pointers the calls were given are baked in.  No reference counterpart (the reference launches kernel by
 * kernel, descriptools/slope.py:190-200).  It frees the host (one call per step instead of ~45), it does not
 * shorten the step: the launches are asynchronous and already keep ahead of the GPU at every raster size tried. */
typedef struct dt_graph dt_graph;
int dt_ctx_capture_begin(dt_ctx *ctx);
int dt_ctx_capture_end(dt_ctx *ctx, dt_graph **out);
int dt_graph_launch(dt_graph *graph, dt_ctx *ctx);
int dt_graph_destroy(dt_graph *graph);
/* Sticky status bits raised by kernels since the last call (synchronises the context's stream, clears them).
 * DT_STATUS_ACC_OVERFLOW: a flow accumulation value of a multi-rank raster may have reached 2^31 cells while the step
 * ran with int32 accumulation rasters (the `_w` entry points; a device tile is < 2^31 cells), so its results are
 * not valid: rasters of more than 2^31 cells go through the `_w_a64` entry points (int64 rasters), which never raise it. */
#define DT_STATUS_ACC_OVERFLOW 1
/* DT_STATUS_NOT_CONVERGED: dt_dev_condition_d8_async's budget of rounds ran out before the fixed point (or a flat
 * cell was left without a code): the conditioned rasters of that step are not valid. */
#define DT_STATUS_NOT_CONVERGED 2
int dt_ctx_status(dt_ctx *ctx, int32_t *out);
int64_t dt_ctx_scratch_bytes(dt_ctx *ctx);

/* ---- host-pointer tier (drop-in for the reference's *_cpu shims) -----------------------
 * Device blocks behind these calls are cached per process (dt_host_trim frees the idle ones); dt_host_alloc /
 * dt_host_free give page-locked host memory for rasters that should cross PCIe at the full rate. */
int dt_host_trim(void);
int dt_host_alloc(int64_t bytes, void **out);
int dt_host_free(void *p);
/* dst[i] = (double)src[i] on the host with a few threads (float64 containers of float32 rasters, as the reference
 * returns them). */
int dt_host_f32_to_f64(const float *src, double *dst, int64_t n);


/* slope.slope_cpu + slope_gpu (slope.py:152-259): steepest-descent slope in percent.  The
 * -100 ring the reference pads on (slope.py:175-182) is implicit: neighbours outside the raster
 * are skipped like nodata neighbours. */
int dt_slope_f32(const float *dem, int64_t H, int64_t W, double px, float *slope);

/* Net-new N1 (no reference function; encoding pinned by flowhand.py:801-824): D8 code of the
 * neighbour that sets slope_gpu's maximum, first in its scan order.  `slope` may be NULL. */
int dt_d8_f32(const float *dem, int64_t H, int64_t W, double px, uint8_t *fdr, float *slope);

/* Net-new: D8 on a hydrologically conditioned surface, for DEMs with pits and flats (the reference
 * reads such an `fdr` from a GIS tool, Example/example.py:36).  Depressions are filled (priority-flood surface:
 * outlets = raster edge and cells next to nodata), D8 is taken on the filled surface, and the cells left without a
 * lower neighbour are routed over their flat to the nearest cell that has a code (hop distance through cells of
 * the same filled height; among the neighbours one hop closer the first of N,W,E,S, else of NW,NE,SW,SE); a
 * code-less cell next to nodata drains into its first nodata neighbour.  Every valid cell gets a code and the codes
 * contain no cycle.  filled (may be NULL) receives the filled surface; info3 (may be NULL) = {cells left without a
 * code (0), fill rounds, flat rounds}. */
int dt_d8_conditioned_f32(const float *dem, int64_t H, int64_t W, double px, uint8_t *fdr, float *filled,
                          int32_t *info3);

/* Net-new N2: flow accumulation = number of upstream cells excluding self; `dem` may be NULL,
 * otherwise cells with dem <= -100 are set to -100.  Cells on a D8 cycle get -100. */
int dt_flowacc_u8(const uint8_t *fdr, const float *dem, int64_t H, int64_t W, int64_t *acc);

/* flowhand.flow_distance_index_cpu + flow_distance_index_gpu (flowhand.py:476-846, untiled
 * call: out = 0, row_start = col_start = 0, matrix_columns = W) and flowhand.hand_calculator
 * (flowhand.py:414-442).  `dem`/`hand` may both be NULL to skip HAND. */
int dt_flowhand(const float *dem, const uint8_t *fdr, const int8_t *river, int64_t H, int64_t W,
                double px, float *fdist, int64_t *idx, float *hand);

/* flowhand.hand_calculator alone (flowhand.py:414-442): dem - dem[idx], negatives -> 0. */
int dt_hand_f32(const float *dem, const int64_t *idx, int64_t N, float *hand);

/* topoindexes.topographic_index_cpu + both kernels (topoindexes.py:170-295); slope in radians. */
int dt_twi(const int64_t *fac, const float *slope_rad, int64_t N, double px, double n_top,
           float *ti, float *mti);

/* gfi.river_accumulation (gfi.py:119-147): out[i] = fac[idx[i]] where idx != -100 else fac[0]. */
int dt_river_accumulation(const int64_t *fac, const int64_t *idx, int64_t N, int64_t *out);

/* gfi.geomorphic_flood_index_cpu/_gpu (gfi.py:210-294; zero_guard = 0) and gfi.ln_hl_H_cpu/_gpu
 * (gfi.py:349-440; zero_guard = 1: area == 0 -> 1) on an explicit per-cell area raster. */
int dt_gfi_area(const float *hand, const int64_t *area, int64_t N, double n_gfi, double scale_factor,
                double size, int zero_guard, float *out);

/* gfi.river_accumulation + geomorphic_flood_index_cpu/_gpu (gfi.py:119-147, 210-294). */
int dt_gfi(const float *hand, const int64_t *fac, const int64_t *idx, int64_t N, double n_gfi,
           double scale_factor, double size, float *gfi);

/* gfi.ln_hl_H_cpu/_gpu (gfi.py:349-440). */
int dt_lnhlh(const float *hand, const int64_t *fac, int64_t N, double n_gfi, double scale_factor,
             double size, float *out);

/* downslope.downslope_cpu + downslope_gpu + the -50 repair of downslope_sequential_jit
 * (downslope.py:379-532, 161-314), untiled.
 * raw != 0 reproduces downslope_cpu alone: failed walks are left as the marker -50. */
int dt_downslope(const float *dem, const uint8_t *fdr, int64_t H, int64_t W, double px,
                 double elevation_difference, int raw, float *out);

/* ---- heights in float64: a DEM (or HAND) that float32 cannot hold --------------------------------------------
 * The reference takes height differences in the raster's OWN dtype (slope.py:244-258 under Numba typing,
 * flowhand.py:436-438 `dem - dem[indices]`, downslope.py:468) and adds 0.01 to the HAND it is given in float64
 * (gfi.py:289-294, :429-440).  The float32 entry points above are that arithmetic exactly when every height is a
 * float32 value; for the rest -- a genuinely float64 DEM, integer heights beyond 2^24 (exact in float64 up to 2^53)
 * -- these take the heights as float64 and evaluate the literal expressions (one thread per cell on global memory:
 * the capability, not the tuned path).  descriptools_amd/_lib.py picks the tier per raster. */
/* slope.py:152-259 */
int dt_slope_f64(const double *dem, int64_t H, int64_t W, double px, float *slope);
/* flowhand.hand_calculator, flowhand.py:414-442 (flow distance / river index do not read heights: dt_flowhand with
 * dem = hand = NULL) */
int dt_hand_f64(const double *dem, const int64_t *idx, int64_t N, double *hand);
/* downslope.py:379-532 + the repair :161-314; raw as in dt_downslope */
int dt_downslope_f64(const double *dem, const uint8_t *fdr, int64_t H, int64_t W, double px,
                     double elevation_difference, int raw, float *out);
/* gfi.py:119-147 + :210-294 (own_area = 0: area of the river cell idx points at, no zero guard; idx required),
 * gfi.py:349-440 (own_area = 1: the cell's own accumulation, 0 -> 1; idx ignored) and geomorphic_flood_index_cpu on
 * an explicit per-cell area raster passed as `fac` (own_area = 2: no zero guard), on a float64 HAND */
int dt_gfi_f64h(const double *hand, const int64_t *fac, const int64_t *idx, int64_t N, double n_gfi,
                double scale_factor, double size, int own_area, float *out);

/* evaluation.binary_map + avaliacao for `nth` thresholds in one pass (evaluation.py:90-171):
 * counts4[t*4 + v] = #cells with binary(desc, th[t]) + remapped(flood) == v, v = 0..3.
 * Cells equal to `nodata_value` (the caller passes desc[0,0], evaluation.py:111) or NaN
 * classify 0; flood is remapped 1 -> 2, -100 -> 0 on the fly (evaluation.py:149-150). */
int dt_confusion_multi(const double *desc, const int8_t *flood, int64_t N, double nodata_value,
                       const double *th, int nth, int under, int64_t *counts4);

/* evaluation.minMaxScale (evaluation.py:5-9): out = NaN where x == nodata or x is NaN, else (x - mn) / (mx - mn),
 * in float32 for a float32 raster (is_f32) and float64 otherwise -- numpy's arithmetic for those dtypes. */
int dt_minmax_scale(const void *x, int is_f32, int64_t N, double mn, double mx, double nodata, void *out);
/* evaluation.binary_map (evaluation.py:90-123): binary = 1 where desc <= threshold ('under') or >= threshold,
 * 0 elsewhere, where desc is NaN and where desc == nodata_value (the caller passes desc[0, 0], :111). */
int dt_binary_map(const void *desc, int is_f32, int64_t N, double nodata_value, double threshold, int under,
                  uint8_t *binary);
/* evaluation.avaliacao (evaluation.py:126-171): flood is remapped IN PLACE (1 -> 2, -100 -> 0, :149-150),
 * klass (may be NULL) = binary + flood, counts4[v] = cells of class v = 0..3. */
int dt_avaliacao(const int32_t *binary, int8_t *flood, int64_t N, int32_t *klass, int64_t *counts4);

/* Synthetic "tilted integer fBm" DEM window, bit-identical to the oracle's. */
int dt_synth_dem(uint32_t seed, int64_t Hg, int64_t Wg, int64_t y0, int64_t x0, int64_t h,
                 int64_t w, int nodata_pct, float *out);

/* ---- device-pointer tier (resident chain; all asynchronous on ctx's stream) ------------- */
/* plain device memory for callers without their own allocator (numpy-only hosts) */
int dt_dev_malloc(dt_ctx *ctx, int64_t bytes, void **out);
int dt_dev_free(dt_ctx *ctx, void *p);
int dt_dev_h2d(dt_ctx *ctx, void *dst_dev, const void *src_host, int64_t bytes); /* synchronous */
int dt_dev_d2h(dt_ctx *ctx, void *dst_host, const void *src_dev, int64_t bytes); /* synchronous */
/* enqueue only: dt_ctx_sync before the host reads dst_host (page-locked memory: dt_host_alloc) */
int dt_dev_d2h_async(dt_ctx *ctx, void *dst_host, const void *src_dev, int64_t bytes);
int dt_dev_synth_dem(dt_ctx *ctx, uint32_t seed, int64_t Hg, int64_t Wg, int64_t y0, int64_t x0,
                     int64_t h, int64_t w, int nodata_pct, float *out);
/* slope (may be NULL), fdr (may be NULL), slope_rad (may be NULL): fused 3x3 stencil.
 * slope_rad = float32(atan(slope/100)), -100 where dem == -100 (Example/example.py:63-64). */
int dt_dev_slope_d8(dt_ctx *ctx, const float *dem, int64_t H, int64_t W, double px, float *slope,
                    uint8_t *fdr, float *slope_rad);
/* The north_star's fused "slope+TWI" stencil: one pass over dem (+ acc32) producing slope %
 * (may be NULL), slope in radians (may be NULL), TI and MTI -- the slope raster never has to be
 * re-read (topoindexes.py:234-295 on top of slope.py:210-259). */
int dt_dev_slope_twi(dt_ctx *ctx, const float *dem, const int32_t *acc32, int64_t H, int64_t W,
                     double px, double n_top, float *slope, float *slope_rad, float *ti, float *mti);
/* dt_d8_conditioned_f32 on device rasters; synchronous (the fixed-point iterations read a flag back). */
int dt_dev_condition_d8(dt_ctx *ctx, const float *dem, int64_t H, int64_t W, double px, float *filled, uint8_t *fdr,
                        int32_t *info3);
/* The same without any host synchronisation (the resident chain's form, chain.Chain(condition=True)): `rounds`
 * fill rounds and `rounds` flat rounds are enqueued (1..500), a round that follows a quiet one returns at once, and
 * DT_STATUS_NOT_CONVERGED is raised on the context (dt_ctx_status) when the budget did not reach the fixed point. */
int dt_dev_condition_d8_async(dt_ctx *ctx, const float *dem, int64_t H, int64_t W, double px, float *filled,
                              uint8_t *fdr, int rounds);
/* acc32: int32 accumulation (H*W < 2^31); dem may be NULL. */
int dt_dev_flowacc(dt_ctx *ctx, const uint8_t *fdr, const float *dem, int64_t H, int64_t W,
                   int32_t *acc32);
/* flow accumulation with the river mask (acc > threshold, Example/example.py:52) written by the
 * same final pass */
int dt_dev_flowacc_river(dt_ctx *ctx, const uint8_t *fdr, const float *dem, int64_t H, int64_t W,
                         int64_t threshold, int32_t *acc32, int8_t *river);
/* the same and the first phase of HAND (tile solve, perimeter node doubling) in one call: the single-raster form
 * of dt_dev_flowacc_finish_flowhand_local_w.  dt_dev_flowhand_finish_w / dt_dev_flowhand_gfi_finish_w with the whole
 * raster as the window (ld = W, halo 0) follow on the same context. */
int dt_dev_flowacc_river_flowhand_local(dt_ctx *ctx, const uint8_t *fdr, const float *dem, int64_t H, int64_t W,
                                        int64_t threshold, int32_t *acc32, int8_t *river);
/* The same with the nodata mask the D8 kernel can write on its way: dt_dev_slope_d8_m = dt_dev_slope_d8 (codes only)
 * that also fills `nodata4` -- one 16-bit word per 4 x 4 patch of cells, bit 4 j + k = cell (4 r + j, 4 i + k) holds the
 * sentinel (z <= -100); dt_nodata_mask_bytes(4, W) bytes per row of patches, dt_nodata_mask_bytes(H, W) in all -- and
 * dt_dev_flowacc_river_flowhand_local_m reads that mask instead of the DEM where it only needs "is this cell nodata"
 * (0.125 instead of 4 bytes per cell; the resident chain's form).  Heights that are NaN or +inf are outside the
 * contract of the mask (the D8 kernel treats them as nodata -- code 0, mask bit set --, `dem <= -100` does not). */
int64_t dt_nodata_mask_bytes(int64_t H, int64_t W);
int dt_dev_slope_d8_m(dt_ctx *ctx, const float *dem, int64_t H, int64_t W, double px, uint8_t *fdr, uint8_t *nodata4);
int dt_dev_flowacc_river_flowhand_local_m(dt_ctx *ctx, const uint8_t *fdr, const float *dem, const uint8_t *nodata4,
                                          int64_t H, int64_t W, int64_t threshold, int32_t *acc32, int8_t *river);
int dt_dev_river_mask(dt_ctx *ctx, const int32_t *acc32, int64_t N, int64_t threshold,
                      int8_t *river);
/* idx32: local flat index of the drained-to river cell (int32), -100 = none; a_river (may be
 * NULL, needs acc32) = acc32[idx] carried as payload (removes gfi.river_accumulation's gather;
 * -100 where there is no river cell, i.e. where hand is -100 and GFI is -100 whatever the area) */
int dt_dev_flowhand(dt_ctx *ctx, const float *dem, const uint8_t *fdr, const int8_t *river,
                    const int32_t *acc32, int64_t H, int64_t W, double px, float *fdist,
                    int32_t *idx32, float *hand, int32_t *a_river);
/* the same plus GFI and ln(hl/H) (gfi.py:268-294, :404-440; size = px as in example.py:81-91) evaluated in the
 * last tile pass from the HAND / river accumulation it holds in registers: one pass over the rasters less than
 * dt_dev_flowhand + dt_dev_gfi_lnhlh.  a_river may be NULL. */
int dt_dev_flowhand_gfi(dt_ctx *ctx, const float *dem, const uint8_t *fdr, const int8_t *river,
                        const int32_t *acc32, int64_t H, int64_t W, double px, double n_gfi, double b,
                        float *fdist, int32_t *idx32, float *hand, int32_t *a_river, float *gfi, float *lnhlh);
int dt_dev_twi(dt_ctx *ctx, const int32_t *acc32, const float *slope_rad, int64_t N, double px,
               double n_top, float *ti, float *mti);
/* a_river[i] = fac[idx[i]] (or anything where hand <= -100) */
int dt_dev_gfi(dt_ctx *ctx, const float *hand, const int32_t *a_river, int64_t N, double n_gfi,
               double scale_factor, double size, float *gfi);
int dt_dev_lnhlh(dt_ctx *ctx, const float *hand, const int32_t *acc32, int64_t N, double n_gfi,
                 double scale_factor, double size, float *out);
/* GFI and ln(hl/H) fused: one read of hand, ln(hand + 0.01) evaluated once */
int dt_dev_gfi_lnhlh(dt_ctx *ctx, const float *hand, const int32_t *a_river, const int32_t *acc32,
                     int64_t N, double n_gfi, double scale_factor, double size, float *gfi, float *lnhlh);
int dt_dev_downslope(dt_ctx *ctx, const float *dem, const uint8_t *fdr, int64_t H, int64_t W,
                     double px, double elevation_difference, int raw, float *out);
/* The same with the long-walk acceleration: walks that leave the kernel's window and are still short of the elevation
 * difference after 32 further moves are queued and finished with skip tables (8 moves per skip for every cell, 16 / 32 /
 * 64 for the queued cells, built on the device when at least 256 walks were queued) -- on real, conditioned terrain,
 * where flats and valley floors make walks thousands of moves long, an order of magnitude faster; same results.
 * `work`: dt_downslope_lift_workspace(H, W) bytes of device memory (33 bytes per cell), the library's for the duration
 * of the call's kernels. */
int64_t dt_downslope_lift_workspace(int64_t H, int64_t W);
int dt_dev_downslope_lift(dt_ctx *ctx, const float *dem, const uint8_t *fdr, int64_t H, int64_t W, double px,
                          double dz, int raw, float *out, void *work, int64_t work_bytes);
/* The same in two steps, for callers that may synchronise in between and want the tables' 25 bytes per cell only for
 * rasters that need them: dt_dev_downslope_queue runs the window kernel and queues the long walks (qwork:
 * dt_downslope_queue_workspace bytes, 8 per cell); dt_dev_downslope_queued waits for it and returns their number;
 * dt_dev_downslope_finish finishes them -- with skip tables when twork (dt_downslope_tables_workspace bytes) is given
 * and at least dt_downslope_tables_threshold walks are queued, move by move otherwise (twork may be NULL). */
int64_t dt_downslope_queue_workspace(int64_t H, int64_t W);
int64_t dt_downslope_tables_workspace(int64_t H, int64_t W);
int64_t dt_downslope_tables_threshold(int64_t H, int64_t W);
int dt_dev_downslope_queue(dt_ctx *ctx, const float *dem, const uint8_t *fdr, int64_t H, int64_t W, double px,
                           double dz, int raw, float *out, void *qwork, int64_t qbytes);
int dt_dev_downslope_queued(dt_ctx *ctx, const void *qwork, int64_t *count);
int dt_dev_downslope_finish(dt_ctx *ctx, const float *dem, const uint8_t *fdr, int64_t H, int64_t W, double px,
                            double dz, int raw, float *out, void *qwork, int64_t qbytes, void *twork, int64_t tbytes);
/* counts4_dev: device int64[nth*4], zeroed by the call */
int dt_dev_confusion_multi(dt_ctx *ctx, const double *desc, const int8_t *flood, int64_t N,
                           double nodata_value, const double *th_host, int nth, int under,
                           int64_t *counts4_dev);
/* ---- windowed device tier: the same kernels on one rank's window of a larger raster (multi-GPU).
 * Flow accumulation and HAND are split in two phases so the ranks can exchange one summary row per
 * cell of their core ring in between (descriptools_amd/tiling.py).  Ring order: top row, bottom row,
 * left column, right column (dt_perim_cells(H, W) rows). ------------------------------------------ */
int64_t dt_perim_cells(int64_t H, int64_t W);
/* hydrological conditioning on one rank's window (SURVEY.md 8f-4 tiled over ranks): the fixed points of
 * dt_dev_condition_d8 are iterated per rank, with a halo exchange of the filled surface / the flat distances and an
 * all-reduce of the "changed" flag in between (descriptools_amd/tiling.py: condition_ranks).  stage 0: init of the
 * surface (outlets: edge of the GLOBAL raster, cells next to nodata); 1: `rounds` fill rounds over the core, reading
 * the halo; 2: init of the flat distances (after D8 on the surface, dt_dev_slope_d8_w); 3: `rounds` flat rounds;
 * 4: the flat cells' codes.  *flag_dev (device int32, zeroed by the caller): raised by stages 1 / 3 when a cell changed,
 * number of cells left without a code after stage 4.  dist: uint32 raster laid out like the others. */
int dt_dev_condition_stage_w(dt_ctx *ctx, const dt_window *win, int stage, int rounds, const float *dem, float *filled,
                             uint8_t *fdr, uint32_t *dist, int32_t *flag_dev);
/* The same with a byte raster `nsame` (laid out like the others; the library's between stage 2 and stage 4): stage 2
 * leaves one byte per cell there -- which neighbours have another filled height -- and stages 3 / 4 work from it instead
 * of from the surface (less traffic and LDS per tile visit; same results). */
int dt_dev_condition_stage_m_w(dt_ctx *ctx, const dt_window *win, int stage, int rounds, const float *dem, float *filled,
                               uint8_t *fdr, uint32_t *dist, int32_t *flag_dev, uint8_t *nsame);
int dt_dev_slope_d8_w(dt_ctx *ctx, const dt_window *win, const float *dem, double px, float *slope,
                      uint8_t *fdr, float *slope_rad);
int dt_dev_slope_twi_w(dt_ctx *ctx, const dt_window *win, const float *dem, const int32_t *acc32, double px,
                       double n_top, float *slope, float *slope_rad, float *ti, float *mti);
/* n_unresolved_dev (device int32, may be NULL): walks that left this rank's halo (marked -50) */
int dt_dev_downslope_w(dt_ctx *ctx, const dt_window *win, const float *dem, const uint8_t *fdr, double px,
                       double elevation_difference, int raw, float *out, int32_t *n_unresolved_dev);
/* The same with the long-walk workspace (dt_dev_downslope_lift, for a rank): `work` = dt_downslope_lift_workspace_w(win)
 * bytes (8 per core cell for the queue, 24 per cell of the rank's memory -- core + halo -- for the skip tables).  On
 * real terrain (flats, valley floors) the walks of thousands of moves that stay in the rank's memory are finished in
 * skips of 64 moves instead of one dependent load per move (four ranks of the Example tiled 4 x 4: 69 -> see DESIGN.md
 * ms); the ones that leave it are marked and counted exactly as by dt_dev_downslope_w.  Same results bit for bit. */
int64_t dt_downslope_lift_workspace_w(const dt_window *win);
int dt_dev_downslope_lift_w(dt_ctx *ctx, const dt_window *win, const float *dem, const uint8_t *fdr, double px,
                            double elevation_difference, int raw, float *out, int32_t *n_unresolved_dev, void *work,
                            int64_t work_bytes);
/* Downslope walks that leave a rank's memory (real terrain: walks of thousands of moves along valley floors cross rank
 * borders).  dt_dev_downslope_emit_w = dt_dev_downslope_w (work NULL) / dt_dev_downslope_lift_w (work = the long-walk
 * workspace) that additionally EMITS every such walk, where it leaves, as a 48-byte walker record into `walkers`
 * (bytes 0-3: number of walks emitted, possibly more than fit; records from byte 256):
 *   words 0-3  start cell (global row, column), cell the walk stands on (global row, column)
 *   words 4-7  moves made, diagonal moves, float bits of the start height, flags (1 = carries the reference's
 *              sequential float64 path length instead of counts, 2 = finished)
 *   words 8-11 that float64 sum (two words), float bits of the result (finished walkers), 0
 * dt_dev_downslope_walk_w advances, in place, the n records that stand in this rank's memory until they finish or
 * reach the end of it again (the host sends them on: descriptools_amd/tiling.finish_downslope uses an all-to-all of
 * device buffers; the reference's analogue is the CPU repair downslope.py:373-374).  dt_dev_downslope_walk_seed_w
 * writes records of walkers at their start cells (core coordinates) for cells that are marked -50 without one. */
int dt_dev_downslope_emit_w(dt_ctx *ctx, const dt_window *win, const float *dem, const uint8_t *fdr, double px,
                            double elevation_difference, int raw, float *out, int32_t *n_unresolved_dev, void *work,
                            int64_t work_bytes, void *walkers, int64_t walkers_bytes);
int dt_dev_downslope_walk_w(dt_ctx *ctx, const dt_window *win, const float *dem, const uint8_t *fdr, double px,
                            double elevation_difference, int64_t n, void *records, void *work, int64_t work_bytes);
/* One iteration of the walkers' journey prepared on the device: records that arrived finished (flag 2) are home -- their
 * value is written into `out` (this rank's downslope raster, core origin) and they are marked (flag 4) -- the others
 * advance like dt_dev_downslope_walk_w; every record still wanted somewhere is then copied into `send` grouped by
 * destination rank (a walker that has just finished: the owner of its start cell; otherwise the owner of the cell it
 * stands on), counts[d] = records for rank d and counts[ty * tx] = how many of them are still on their way.  The host
 * reads `counts` (its one synchronisation), exchanges them and the groups (all-to-all), and calls again with what it
 * received until no rank sends anything.  row_starts[ty + 1] / col_starts[tx + 1]: first global row / column of every
 * rank row / column and the raster's end (device arrays; rank = rank row * tx + rank column, at most 1024 ranks);
 * counts: int32[ty * tx + 1]; scratch: int32[n + ty * tx]; send: room for n records. */
int dt_dev_downslope_walk_route_w(dt_ctx *ctx, const dt_window *win, const float *dem, const uint8_t *fdr, double px,
                                  double elevation_difference, int64_t n, void *records, void *work, int64_t work_bytes,
                                  float *out, const int32_t *row_starts, int32_t ty, const int32_t *col_starts,
                                  int32_t tx, void *send, int32_t *counts, int32_t *scratch);
int dt_dev_downslope_walk_seed_w(dt_ctx *ctx, const dt_window *win, const float *dem, int64_t n, const int32_t *ys,
                                 const int32_t *xs, void *records);
/* phase 1: in-rank accumulation; per ring cell: A = cells of this rank draining OUT through it (0 unless
 * its D8 step leaves the core), code = that step's D8 code, xr = ring index of the rank exit reached by
 * a path ENTERING at this cell (-1 none, -2 cycle inside the rank).  acc32 is not touched by this phase (the raster
 * is written by dt_dev_flowacc_finish_w / _w_a64) and may be NULL. */
int dt_dev_flowacc_local_w(dt_ctx *ctx, const dt_window *win, const uint8_t *fdr, int32_t *acc32,
                           int64_t *A_perim, int32_t *xr_perim, uint8_t *code_perim);
/* phase 2: ext_perim[i] = inflow arriving at ring cell i from other ranks (bit 63: fed by a D8 cycle
 * spanning ranks); NULL = none.  Must directly follow phase 1 on the same context. */
int dt_dev_flowacc_finish_w(dt_ctx *ctx, const dt_window *win, const uint8_t *fdr, const float *dem,
                            const uint64_t *ext_perim, int64_t threshold, int32_t *acc32, int8_t *river);
/* phase 2 of flow accumulation and phase 1 of HAND in one call: the last accumulation tile pass and HAND's first
 * stage the same 64 x 64 tiles of direction codes, and HAND's river mask is what the accumulation pass has just
 * computed, so in the common form (int32 accumulation, core width a multiple of 64, 16-byte aligned rasters) they are
 * ONE kernel; otherwise the separate kernels run back to back.  Results and the state left in the context are those
 * of dt_dev_flowacc_finish_w followed by dt_dev_flowhand_local_w (whose summary outputs kind ... ar these are). */
int dt_dev_flowacc_finish_flowhand_local_w(dt_ctx *ctx, const dt_window *win, const uint8_t *fdr, const float *dem,
                                           const uint64_t *ext_perim, int64_t threshold, int32_t *acc32, int8_t *river,
                                           uint8_t *kind, int32_t *ref, int32_t *nc, int32_t *nd, float *zr,
                                           int64_t *ar);
/* phase 1: per ring cell, the path ENTERING the rank there: kind 1 = ends on river cell `ref` (core-local
 * flat index; zr / ar = its height / accumulation), 2 = dead, 4 = leaves the rank again through ring
 * cell `ref`; nc / nd = cardinal / diagonal moves (kind 4: including the step out of the rank).  Everything that
 * crosses ranks carries accumulations as int64 (ar here, rem_ar below), whatever the rasters' width. */
int dt_dev_flowhand_local_w(dt_ctx *ctx, const dt_window *win, const float *dem, const uint8_t *fdr,
                            const int8_t *river, const int32_t *acc32, uint8_t *kind, int32_t *ref,
                            int32_t *nc, int32_t *nd, float *zr, int64_t *ar);
/* phase 2: for every ring cell whose step leaves the rank: res_ok != 0 -> that path ends on a river
 * cell after res_nc / res_nd further moves, global flat index rem_gidx, height rem_zr, accumulation
 * rem_ar (all NULL = no other ranks).  idx64 (may be NULL) receives GLOBAL flat indices; so does idx32 (may be NULL)
 * whenever the rank tables or idx64 are given -- meaningful while the global raster has <= 2^31 cells, and half
 * the bytes -- and the core-local flat index otherwise (a single raster: the same thing). */
int dt_dev_flowhand_finish_w(dt_ctx *ctx, const dt_window *win, const float *dem, const uint8_t *fdr,
                             const int8_t *river, const int32_t *acc32, double px, const uint8_t *res_ok,
                             const int32_t *res_nc, const int32_t *res_nd, const int64_t *rem_gidx,
                             const float *rem_zr, const int64_t *rem_ar, float *fdist, int32_t *idx32,
                             int64_t *idx64, float *hand, int32_t *a_river);
int dt_dev_flowhand_gfi_finish_w(dt_ctx *ctx, const dt_window *win, const float *dem, const uint8_t *fdr,
                                 const int8_t *river, const int32_t *acc32, double px, double n_gfi, double b,
                                 const uint8_t *res_ok, const int32_t *res_nc, const int32_t *res_nd,
                                 const int64_t *rem_gidx, const float *rem_zr, const int64_t *rem_ar,
                                 float *fdist, int32_t *idx32, int64_t *idx64, float *hand, int32_t *a_river,
                                 float *gfi, float *lnhlh);

/* ---- the same steps on int64 accumulation rasters (`_a64`).  The reference's flow accumulation is int64 end to
 * end (Example/example.py:39 reads it as int64; topoindexes.py:252-261, gfi.py:141-143 and :432-440 consume it).  On
 * the device a raster of up to 2^31 cells keeps it as int32 (half the bytes; exact, an accumulation being at most
 * cells - 1); a larger raster split
 * over ranks -- BASELINE.json configs[4], 65536^2 -- can hold basins beyond 32 bits, and its ranks run these entry
 * points instead: accumulation, river accumulation payload and every consumer (TI / MTI, HAND's river payload, GFI,
 * ln(hl/H)) in 64 bits; DT_STATUS_ACC_OVERFLOW is never raised.  dt_dev_flowacc_local_w, the rank-level solves and
 * dt_dev_downslope_w do not touch the accumulation raster and are shared. */
int dt_dev_flowacc_finish_w_a64(dt_ctx *ctx, const dt_window *win, const uint8_t *fdr, const float *dem,
                                const uint64_t *ext_perim, int64_t threshold, int64_t *acc64, int8_t *river);
int dt_dev_flowacc_finish_flowhand_local_w_a64(dt_ctx *ctx, const dt_window *win, const uint8_t *fdr, const float *dem,
                                               const uint64_t *ext_perim, int64_t threshold, int64_t *acc64,
                                               int8_t *river, uint8_t *kind, int32_t *ref, int32_t *nc, int32_t *nd,
                                               float *zr, int64_t *ar);
int dt_dev_slope_twi_w_a64(dt_ctx *ctx, const dt_window *win, const float *dem, const int64_t *acc64, double px,
                           double n_top, float *slope, float *slope_rad, float *ti, float *mti);
int dt_dev_flowhand_local_w_a64(dt_ctx *ctx, const dt_window *win, const float *dem, const uint8_t *fdr,
                                const int8_t *river, const int64_t *acc64, uint8_t *kind, int32_t *ref,
                                int32_t *nc, int32_t *nd, float *zr, int64_t *ar);
int dt_dev_flowhand_finish_w_a64(dt_ctx *ctx, const dt_window *win, const float *dem, const uint8_t *fdr,
                                 const int8_t *river, const int64_t *acc64, double px, const uint8_t *res_ok,
                                 const int32_t *res_nc, const int32_t *res_nd, const int64_t *rem_gidx,
                                 const float *rem_zr, const int64_t *rem_ar, float *fdist, int32_t *idx32,
                                 int64_t *idx64, float *hand, int64_t *a_river64);
int dt_dev_flowhand_gfi_finish_w_a64(dt_ctx *ctx, const dt_window *win, const float *dem, const uint8_t *fdr,
                                     const int8_t *river, const int64_t *acc64, double px, double n_gfi, double b,
                                     const uint8_t *res_ok, const int32_t *res_nc, const int32_t *res_nd,
                                     const int64_t *rem_gidx, const float *rem_zr, const int64_t *rem_ar,
                                     float *fdist, int32_t *idx32, int64_t *idx64, float *hand,
                                     int64_t *a_river64, float *gfi, float *lnhlh);
int dt_dev_gfi_lnhlh_a64(dt_ctx *ctx, const float *hand, const int64_t *a_river64, const int64_t *acc64, int64_t N,
                         double n_gfi, double scale_factor, double size, float *gfi, float *lnhlh);

/* evaluation on resident rasters.  out3_dev (device float[3]) = smallest,
 * second-smallest distinct and largest value of x, i.e. np.unique(x)[0], [1], [-1] as
 * Example/example.py:113-115 uses them (NaN when absent). */
int dt_dev_unique_extremes_f32(dt_ctx *ctx, const float *x, int64_t N, float *out3_dev);
/* evaluation.minMaxScale (evaluation.py:5-9) of a float32 raster: float32 arithmetic like numpy,
 * NaN where x == nodata, written as the float64 raster dt_dev_confusion_multi reads. */
int dt_dev_minmax_scale_f32(dt_ctx *ctx, const float *x, int64_t N, float mn, float mx, float nodata,
                            double *desc);

/* the same for an integer-valued raster kept as float32 on the device (the example's int16 HAND): float64
 * arithmetic, as numpy scales integer rasters */
int dt_dev_minmax_scale_f32_f64(dt_ctx *ctx, const float *x, int64_t N, double mn, double mx, double nodata,
                                double *desc);
/* binary_map + avaliacao at one threshold on resident rasters (Example/example.py:139-147): binary (may be NULL),
 * klass = binary + remapped flood (may be NULL), counts4_dev[v] = cells of class v; remap_flood != 0 rewrites the
 * benchmark map in place (1 -> 2, -100 -> 0) as avaliacao does. */
int dt_dev_classify(dt_ctx *ctx, const double *desc, int8_t *flood, int64_t N, double nodata_value, double threshold,
                    int under, int remap_flood, uint8_t *binary, int32_t *klass, int64_t *counts4_dev);

/* Device-to-device copy of N floats, the practical HBM ceiling the roofline fractions are put beside:
 * blocks > 0: float4 grid-stride copy with that many workgroups; blocks < 0: the buffer walked as rows of 16384
 * floats in 1024 x 4 patches, one per workgroup (N a multiple of 65536) -- the faster of the two forms on
 * MI355X (6.1 vs 5.4 TB/s), bench.py reports the better one */
int dt_dev_membench_copy(dt_ctx *ctx, const float *a, float *b, int64_t N, int blocks);
/* The same patches with n_reads (0-2) read streams summed into n_writes (1-3) write streams, optionally with
 * non-temporal loads / stores: the rate the memory system gives a read / write MIX with no arithmetic in the way.
 * The fused slope + TI + MTI stencil is 2 reads + 3 writes (60 % of its bytes are written). */
int dt_dev_membench_mix(dt_ctx *ctx, const float *r0, const float *r1, float *w0, float *w1, float *w2, int64_t N,
                        int n_reads, int n_writes, int nontemporal);
/* `reps` launches of dt_dev_membench_mix bracketed by HIP events on the context's stream (after one untimed launch):
 * *ms = mean duration of one launch.  Synchronises.  What descriptools_amd/placement.py labels blocks with. */
int dt_dev_membench_mix_timed(dt_ctx *ctx, const float *r0, const float *r1, float *w0, float *w1, float *w2, int64_t N,
                              int n_reads, int n_writes, int nontemporal, int reps, double *ms);
/* free / total bytes of the context's device (hipMemGetInfo) */
int dt_dev_mem_info(dt_ctx *ctx, int64_t *free_bytes, int64_t *total_bytes);

/* Rank-level solves on the GPU (multi-GPU): `rows_dev` holds one all-gathered byte row per rank
 * (rowbytes apart); field k of rank r starts at rows_dev + r * rowbytes + field_offsets[k] and has Pmax
 * entries.  heights / widths (host) describe the ty x tx rank grid.
 *   flow accumulation fields: {A int64, xr int32, code uint8} (the outputs of dt_dev_flowacc_local_w)
 *     -> ext_out_dev[P_rank] for dt_dev_flowacc_finish_w
 *   HAND fields: {ref int32, nc int32, nd int32, zr float, ar int64, kind uint8, ring D8 code uint8}
 *     -> the res_* / rem_* arrays [P_rank] for dt_dev_flowhand_finish_w */
int dt_dev_rank_solve_flowacc(dt_ctx *ctx, int ty, int tx, const int64_t *heights, const int64_t *widths,
                              int64_t Pmax, const void *rows_dev, int64_t rowbytes,
                              const int64_t *field_offsets3, int rank, int64_t P_rank, uint64_t *ext_out_dev);
int dt_dev_rank_solve_flowhand(dt_ctx *ctx, int ty, int tx, const int64_t *heights, const int64_t *widths,
                               int64_t Pmax, const void *rows_dev, int64_t rowbytes,
                               const int64_t *field_offsets7, int rank, int64_t P_rank, uint8_t *res_ok,
                               int32_t *res_nc, int32_t *res_nd, int64_t *rem_gidx, float *rem_zr,
                               int64_t *rem_ar);

/* widen / narrow helpers for the int64 API dtypes */
int dt_dev_i32_to_i64(dt_ctx *ctx, const int32_t *src, int64_t N, int64_t *dst);
int dt_dev_i64_to_i32(dt_ctx *ctx, const int64_t *src, int64_t N, int32_t *dst);

#ifdef __cplusplus
}
#endif
#endif /* DESCRIPTOOLS_HIP_H */
