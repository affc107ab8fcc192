// dt_kernels.h -- launchers implemented in dt_kernels.hip (all asynchronous on `s`).
#pragma once
#include "dt_common.h"

int dt_launch_synth_dem(hipStream_t s, uint32_t seed, int O, int64_t Hg, int64_t y0, int64_t x0,
                        int64_t h, int64_t w, int nodata_pct, float *out);
int dt_launch_stencil(hipStream_t s, const DtWin &w, const float *dem, double px, float *slope,
                      uint8_t *fdr, float *slope_rad, const void *acc, int acc64, double n_top, float *ti,
                      float *mti, void *aux = nullptr, uint8_t *nod4 = nullptr, int ldm = 0);
// the nodata mask of the D8-only kernel: one 16-bit word per 4 x 4 patch of cells (bit 4 j + k = cell (4 r + j, 4 i + k)
// is nodata), ldm words per row of patches
static inline int dt_nodata4_ld(int64_t W) { return (int)((((W + 3) / 4) + 7) & ~(int64_t)7); }
static inline size_t dt_nodata4_bytes(int64_t H, int64_t W) { return (size_t)((H + 3) / 4) * (size_t)dt_nodata4_ld(W) * 2; }
// workspace (tile marks + lane masks) of the fused slope + TI + MTI launch; see dt_stencil.hip
size_t dt_stencil_aux_bytes(int64_t H, int64_t W);
int dt_launch_flowacc(hipStream_t s, const uint8_t *fdr, const float *dem, int64_t H, int64_t W,
                      unsigned long long *state, int32_t *acc32);
int dt_launch_river_mask(hipStream_t s, const int32_t *acc32, int64_t n, int64_t thr, int8_t *river);
int dt_launch_flowhand(hipStream_t s, const float *dem, const uint8_t *fdr, const int8_t *river,
                       const int32_t *acc32, int64_t H, int64_t W, double px,
                       unsigned long long *state, float *fdist, int32_t *idx32, float *hand,
                       int32_t *a_river);
int dt_launch_twi(hipStream_t s, const int32_t *acc32, const float *srad, int64_t n, double px,
                  double n_top, float *ti, float *mti);
int dt_launch_twi_i64(hipStream_t s, const int64_t *fac, const float *srad, int64_t n, double px,
                      double n_top, float *ti, float *mti);
int dt_launch_gfi(hipStream_t s, const float *hand, const int32_t *area, int64_t n, double expo,
                  double b, double size, float *out, int own_cell);
int dt_launch_gfi_i64(hipStream_t s, const float *hand, const int64_t *area, int64_t n, double expo,
                      double b, double size, float *out, int own_cell);
int dt_launch_river_acc_i64(hipStream_t s, const int64_t *fac, const int64_t *idx, int64_t n,
                            int64_t *out);
size_t dt_downslope_lift_bytes(int64_t H, int64_t W);
size_t dt_downslope_queue_bytes(int64_t H, int64_t W);
size_t dt_downslope_tables_bytes(int64_t H, int64_t W);
uint32_t dt_downslope_lift_min(int64_t H, int64_t W);
size_t dt_downslope_tables_bytes_w(const DtWin &w);
size_t dt_downslope_lift_bytes_w(const DtWin &w);
int dt_launch_downslope(hipStream_t s, const DtWin &w, const float *dem, const uint8_t *fdr, double px,
                        double dz, int raw, float *out, int *n_unresolved, void *qwork = nullptr,
                        void *twork = nullptr, int phase = 0, void *walkers = nullptr, size_t walkers_bytes = 0);
int dt_launch_ds_walk(hipStream_t s, const DtWin &w, const float *dem, const uint8_t *fdr, double px, double dz,
                      int64_t n, void *rec, void *work, float *out = nullptr);
int dt_launch_ds_route(hipStream_t s, int64_t n, const void *rec, const int32_t *row_starts, int ty,
                       const int32_t *col_starts, int tx, void *send, int32_t *counts, int32_t *scratch);
int dt_launch_ds_walk_seed(hipStream_t s, const DtWin &w, const float *dem, int64_t n, const int32_t *ys,
                           const int32_t *xs, void *rec);
int dt_launch_downslope_v1(hipStream_t s, const float *dem, const uint8_t *fdr, int64_t H, int64_t W,
                           double px, double dz, int raw, float *out);
int dt_launch_hand_i64(hipStream_t s, const float *dem, const int64_t *idx, int64_t n, float *hand);
// dt_wide.hip: heights in float64
int dt_launch_slope_f64(hipStream_t s, const double *dem, int64_t H, int64_t W, double px, float *slope);
int dt_launch_hand_f64(hipStream_t s, const double *dem, const int64_t *idx, int64_t n, double *hand);
int dt_launch_downslope_f64(hipStream_t s, const double *dem, const uint8_t *fdr, int64_t H, int64_t W, double px,
                            double dz, int raw, float *out);
int dt_launch_gfi_f64h(hipStream_t s, const double *hand, const int64_t *fac, const int64_t *idx, int64_t n,
                       double expo, double b, double size, int own_area, float *out);
int dt_launch_confusion(hipStream_t s, const double *desc, const int8_t *flood, int64_t n,
                        double nodata, const double *th_host, int nth, int under,
                        unsigned long long *counts4);
int dt_launch_i32_to_i64(hipStream_t s, const int32_t *a, int64_t n, int64_t *b);
int dt_launch_i64_to_i32(hipStream_t s, const int64_t *a, int64_t n, int32_t *b);

// tile-hierarchical versions (dt_tiles.hip); two phases so that a multi-GPU run can exchange the
// rank-level summaries in between (single GPU: local + finish with no injection)
size_t dt_flowacc_tiled_scratch(int64_t H, int64_t W);
int dt_launch_fa_local(hipStream_t s, const DtWin &w, const uint8_t *fdr, void *scratch, size_t scratch_bytes,
                       int32_t *acc32, int rank_level);
int dt_launch_fa_summary(hipStream_t s, const DtWin &w, void *scratch, int64_t *A, int32_t *xr, uint8_t *code);
int dt_launch_fa_finish(hipStream_t s, const DtWin &w, const uint8_t *fdr, const float *dem, void *scratch,
                        const unsigned long long *ext_perim, int64_t river_thr, void *acc, int acc64,
                        int8_t *river, int *status = nullptr);
size_t dt_flowhand_tiled_scratch(int64_t H, int64_t W);
// nod4 (optional): the D8 kernel's nodata mask; when the fused kernel runs it replaces the read of `dem`
int dt_launch_fa_finish_fh_local(hipStream_t s, const DtWin &w, const uint8_t *fdr, const float *dem, void *fa_scratch,
                                 void *fh_scratch, size_t fh_bytes, const unsigned long long *ext_perim,
                                 int64_t river_thr, void *acc, int acc64, int8_t *river, int *status,
                                 const uint8_t *nod4 = nullptr, int ldm = 0);
int dt_launch_fh_local(hipStream_t s, const DtWin &w, const uint8_t *fdr, const int8_t *river, void *scratch,
                       size_t scratch_bytes);
int dt_launch_fh_summary(hipStream_t s, const DtWin &w, void *scratch, const float *dem, const void *acc, int acc64,
                         uint8_t *kind, int32_t *ref, int32_t *nc, int32_t *nd, float *zr, long long *ar);
int dt_launch_fh_finish(hipStream_t s, const DtWin &w, const float *dem, const uint8_t *fdr,
                        const int8_t *river, const void *acc, int acc64, double px, void *scratch,
                        const uint8_t *res_ok, const int32_t *res_nc, const int32_t *res_nd,
                        const long long *rem_gidx, const float *rem_zr, const long long *rem_ar, float *fdist,
                        int32_t *idx32, long long *idx64, float *hand, void *a_river, float *gfi = nullptr,
                        float *lnhlh = nullptr, double n_gfi = 0.0, double b_gfi = 1.0, double size = 1.0);
int dt_launch_gfi_both(hipStream_t s, const float *hand, const void *a_river, const void *fac, int acc64,
                       int64_t n, double expo, double b, double size, float *gfi, float *lnhlh);
int dt_launch_unique_extremes(hipStream_t s, const float *x, int64_t n, uint32_t *work4, float *out3);
int dt_launch_minmax_scale(hipStream_t s, const float *x, int64_t n, float mn, float mx, float nodata,
                           double *out);
int dt_launch_minmax_scale_f32f32(hipStream_t s, const float *x, int64_t n, float mn, float mx, float nodata,
                                  float *out);
int dt_launch_minmax_scale_f64(hipStream_t s, const double *x, int64_t n, double mn, double mx, double nodata,
                               double *out);
int dt_launch_minmax_scale_f32f64(hipStream_t s, const float *x, int64_t n, double mn, double mx, double nodata,
                                  double *out);
int dt_launch_classify_f64(hipStream_t s, const double *desc, const int32_t *bin_in, int8_t *flood, int64_t n,
                           double nodata, double th, int under, int remap, uint8_t *binary, int32_t *klass,
                           unsigned long long *counts4);
int dt_launch_classify_f32(hipStream_t s, const float *desc, const int32_t *bin_in, int8_t *flood, int64_t n,
                           float nodata, float th, int under, int remap, uint8_t *binary, int32_t *klass,
                           unsigned long long *counts4);
int dt_launch_membench_copy(hipStream_t s, const float *a, float *b, int64_t n, int blocks);
int dt_launch_membench_mix(hipStream_t s, const float *r0, const float *r1, float *w0, float *w1, float *w2, int64_t n,
                           int nr, int nw, int nt);
// rank-level solves on all-gathered summary rows (multi-GPU)
size_t dt_rank_solve_scratch(int nranks, int64_t Pmax);
int dt_launch_rank_solve_flowacc(hipStream_t s, int ty, int tx, const int64_t *heights, const int64_t *widths,
                                 int64_t Pmax, const void *rows, int64_t rowbytes, const int64_t *offs, int rank,
                                 int64_t P_rank, void *scratch, unsigned long long *ext_out);
int dt_launch_rank_solve_flowhand(hipStream_t s, int ty, int tx, const int64_t *heights, const int64_t *widths,
                                  int64_t Pmax, const void *rows, int64_t rowbytes, const int64_t *offs, int rank,
                                  int64_t P_rank, void *scratch, uint8_t *res_ok, int32_t *res_nc,
                                  int32_t *res_nd, long long *gidx, float *zr, long long *ar);

// hydrological conditioning (dt_hydro.hip): depression filling + flat resolution; synchronous
size_t dt_hydro_scratch(int64_t H, int64_t W);
int dt_launch_condition(hipStream_t s, const float *dem, int64_t H, int64_t W, double px, float *filled, uint8_t *fdr,
                        void *scratch, int *unresolved_host, int *rounds_host);
int dt_launch_condition_stage(hipStream_t s, const DtWin &w, int stage, int rounds, const float *dem, float *filled,
                              uint8_t *fdr, uint32_t *dist, int *flag_dev, uint8_t *nsame = nullptr);
int dt_launch_condition_async(hipStream_t s, const float *dem, int64_t H, int64_t W, double px, float *filled,
                              uint8_t *fdr, void *scratch, int rounds, int *status);

int dt_flow_impl();  // 1 global kernels, 2 tile-hierarchical (default)
