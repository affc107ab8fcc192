// dt_capi.hip -- extern "C" boundary of libdescriptools_hip.so (see include/descriptools_hip.h).
#include <stdarg.h>
#include <stdlib.h>

#include <sys/mman.h>
#include <atomic>
#include <chrono>
#include <mutex>
#include <system_error>
#include <thread>
#include <utility>
#include <vector>

#include "dt_common.h"
#include "dt_kernels.h"

// ---- errors -----------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void dt_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char *dt_last_error(void) { return g_err; }
extern "C" const char *dt_version(void) { return "descriptools_hip 0.1 (gfx950)"; }

static int g_debug[DT_DBG_COUNT] = {0};
int dt_debug_get(int key) { return (key >= 0 && key < DT_DBG_COUNT) ? g_debug[key] : 0; }
extern "C" int dt_debug_set(int key, int value) {
  DT_REQUIRE(key >= 0 && key < DT_DBG_COUNT, "unknown debug key");
  g_debug[key] = value;
  return DT_OK;
}

extern "C" int dt_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

// ---- context ----------------------------------------------------------------------------------
// live contexts (a graph remembers the context it was captured on and must know whether that still exists)
static std::mutex g_ctx_mu;
static std::vector<dt_ctx *> g_live_ctx;
static bool dt_ctx_alive(dt_ctx *c) {
  std::lock_guard<std::mutex> lk(g_ctx_mu);
  for (dt_ctx *p : g_live_ctx)
    if (p == c) return true;
  return false;
}

extern "C" int dt_ctx_create(int device, void *stream, dt_ctx **out) {
  DT_REQUIRE(out != nullptr, "out is NULL");
  int n = dt_device_count();
  if (n <= 0) {
    dt_set_error("no HIP device visible");
    return DT_ENODEV;
  }
  DT_REQUIRE(device >= 0 && device < n, "device index out of range");
  DT_HIP(hipSetDevice(device));
  dt_ctx *c = new dt_ctx();
  c->device = device;
  c->scratch = nullptr;
  c->scratch_bytes = 0;
  c->scratch_used = 0;
  c->scratch_owner = 0;
  c->owner_h = c->owner_w = 0;
  c->owner_ptr = c->owner_ptr2 = nullptr;
  c->scratch2 = nullptr;
  c->scratch2_bytes = 0;
  c->ev = nullptr;
  c->aux = nullptr;
  c->aux_bytes = 0;
  c->ws_gen = 0;
  c->status = nullptr;
  if (hipMalloc((void **)&c->status, 64) != hipSuccess || hipMemset(c->status, 0, 64) != hipSuccess) {
    dt_set_error("cannot allocate the context's status word");
    delete c;
    return DT_ENOMEM;
  }
  if (stream) {
    c->stream = (hipStream_t)stream;
    c->own_stream = false;
  } else {
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
      dt_set_error("hipStreamCreate failed: %s", hipGetErrorString(e));
      delete c;
      return DT_EHIP;
    }
    c->own_stream = true;
  }
  {
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    g_live_ctx.push_back(c);
  }
  *out = c;
  return DT_OK;
}

extern "C" int dt_ctx_destroy(dt_ctx *c) {
  if (!c) return DT_OK;
  {
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    for (size_t i = 0; i < g_live_ctx.size(); i++)
      if (g_live_ctx[i] == c) {
        g_live_ctx.erase(g_live_ctx.begin() + (long)i);
        break;
      }
  }
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  if (c->scratch) (void)hipFree(c->scratch);
  if (c->scratch2) (void)hipFree(c->scratch2);
  if (c->aux) (void)hipFree(c->aux);
  if (c->status) (void)hipFree(c->status);
  if (c->ev) (void)hipEventDestroy(c->ev);
  if (c->own_stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return DT_OK;
}

extern "C" int dt_ctx_set_stream(dt_ctx *c, void *stream) {
  DT_REQUIRE(c != nullptr, "ctx is NULL");
  DT_HIP(hipStreamSynchronize(c->stream));
  if (c->own_stream) DT_HIP(hipStreamDestroy(c->stream));
  if (stream) {
    c->stream = (hipStream_t)stream;
    c->own_stream = false;
  } else {
    DT_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    c->own_stream = true;
  }
  return DT_OK;
}
extern "C" void *dt_ctx_stream(dt_ctx *c) { return c ? (void *)c->stream : nullptr; }

// The context's own stream re-created with a scheduling priority: -1 high, 0 normal, +1 low (clamped to what the
// device offers).  A side branch of a pipeline on a LOW-priority stream fills the slots the main branch leaves
// idle instead of competing with it for every freed slot.
extern "C" int dt_ctx_set_priority(dt_ctx *c, int priority) {
  DT_REQUIRE(c != nullptr, "ctx is NULL");
  DT_REQUIRE(c->own_stream, "the context runs on a caller's stream: create that stream with the priority wanted");
  DT_HIP(hipSetDevice(c->device));
  int least = 0, greatest = 0;  // numerically: least priority = largest value
  DT_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
  int pr = priority < greatest ? greatest : (priority > least ? least : priority);
  hipStream_t ns = nullptr;
  DT_HIP(hipStreamCreateWithPriority(&ns, hipStreamNonBlocking, pr));
  DT_HIP(hipStreamSynchronize(c->stream));
  DT_HIP(hipStreamDestroy(c->stream));
  c->stream = ns;
  return DT_OK;
}

// `waiter`'s stream waits for everything enqueued so far on `signaller`'s stream (no host synchronisation)
static int dt_ctx_order(dt_ctx *signaller, dt_ctx *waiter) {
  DT_REQUIRE(signaller != nullptr && waiter != nullptr, "ctx is NULL");
  DT_REQUIRE(signaller->device == waiter->device, "contexts on different devices");
  if (signaller->stream == waiter->stream) return DT_OK;
  DT_HIP(hipSetDevice(signaller->device));
  if (!signaller->ev) DT_HIP(hipEventCreateWithFlags(&signaller->ev, hipEventDisableTiming));
  DT_HIP(hipEventRecord(signaller->ev, signaller->stream));
  DT_HIP(hipStreamWaitEvent(waiter->stream, signaller->ev, 0));
  return DT_OK;
}
extern "C" int dt_ctx_fork(dt_ctx *parent, dt_ctx *child) { return dt_ctx_order(parent, child); }
extern "C" int dt_ctx_join(dt_ctx *parent, dt_ctx *child) { return dt_ctx_order(child, parent); }
extern "C" int dt_ctx_sync(dt_ctx *c) {
  DT_REQUIRE(c != nullptr, "ctx is NULL");
  DT_HIP(hipStreamSynchronize(c->stream));
  return DT_OK;
}
// ---- HIP graphs: record what is enqueued on a context's stream once, replay it with one launch ----------------
// (the chain is ~45 launches per step: one host call per step instead; the step itself is no shorter, see
// chain.Chain.capture)
struct dt_graph {
  hipGraph_t graph;
  hipGraphExec_t exec;
  int device;
  dt_ctx *owner;       // the context it was captured on: the kernels' workspace pointers are that context's
  uint64_t owner_gen;  // its workspace generation at capture time
};
extern "C" int dt_ctx_capture_begin(dt_ctx *c) {
  DT_REQUIRE(c != nullptr, "ctx is NULL");
  DT_HIP(hipSetDevice(c->device));
  DT_HIP(hipStreamBeginCapture(c->stream, hipStreamCaptureModeRelaxed));
  return DT_OK;
}
extern "C" int dt_ctx_capture_end(dt_ctx *c, dt_graph **out) {
  DT_REQUIRE(c != nullptr && out != nullptr, "NULL argument");
  *out = nullptr;
  DT_HIP(hipSetDevice(c->device));
  hipGraph_t g = nullptr;
  DT_HIP(hipStreamEndCapture(c->stream, &g));
  DT_REQUIRE(g != nullptr, "nothing was captured");
  hipGraphExec_t e = nullptr;
  hipError_t err = hipGraphInstantiate(&e, g, nullptr, nullptr, 0);
  if (err != hipSuccess) {
    (void)hipGraphDestroy(g);
    DT_HIP(err);
  }
  *out = new dt_graph{g, e, c->device, c, c->ws_gen};
  return DT_OK;
}
extern "C" int dt_graph_launch(dt_graph *g, dt_ctx *c) {
  DT_REQUIRE(g != nullptr && c != nullptr, "NULL argument");
  DT_REQUIRE(g->device == c->device, "graph and context on different devices");
  // The captured kernels hold the raw addresses of the capturing context's grow-only workspaces.  A later call
  // that needed a larger one (a bigger raster, dt_dev_condition_d8, a rank-level solve) has freed the old block,
  // and so has destroying that context: replaying would write into freed memory.
  DT_REQUIRE(dt_ctx_alive(g->owner), "the context this graph was captured on has been destroyed: capture again");
  DT_REQUIRE(g->owner->ws_gen == g->owner_gen,
             "a workspace of the capturing context was reallocated after the capture (a call needed more scratch): "
             "the graph's pointers are stale, capture the step again");
  DT_HIP(hipSetDevice(c->device));
  DT_HIP(hipGraphLaunch(g->exec, c->stream));
  return DT_OK;
}
extern "C" int dt_graph_destroy(dt_graph *g) {
  if (!g) return DT_OK;
  (void)hipGraphExecDestroy(g->exec);
  (void)hipGraphDestroy(g->graph);
  delete g;
  return DT_OK;
}
// sticky status bits raised by kernels since the last call (synchronises the stream); bit 0: a flow accumulation
// value of a multi-rank raster may have reached 2^31 and does not fit the int32 accumulation rasters
extern "C" int dt_ctx_status(dt_ctx *c, int32_t *out) {
  DT_REQUIRE(c != nullptr && out != nullptr, "NULL argument");
  DT_HIP(hipSetDevice(c->device));
  DT_HIP(hipMemcpyAsync(out, c->status, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  DT_HIP(hipMemsetAsync(c->status, 0, sizeof(int32_t), c->stream));
  DT_HIP(hipStreamSynchronize(c->stream));
  return DT_OK;
}
extern "C" int64_t dt_ctx_scratch_bytes(dt_ctx *c) { return c ? (int64_t)c->scratch_bytes : 0; }

int dt_scratch_reset(dt_ctx *c, size_t total) {
  total = dt_align256(total) + 256;
  if (total > c->scratch_bytes) {
    // earlier work on the stream may still use the old block
    DT_HIP(hipStreamSynchronize(c->stream));
    if (c->scratch) DT_HIP(hipFree(c->scratch));
    c->scratch = nullptr;
    c->scratch_bytes = 0;
    c->ws_gen++;
    DT_HIP(hipMalloc((void **)&c->scratch, total));
    c->scratch_bytes = total;
  }
  c->scratch_used = 0;
  c->scratch_owner = 0;  // whatever two-phase state was here is about to be overwritten
  c->owner_ptr = c->owner_ptr2 = nullptr;
  return DT_OK;
}
void *dt_scratch_take(dt_ctx *c, size_t bytes) {
  size_t off = c->scratch_used;
  c->scratch_used += dt_align256(bytes);
  if (c->scratch_used > c->scratch_bytes) return nullptr;
  return c->scratch + off;
}

// grow-only side buffers (scratch2: rank-level solves; aux: stencil marks)
static int dt_side_reserve(dt_ctx *c, char **buf, size_t *have, size_t bytes) {
  if (bytes > *have) {
    DT_HIP(hipStreamSynchronize(c->stream));
    if (*buf) DT_HIP(hipFree(*buf));
    *buf = nullptr;
    *have = 0;
    c->ws_gen++;
    DT_HIP(hipMalloc((void **)buf, bytes));
    *have = bytes;
  }
  return DT_OK;
}

#define DT_CTX(c)                            \
  DT_REQUIRE((c) != nullptr, "ctx is NULL"); \
  DT_HIP(hipSetDevice((c)->device))

// DT_FLOW_IMPL=v1 selects the first-generation global kernels for flow accumulation / HAND
static int g_flow_impl = -1;
int dt_flow_impl() {
  if (g_flow_impl < 0) {
    const char *e = getenv("DT_FLOW_IMPL");
    g_flow_impl = (e && strcmp(e, "v1") == 0) ? 1 : 2;
  }
  return g_flow_impl;
}
extern "C" int dt_set_flow_impl(int impl) {
  DT_REQUIRE(impl == 1 || impl == 2, "impl must be 1 (global kernels) or 2 (tile-hierarchical)");
  g_flow_impl = impl;
  return DT_OK;
}

static DtWin dt_full_window(int64_t H, int64_t W) {
  DtWin w;
  w.H = (int)H; w.W = (int)W; w.ld = W; w.gy0 = 0; w.gx0 = 0; w.Hg = (int)H; w.Wg = (int)W; w.halo = 0;
  return w;
}
static int dt_convert_window(const dt_window *in, DtWin *out) {
  DT_REQUIRE(in != nullptr, "window is NULL");
  DT_REQUIRE(in->H >= 0 && in->W >= 0 && in->H * in->W < (1ll << 31), "bad core shape");
  DT_REQUIRE(in->ld >= in->W, "ld < W");
  DT_REQUIRE(in->Hg < (1ll << 31) && in->Wg < (1ll << 31) && in->gy0 >= 0 && in->gx0 >= 0 &&
                 in->gy0 + in->H <= in->Hg && in->gx0 + in->W <= in->Wg, "core window outside the global raster");
  DT_REQUIRE(in->halo >= 0, "negative halo");
  bool touches_all = in->gy0 == 0 && in->gx0 == 0 && in->gy0 + in->H == in->Hg && in->gx0 + in->W == in->Wg;
  DT_REQUIRE(touches_all || in->halo >= 1, "a window inside a larger raster needs a halo of >= 1 cell");
  out->H = (int)in->H; out->W = (int)in->W; out->ld = in->ld; out->gy0 = (int)in->gy0; out->gx0 = (int)in->gx0;
  out->Hg = (int)in->Hg; out->Wg = (int)in->Wg; out->halo = (int)in->halo;
  return DT_OK;
}

static int dt_check_hw(int64_t H, int64_t W) {
  DT_REQUIRE(H >= 0 && W >= 0, "negative raster shape");
  DT_REQUIRE(H * W < (1ll << 31), "rasters of >= 2^31 cells must be tiled (one tile per GPU)");
  return DT_OK;
}

// ---- device tier ------------------------------------------------------------------------------
extern "C" int dt_dev_malloc(dt_ctx *c, int64_t bytes, void **out) {
  DT_CTX(c);
  DT_REQUIRE(out && bytes >= 0, "bad arguments");
  *out = nullptr;
  const size_t n = bytes > 0 ? (size_t)bytes : 16;
  if (hipMalloc(out, n) != hipSuccess) {
    (void)hipGetLastError();
    *out = nullptr;
    dt_host_trim();  // the host tier's cached device blocks are the only memory this library holds on to
    const hipError_t e = hipMalloc(out, n);
    if (e != hipSuccess) {
      // the runtime keeps the last error until it is read: left in place, the hipGetLastError() behind the next
      // kernel launch of ANY entry point would report this out-of-memory (placement.assign uses a full device as
      // control flow)
      (void)hipGetLastError();
      *out = nullptr;
      dt_set_error("hipMalloc(%zu bytes) failed: %s", n, hipGetErrorString(e));
      return e == hipErrorOutOfMemory ? DT_ENOMEM : DT_EHIP;
    }
  }
  return DT_OK;
}
extern "C" int dt_dev_free(dt_ctx *c, void *p) {
  DT_CTX(c);
  if (!p) return DT_OK;
  DT_HIP(hipStreamSynchronize(c->stream));
  DT_HIP(hipFree(p));
  return DT_OK;
}
extern "C" int dt_dev_h2d(dt_ctx *c, void *dst, const void *src, int64_t bytes) {
  DT_CTX(c);
  if (bytes <= 0) return DT_OK;
  DT_REQUIRE(dst && src, "NULL pointer");
  DT_HIP(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyHostToDevice, c->stream));
  DT_HIP(hipStreamSynchronize(c->stream));
  return DT_OK;
}
extern "C" int dt_dev_d2h(dt_ctx *c, void *dst, const void *src, int64_t bytes) {
  DT_CTX(c);
  if (bytes <= 0) return DT_OK;
  DT_REQUIRE(dst && src, "NULL pointer");
  DT_HIP(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToHost, c->stream));
  DT_HIP(hipStreamSynchronize(c->stream));
  return DT_OK;
}

// enqueue only (dt_ctx_sync before the host reads dst; dst should be page-locked: dt_host_alloc)
extern "C" int dt_dev_d2h_async(dt_ctx *c, void *dst, const void *src, int64_t bytes) {
  DT_CTX(c);
  if (bytes <= 0) return DT_OK;
  DT_REQUIRE(dst && src, "NULL pointer");
  DT_HIP(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToHost, c->stream));
  return DT_OK;
}

extern "C" int dt_dev_slope_twi(dt_ctx *c, const float *dem, const int32_t *acc32, int64_t H, int64_t W,
                                double px, double n_top, float *slope, float *slope_rad, float *ti,
                                float *mti) {
  DT_CTX(c);
  DT_TRY(dt_check_hw(H, W));
  DT_REQUIRE((dem && acc32 && ti && mti) || H * W == 0, "NULL raster");
  DT_TRY(dt_side_reserve(c, &c->aux, &c->aux_bytes, dt_stencil_aux_bytes(H, W)));
  DT_TRY(dt_launch_stencil(c->stream, dt_full_window(H, W), dem, px, slope, nullptr, slope_rad, acc32, 0, n_top, ti,
                           mti, c->aux));
  DT_HIP(hipGetLastError());
  return DT_OK;
}

static int synth_octaves(int64_t Hg, int64_t Wg) {
  int64_t m = Hg < Wg ? Hg : Wg;
  int lg = 0;
  while ((m >> (lg + 1)) > 0) lg++;
  int O = lg - 2;
  if (O < 5) O = 5;
  if (O > 14) O = 14;
  return O;
}

extern "C" int dt_dev_synth_dem(dt_ctx *c, uint32_t seed, int64_t Hg, int64_t Wg, int64_t y0,
                                int64_t x0, int64_t h, int64_t w, int nodata_pct, float *out) {
  DT_CTX(c);
  DT_REQUIRE(out || h * w == 0, "out is NULL");
  DT_REQUIRE(Hg > 0 && Wg > 0 && h >= 0 && w >= 0, "bad shape");
  DT_TRY(dt_launch_synth_dem(c->stream, seed, synth_octaves(Hg, Wg), Hg, y0, x0, h, w, nodata_pct, out));
  DT_HIP(hipGetLastError());
  return DT_OK;
}

extern "C" int dt_dev_slope_d8(dt_ctx *c, const float *dem, int64_t H, int64_t W, double px,
                               float *slope, uint8_t *fdr, float *slope_rad) {
  DT_CTX(c);
  DT_TRY(dt_check_hw(H, W));
  DT_REQUIRE(dem || H * W == 0, "dem is NULL");
  DT_REQUIRE(slope || fdr || slope_rad, "no output requested");
  DT_TRY(dt_side_reserve(c, &c->aux, &c->aux_bytes, dt_stencil_aux_bytes(H, W)));
  DT_TRY(dt_launch_stencil(c->stream, dt_full_window(H, W), dem, px, slope, fdr, slope_rad, nullptr, 0, 0.0, nullptr,
                           nullptr, c->aux));
  DT_HIP(hipGetLastError());
  return DT_OK;
}

// Conditioned D8: fill depressions, D8 on the filled surface, resolve flats.  `filled` (device,
// H*W floats) receives the filled surface; info3 (host, may be NULL) = {flat cells left without a code (0), fill
// rounds, flat rounds}.  Synchronous: the fixed-point iterations read a flag back per batch of rounds.
extern "C" int dt_dev_condition_d8(dt_ctx *c, const float *dem, int64_t H, int64_t W, double px, float *filled,
                                   uint8_t *fdr, int32_t *info3) {
  DT_CTX(c);
  DT_TRY(dt_check_hw(H, W));
  DT_REQUIRE((dem && filled) || H * W == 0, "NULL raster");
  size_t need = dt_hydro_scratch(H, W);
  DT_TRY(dt_scratch_reset(c, need));
  void *scr = dt_scratch_take(c, need);
  int unresolved = 0, rounds[2] = {0, 0};
  DT_TRY(dt_launch_condition(c->stream, dem, H, W, px, filled, fdr, scr, &unresolved, rounds));
  DT_HIP(hipGetLastError());
  if (info3) {
    info3[0] = unresolved;
    info3[1] = rounds[0];
    info3[2] = rounds[1];
  }
  return DT_OK;
}

extern "C" int dt_dev_condition_d8_async(dt_ctx *c, const float *dem, int64_t H, int64_t W, double px, float *filled,
                                         uint8_t *fdr, int rounds) {
  DT_CTX(c);
  DT_TRY(dt_check_hw(H, W));
  DT_REQUIRE((dem && filled && fdr) || H * W == 0, "NULL raster");
  size_t need = dt_hydro_scratch(H, W);
  DT_TRY(dt_scratch_reset(c, need));
  void *scr = dt_scratch_take(c, need);
  DT_TRY(dt_launch_condition_async(c->stream, dem, H, W, px, filled, fdr, scr, rounds, c->status));
  DT_HIP(hipGetLastError());
  return DT_OK;
}

extern "C" int dt_dev_condition_stage_w(dt_ctx *c, const dt_window *win, int stage, int rounds, const float *dem,
                                        float *filled, uint8_t *fdr, uint32_t *dist, int32_t *flag_dev) {
  DT_CTX(c);
  DtWin w;
  DT_TRY(dt_convert_window(win, &w));
  DT_TRY(dt_launch_condition_stage(c->stream, w, stage, rounds, dem, filled, fdr, dist, (int *)flag_dev));
  DT_HIP(hipGetLastError());
  return DT_OK;
}

// ... with a byte raster `nsame` (laid out like the others, the library's between stage 2 and stage 4): the flat stages
// then work from one byte per cell instead of the surface -- less traffic and LDS per tile visit
extern "C" int dt_dev_condition_stage_m_w(dt_ctx *c, const dt_window *win, int stage, int rounds, const float *dem,
                                          float *filled, uint8_t *fdr, uint32_t *dist, int32_t *flag_dev,
                                          uint8_t *nsame) {
  DT_CTX(c);
  DtWin w;
  DT_TRY(dt_convert_window(win, &w));
  DT_REQUIRE(nsame != nullptr || stage < 2, "the byte raster is missing");
  DT_TRY(dt_launch_condition_stage(c->stream, w, stage, rounds, dem, filled, fdr, dist, (int *)flag_dev, nsame));
  DT_HIP(hipGetLastError());
  return DT_OK;
}

extern "C" int dt_dev_flowacc(dt_ctx *c, const uint8_t *fdr, const float *dem, int64_t H, int64_t W,
                              int32_t *acc32) {
  DT_CTX(c);
  DT_TRY(dt_check_hw(H, W));
  DT_REQUIRE((fdr && acc32) || H * W == 0, "NULL raster");
  if (dt_flow_impl() == 1) {  // v1: one global countdown (kept for A/B runs: DT_FLOW_IMPL=v1)
    DT_TRY(dt_scratch_reset(c, (size_t)H * W * 8));
    unsigned long long *state = (unsigned long long *)dt_scratch_take(c, (size_t)H * W * 8);
    DT_TRY(dt_launch_flowacc(c->stream, fdr, dem, H, W, state, acc32));
  } else {
    size_t need = dt_flowacc_tiled_scratch(H, W);
    DT_TRY(dt_scratch_reset(c, need));
    void *scr = dt_scratch_take(c, need);
    DtWin w = dt_full_window(H, W);
    DT_TRY(dt_launch_fa_local(c->stream, w, fdr, scr, need, acc32, 0));
    DT_TRY(dt_launch_fa_finish(c->stream, w, fdr, dem, scr, nullptr, 0, acc32, 0, nullptr));
  }
  DT_HIP(hipGetLastError());
  return DT_OK;
}

extern "C" int dt_dev_river_mask(dt_ctx *c, const int32_t *acc32, int64_t N, int64_t threshold,
                                 int8_t *river) {
  DT_CTX(c);
  DT_REQUIRE((acc32 && river) || N == 0, "NULL raster");
  DT_TRY(dt_launch_river_mask(c->stream, acc32, N, threshold, river));
  DT_HIP(hipGetLastError());
  return DT_OK;
}

extern "C" int dt_dev_flowhand(dt_ctx *c, const float *dem, const uint8_t *fdr, const int8_t *river,
                               const int32_t *acc32, int64_t H, int64_t W, double px, float *fdist,
                               int32_t *idx32, float *hand, int32_t *a_river) {
  DT_CTX(c);
  DT_TRY(dt_check_hw(H, W));
  DT_REQUIRE((fdr && river) || H * W == 0, "NULL raster");
  DT_REQUIRE(!hand || dem, "hand needs dem");
  DT_REQUIRE(!a_river || acc32, "a_river needs acc32");
  if (dt_flow_impl() == 1) {
    DT_TRY(dt_scratch_reset(c, (size_t)H * W * 8));
    unsigned long long *state = (unsigned long long *)dt_scratch_take(c, (size_t)H * W * 8);
    DT_TRY(dt_launch_flowhand(c->stream, dem, fdr, river, acc32, H, W, px, state, fdist, idx32, hand, a_river));
  } else {
    size_t need = dt_flowhand_tiled_scratch(H, W);
    DT_TRY(dt_scratch_reset(c, need));
    void *scr = dt_scratch_take(c, need);
    DtWin w = dt_full_window(H, W);
    DT_TRY(dt_launch_fh_local(c->stream, w, fdr, river, scr, need));
    DT_TRY(dt_launch_fh_finish(c->stream, w, dem, fdr, river, acc32, 0, px, scr, nullptr, nullptr, nullptr, nullptr,
                               nullptr, nullptr, fdist, idx32, nullptr, hand, a_river));
  }
  DT_HIP(hipGetLastError());
  return DT_OK;
}

// HAND + GFI + ln(hl/H) in one go: the last tile pass of dt_dev_flowhand also evaluates gfi.py:268-294 and
// :404-440 from the values it holds in registers.  a_river may be NULL (it is only an intermediate).
extern "C" int dt_dev_flowhand_gfi(dt_ctx *c, const float *dem, const uint8_t *fdr, const int8_t *river,
                                   const int32_t *acc32, int64_t H, int64_t W, double px, double n_gfi,
                                   double b, float *fdist, int32_t *idx32, float *hand, int32_t *a_river,
                                   float *gfi, float *lnhlh) {
  DT_CTX(c);
  DT_TRY(dt_check_hw(H, W));
  DT_REQUIRE((dem && fdr && river && acc32 && gfi && lnhlh) || H * W == 0, "NULL raster");
  size_t need = dt_flowhand_tiled_scratch(H, W);
  DT_TRY(dt_scratch_reset(c, need));
  void *scr = dt_scratch_take(c, need);
  DtWin w = dt_full_window(H, W);
  DT_TRY(dt_launch_fh_local(c->stream, w, fdr, river, scr, need));
  DT_TRY(dt_launch_fh_finish(c->stream, w, dem, fdr, river, acc32, 0, px, scr, nullptr, nullptr, nullptr, nullptr,
                             nullptr, nullptr, fdist, idx32, nullptr, hand, a_river, gfi, lnhlh, n_gfi, b, px));
  DT_HIP(hipGetLastError());
  return DT_OK;
}

extern "C" int dt_dev_twi(dt_ctx *c, const int32_t *acc32, const float *slope_rad, int64_t N, double px,
                          double n_top, float *ti, float *mti) {
  DT_CTX(c);
  DT_REQUIRE((acc32 && slope_rad && ti && mti) || N == 0, "NULL raster");
  DT_TRY(dt_launch_twi(c->stream, acc32, slope_rad, N, px, n_top, ti, mti));
  DT_HIP(hipGetLastError());
  return DT_OK;
}

extern "C" int dt_dev_gfi(dt_ctx *c, const float *hand, const int32_t *a_river, int64_t N, double n_gfi,
                          double b, double size, float *gfi) {
  DT_CTX(c);
  DT_REQUIRE((hand && a_river && gfi) || N == 0, "NULL raster");
  DT_TRY(dt_launch_gfi(c->stream, hand, a_river, N, n_gfi, b, size, gfi, 0));
  DT_HIP(hipGetLastError());
  return DT_OK;
}

extern "C" int dt_dev_lnhlh(dt_ctx *c, const float *hand, const int32_t *acc32, int64_t N, double n_gfi,
                            double b, double size, float *out) {
  DT_CTX(c);
  DT_REQUIRE((hand && acc32 && out) || N == 0, "NULL raster");
  DT_TRY(dt_launch_gfi(c->stream, hand, acc32, N, n_gfi, b, size, out, 1));
  DT_HIP(hipGetLastError());
  return DT_OK;
}

static int dev_gfi_lnhlh(dt_ctx *c, const float *hand, const void *a_river, const void *acc, int acc64, int64_t N,
                         double n_gfi, double b, double size, float *gfi, float *lnhlh) {
  DT_CTX(c);
  DT_REQUIRE((hand && a_river && acc && gfi && lnhlh) || N == 0, "NULL raster");
  DT_TRY(dt_launch_gfi_both(c->stream, hand, a_river, acc, acc64, N, n_gfi, b, size, gfi, lnhlh));
  DT_HIP(hipGetLastError());
  return DT_OK;
}
extern "C" int dt_dev_gfi_lnhlh(dt_ctx *c, const float *hand, const int32_t *a_river, const int32_t *acc32,
                                int64_t N, double n_gfi, double b, double size, float *gfi, float *lnhlh) {
  return dev_gfi_lnhlh(c, hand, a_river, acc32, 0, N, n_gfi, b, size, gfi, lnhlh);
}
extern "C" int dt_dev_gfi_lnhlh_a64(dt_ctx *c, const float *hand, const int64_t *a_river, const int64_t *acc64,
                                    int64_t N, double n_gfi, double b, double size, float *gfi, float *lnhlh) {
  return dev_gfi_lnhlh(c, hand, a_river, acc64, 1, N, n_gfi, b, size, gfi, lnhlh);
}

extern "C" int dt_dev_flowacc_river(dt_ctx *c, const uint8_t *fdr, const float *dem, int64_t H, int64_t W,
                                    int64_t threshold, int32_t *acc32, int8_t *river) {
  DT_CTX(c);
  DT_TRY(dt_check_hw(H, W));
  DT_REQUIRE((fdr && acc32 && river) || H * W == 0, "NULL raster");
  if (dt_flow_impl() == 1) {
    DT_TRY(dt_dev_flowacc(c, fdr, dem, H, W, acc32));
    return dt_dev_river_mask(c, acc32, H * W, threshold, river);
  }
  size_t need = dt_flowacc_tiled_scratch(H, W);
  DT_TRY(dt_scratch_reset(c, need));
  void *scr = dt_scratch_take(c, need);
  DtWin w = dt_full_window(H, W);
  DT_TRY(dt_launch_fa_local(c->stream, w, fdr, scr, need, acc32, 0));
  DT_TRY(dt_launch_fa_finish(c->stream, w, fdr, dem, scr, nullptr, threshold, acc32, 0, river));
  DT_HIP(hipGetLastError());
  return DT_OK;
}

extern "C" int dt_dev_downslope(dt_ctx *c, const float *dem, const uint8_t *fdr, int64_t H, int64_t W,
                                double px, double dz, int raw, float *out) {
  DT_CTX(c);
  DT_TRY(dt_check_hw(H, W));
  DT_REQUIRE((dem && fdr && out) || H * W == 0, "NULL raster");
  if (dt_flow_impl() == 1) {  // v1: one thread per cell walking global memory (kept for A/B and verification runs)
    DT_TRY(dt_launch_downslope_v1(c->stream, dem, fdr, H, W, px, dz, raw, out));
  } else {
    DT_TRY(dt_launch_downslope(c->stream, dt_full_window(H, W), dem, fdr, px, dz, raw, out, nullptr));
  }
  DT_HIP(hipGetLastError());
  return DT_OK;
}

extern "C" int64_t dt_downslope_lift_workspace(int64_t H, int64_t W) {
  return (H <= 0 || W <= 0) ? 0 : (int64_t)dt_downslope_lift_bytes(H, W);
}
extern "C" int64_t dt_downslope_queue_workspace(int64_t H, int64_t W) {
  return (H <= 0 || W <= 0) ? 0 : (int64_t)dt_downslope_queue_bytes(H, W);
}
extern "C" int64_t dt_downslope_tables_workspace(int64_t H, int64_t W) {
  return (H <= 0 || W <= 0) ? 0 : (int64_t)dt_downslope_tables_bytes(H, W);
}
extern "C" int64_t dt_downslope_tables_threshold(int64_t H, int64_t W) {
  return (H <= 0 || W <= 0) ? 0 : (int64_t)dt_downslope_lift_min(H, W);
}
// dt_dev_downslope with the long-walk acceleration (dt_kernels.hip, DsQueue): `work` = dt_downslope_lift_workspace
// bytes of device memory, the caller's for the duration of the call's kernels
extern "C" int dt_dev_downslope_lift(dt_ctx *c, const float *dem, const uint8_t *fdr, int64_t H, int64_t W, double px,
                                     double dz, int raw, float *out, void *work, int64_t work_bytes) {
  DT_CTX(c);
  DT_TRY(dt_check_hw(H, W));
  DT_REQUIRE((dem && fdr && out) || H * W == 0, "NULL raster");
  DT_REQUIRE(work != nullptr && work_bytes >= dt_downslope_lift_workspace(H, W), "downslope workspace missing or too small");
  DT_TRY(dt_launch_downslope(c->stream, dt_full_window(H, W), dem, fdr, px, dz, raw, out, nullptr, work,
                             (char *)work + dt_downslope_queue_bytes(H, W), 0));
  DT_HIP(hipGetLastError());
  return DT_OK;
}
// The same in two steps, for callers that may synchronise in between and want the 48 bytes per cell of the tables only
// for rasters that need them: dt_dev_downslope_queue runs the window kernel and queues the long walks (qwork:
// dt_downslope_queue_workspace bytes), dt_dev_downslope_queued waits and says how many there are,
// dt_dev_downslope_finish finishes them -- with skip tables when twork (dt_downslope_tables_workspace bytes) is given
// and at least dt_downslope_tables_threshold walks are queued, move by move otherwise.
extern "C" int dt_dev_downslope_queue(dt_ctx *c, const float *dem, const uint8_t *fdr, int64_t H, int64_t W, double px,
                                      double dz, int raw, float *out, void *qwork, int64_t qbytes) {
  DT_CTX(c);
  DT_TRY(dt_check_hw(H, W));
  DT_REQUIRE((dem && fdr && out) || H * W == 0, "NULL raster");
  DT_REQUIRE(qwork != nullptr && qbytes >= dt_downslope_queue_workspace(H, W), "queue workspace missing or too small");
  DT_TRY(dt_launch_downslope(c->stream, dt_full_window(H, W), dem, fdr, px, dz, raw, out, nullptr, qwork, nullptr, 1));
  DT_HIP(hipGetLastError());
  return DT_OK;
}
extern "C" int dt_dev_downslope_queued(dt_ctx *c, const void *qwork, int64_t *count) {
  DT_CTX(c);
  DT_REQUIRE(qwork && count, "NULL pointer");
  uint32_t n = 0;
  DT_HIP(hipMemcpyAsync(&n, qwork, sizeof(n), hipMemcpyDeviceToHost, c->stream));
  DT_HIP(hipStreamSynchronize(c->stream));
  *count = (int64_t)n;
  return DT_OK;
}
extern "C" int dt_dev_downslope_finish(dt_ctx *c, const float *dem, const uint8_t *fdr, int64_t H, int64_t W, double px,
                                       double dz, int raw, float *out, void *qwork, int64_t qbytes, void *twork,
                                       int64_t tbytes) {
  DT_CTX(c);
  DT_TRY(dt_check_hw(H, W));
  DT_REQUIRE((dem && fdr && out) || H * W == 0, "NULL raster");
  DT_REQUIRE(qwork != nullptr && qbytes >= dt_downslope_queue_workspace(H, W), "queue workspace missing or too small");
  DT_REQUIRE(twork == nullptr || tbytes >= dt_downslope_tables_workspace(H, W), "tables workspace too small");
  DT_TRY(dt_launch_downslope(c->stream, dt_full_window(H, W), dem, fdr, px, dz, raw, out, nullptr, qwork, twork, 2));
  DT_HIP(hipGetLastError());
  return DT_OK;
}

extern "C" int dt_dev_confusion_multi(dt_ctx *c, const double *desc, const int8_t *flood, int64_t N,
                                      double nodata_value, const double *th_host, int nth, int under,
                                      int64_t *counts4_dev) {
  DT_CTX(c);
  DT_REQUIRE(th_host && counts4_dev, "NULL thresholds / counts");
  DT_REQUIRE((desc && flood) || N == 0, "NULL raster");
  DT_TRY(dt_launch_confusion(c->stream, desc, flood, N, nodata_value, th_host, nth, under,
                             (unsigned long long *)counts4_dev));
  DT_HIP(hipGetLastError());
  return DT_OK;
}

extern "C" int dt_dev_unique_extremes_f32(dt_ctx *c, const float *x, int64_t N, float *out3_dev) {
  DT_CTX(c);
  DT_REQUIRE(x && out3_dev && N >= 0, "bad arguments");
  DT_TRY(dt_scratch_reset(c, 256));
  uint32_t *work = (uint32_t *)dt_scratch_take(c, 64);
  DT_TRY(dt_launch_unique_extremes(c->stream, x, N, work, out3_dev));
  DT_HIP(hipGetLastError());
  return DT_OK;
}

extern "C" int dt_dev_minmax_scale_f32(dt_ctx *c, const float *x, int64_t N, float mn, float mx, float nodata,
                                       double *desc) {
  DT_CTX(c);
  DT_REQUIRE((x && desc) || N == 0, "NULL raster");
  DT_TRY(dt_launch_minmax_scale(c->stream, x, N, mn, mx, nodata, desc));
  DT_HIP(hipGetLastError());
  return DT_OK;
}

extern "C" int dt_dev_minmax_scale_f32_f64(dt_ctx *c, const float *x, int64_t N, double mn, double mx, double nodata,
                                           double *desc) {
  DT_CTX(c);
  DT_REQUIRE((x && desc) || N == 0, "NULL raster");
  DT_TRY(dt_launch_minmax_scale_f32f64(c->stream, x, N, mn, mx, nodata, desc));
  DT_HIP(hipGetLastError());
  return DT_OK;
}

extern "C" int dt_dev_classify(dt_ctx *c, const double *desc, int8_t *flood, int64_t N, double nodata_value,
                               double threshold, int under, int remap_flood, uint8_t *binary, int32_t *klass,
                               int64_t *counts4_dev) {
  DT_CTX(c);
  DT_REQUIRE(counts4_dev != nullptr, "counts4 is NULL");
  DT_REQUIRE((desc && flood) || N == 0, "NULL raster");
  DT_TRY(dt_launch_classify_f64(c->stream, desc, nullptr, flood, N, nodata_value, threshold, under, remap_flood,
                                binary, klass, (unsigned long long *)counts4_dev));
  DT_HIP(hipGetLastError());
  return DT_OK;
}

extern "C" int dt_dev_membench_copy(dt_ctx *c, const float *a, float *b, int64_t N, int blocks) {
  DT_CTX(c);
  DT_REQUIRE(a && b && N >= 0 && blocks != 0, "bad arguments");
  DT_TRY(dt_launch_membench_copy(c->stream, a, b, N, blocks));
  DT_HIP(hipGetLastError());
  return DT_OK;
}

extern "C" int dt_dev_membench_mix(dt_ctx *c, const float *r0, const float *r1, float *w0, float *w1, float *w2,
                                   int64_t N, int n_reads, int n_writes, int nontemporal) {
  DT_CTX(c);
  DT_REQUIRE((n_reads < 1 || r0) && (n_reads < 2 || r1) && w0 && (n_writes < 2 || w1) && (n_writes < 3 || w2),
             "NULL stream");
  DT_TRY(dt_launch_membench_mix(c->stream, r0, r1, w0, w1, w2, N, n_reads, n_writes, nontemporal));
  DT_HIP(hipGetLastError());
  return DT_OK;
}

extern "C" int dt_dev_membench_mix_timed(dt_ctx *c, const float *r0, const float *r1, float *w0, float *w1, float *w2,
                                         int64_t N, int n_reads, int n_writes, int nontemporal, int reps, double *ms) {
  DT_CTX(c);
  DT_REQUIRE(ms && reps >= 1, "bad arguments");
  DT_TRY(dt_dev_membench_mix(c, r0, r1, w0, w1, w2, N, n_reads, n_writes, nontemporal));
  hipEvent_t e0, e1;
  DT_HIP(hipEventCreate(&e0));
  DT_HIP(hipEventCreate(&e1));
  int rc = DT_OK;
  float t = 0.0f;
  if (hipEventRecord(e0, c->stream) != hipSuccess) rc = DT_EHIP;
  for (int r = 0; r < reps && rc == DT_OK; r++)
    rc = dt_dev_membench_mix(c, r0, r1, w0, w1, w2, N, n_reads, n_writes, nontemporal);
  if (rc == DT_OK && (hipEventRecord(e1, c->stream) != hipSuccess || hipEventSynchronize(e1) != hipSuccess ||
                      hipEventElapsedTime(&t, e0, e1) != hipSuccess)) {
    dt_set_error("event timing failed: %s", hipGetErrorString(hipGetLastError()));
    rc = DT_EHIP;
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *ms = (double)t / reps;
  return rc;
}
extern "C" int dt_dev_mem_info(dt_ctx *c, int64_t *free_bytes, int64_t *total_bytes) {
  DT_CTX(c);
  size_t f = 0, t = 0;
  DT_HIP(hipMemGetInfo(&f, &t));
  if (free_bytes) *free_bytes = (int64_t)f;
  if (total_bytes) *total_bytes = (int64_t)t;
  return DT_OK;
}

extern "C" int dt_dev_i32_to_i64(dt_ctx *c, const int32_t *src, int64_t N, int64_t *dst) {
  DT_CTX(c);
  DT_TRY(dt_launch_i32_to_i64(c->stream, src, N, dst));
  DT_HIP(hipGetLastError());
  return DT_OK;
}
extern "C" int dt_dev_i64_to_i32(dt_ctx *c, const int64_t *src, int64_t N, int32_t *dst) {
  DT_CTX(c);
  DT_TRY(dt_launch_i64_to_i32(c->stream, src, N, dst));
  DT_HIP(hipGetLastError());
  return DT_OK;
}

// ---- windowed device tier (one rank's core window of a larger raster; multi-GPU) ------------------
extern "C" int64_t dt_perim_cells(int64_t H, int64_t W) { return dt_perim_count((int)H, (int)W); }

extern "C" int dt_dev_slope_d8_w(dt_ctx *c, const dt_window *win, const float *dem, double px, float *slope,
                                 uint8_t *fdr, float *slope_rad) {
  DT_CTX(c);
  DtWin w;
  DT_TRY(dt_convert_window(win, &w));
  DT_REQUIRE(dem && (slope || fdr || slope_rad), "NULL raster");
  DT_TRY(dt_side_reserve(c, &c->aux, &c->aux_bytes, dt_stencil_aux_bytes(w.H, w.W)));
  DT_TRY(dt_launch_stencil(c->stream, w, dem, px, slope, fdr, slope_rad, nullptr, 0, 0.0, nullptr, nullptr, c->aux));
  DT_HIP(hipGetLastError());
  return DT_OK;
}

static int dev_slope_twi_w(dt_ctx *c, const dt_window *win, const float *dem, const void *acc, int acc64, double px,
                           double n_top, float *slope, float *slope_rad, float *ti, float *mti) {
  DT_CTX(c);
  DtWin w;
  DT_TRY(dt_convert_window(win, &w));
  DT_REQUIRE(dem && acc && ti && mti, "NULL raster");
  DT_TRY(dt_side_reserve(c, &c->aux, &c->aux_bytes, dt_stencil_aux_bytes(w.H, w.W)));
  DT_TRY(dt_launch_stencil(c->stream, w, dem, px, slope, nullptr, slope_rad, acc, acc64, n_top, ti, mti, c->aux));
  DT_HIP(hipGetLastError());
  return DT_OK;
}
extern "C" int dt_dev_slope_twi_w(dt_ctx *c, const dt_window *win, const float *dem, const int32_t *acc32,
                                  double px, double n_top, float *slope, float *slope_rad, float *ti,
                                  float *mti) {
  return dev_slope_twi_w(c, win, dem, acc32, 0, px, n_top, slope, slope_rad, ti, mti);
}
extern "C" int dt_dev_slope_twi_w_a64(dt_ctx *c, const dt_window *win, const float *dem, const int64_t *acc64,
                                      double px, double n_top, float *slope, float *slope_rad, float *ti,
                                      float *mti) {
  return dev_slope_twi_w(c, win, dem, acc64, 1, px, n_top, slope, slope_rad, ti, mti);
}

extern "C" int dt_dev_downslope_w(dt_ctx *c, const dt_window *win, const float *dem, const uint8_t *fdr,
                                  double px, double dz, int raw, float *out, int32_t *n_unresolved_dev) {
  DT_CTX(c);
  DtWin w;
  DT_TRY(dt_convert_window(win, &w));
  DT_REQUIRE(dem && fdr && out, "NULL raster");
  if (n_unresolved_dev) DT_HIP(hipMemsetAsync(n_unresolved_dev, 0, sizeof(int32_t), c->stream));
  DT_TRY(dt_launch_downslope(c->stream, w, dem, fdr, px, dz, raw, out, (int *)n_unresolved_dev));
  DT_HIP(hipGetLastError());
  return DT_OK;
}

// dt_dev_downslope_w with the long-walk workspace: the long walks that stay in the rank's memory (core + halo) are
// queued and finished with skip tables over that memory; the ones that leave it are marked and counted as ever
extern "C" int64_t dt_downslope_lift_workspace_w(const dt_window *win) {
  DtWin w;
  if (dt_convert_window(win, &w) != DT_OK) return -1;
  return (int64_t)dt_downslope_lift_bytes_w(w);
}
extern "C" int dt_dev_downslope_lift_w(dt_ctx *c, const dt_window *win, const float *dem, const uint8_t *fdr,
                                       double px, double dz, int raw, float *out, int32_t *n_unresolved_dev,
                                       void *work, int64_t work_bytes) {
  DT_CTX(c);
  DtWin w;
  DT_TRY(dt_convert_window(win, &w));
  DT_REQUIRE(dem && fdr && out, "NULL raster");
  DT_REQUIRE(work != nullptr && work_bytes >= (int64_t)dt_downslope_lift_bytes_w(w),
             "downslope workspace missing or too small");
  if (n_unresolved_dev) DT_HIP(hipMemsetAsync(n_unresolved_dev, 0, sizeof(int32_t), c->stream));
  DT_TRY(dt_launch_downslope(c->stream, w, dem, fdr, px, dz, raw, out, (int *)n_unresolved_dev, work,
                             (char *)work + dt_downslope_queue_bytes(w.H, w.W), 0));
  DT_HIP(hipGetLastError());
  return DT_OK;
}

// dt_dev_downslope_w / _lift_w (work = NULL / the long-walk workspace) that also EMITS the walks leaving the rank's
// memory as walker records: walkers = [count u32, pad to 256 bytes | 48-byte records], see DsWalkOut in dt_kernels.hip.
// More walks than records fit: the count says so, the cells are marked -50 all the same.
extern "C" int dt_dev_downslope_emit_w(dt_ctx *c, const dt_window *win, const float *dem, const uint8_t *fdr,
                                       double px, double dz, int raw, float *out, int32_t *n_unresolved_dev,
                                       void *work, int64_t work_bytes, void *walkers, int64_t walkers_bytes) {
  DT_CTX(c);
  DtWin w;
  DT_TRY(dt_convert_window(win, &w));
  DT_REQUIRE(dem && fdr && out, "NULL raster");
  DT_REQUIRE(work == nullptr || work_bytes >= (int64_t)dt_downslope_lift_bytes_w(w), "downslope workspace too small");
  DT_REQUIRE(walkers != nullptr && walkers_bytes >= 256 + 48, "walker buffer missing or too small");
  if (n_unresolved_dev) DT_HIP(hipMemsetAsync(n_unresolved_dev, 0, sizeof(int32_t), c->stream));
  DT_TRY(dt_launch_downslope(c->stream, w, dem, fdr, px, dz, raw, out, (int *)n_unresolved_dev, work,
                             work ? (char *)work + dt_downslope_queue_bytes(w.H, w.W) : nullptr, 0, walkers,
                             (size_t)walkers_bytes));
  DT_HIP(hipGetLastError());
  return DT_OK;
}

// The walkers standing in this rank's memory advanced in place (k_ds_walk): rec = n records of 48 bytes; work
// (optional) = the rank's long-walk workspace as the downslope call of this step left it (its skip tables carry
// counting walkers across the rank 64 moves at a time).
extern "C" int dt_dev_downslope_walk_w(dt_ctx *c, const dt_window *win, const float *dem, const uint8_t *fdr, double px,
                                       double dz, int64_t n, void *rec, void *work, int64_t work_bytes) {
  DT_CTX(c);
  DtWin w;
  DT_TRY(dt_convert_window(win, &w));
  DT_REQUIRE(n >= 0, "negative count");
  DT_REQUIRE(n == 0 || (dem && fdr && rec), "NULL pointer");
  DT_REQUIRE(work == nullptr || work_bytes >= (int64_t)dt_downslope_lift_bytes_w(w), "downslope workspace too small");
  DT_TRY(dt_launch_ds_walk(c->stream, w, dem, fdr, px, dz, n, rec, work));
  DT_HIP(hipGetLastError());
  return DT_OK;
}
// One iteration of the walkers' journey, prepared on the device (tiling.finish_downslope): the records that arrived
// finished are home -- their value goes into `out` (this rank's downslope raster, core origin) -- the others advance like
// dt_dev_downslope_walk_w; then every record that is still wanted somewhere is copied into `send`, grouped by destination
// rank (a walker that has just finished: the owner of its start cell; the others: the owner of the cell they stand on),
// and counts[d] = records for rank d, counts[n_ranks] = how many of them are still on their way.  row_starts /
// col_starts: device arrays of ty + 1 / tx + 1 global rows / columns (the layout's bands and the raster's end);
// counts: int32[ty * tx + 1]; scratch: int32[n + ty * tx]; send: room for n records.
extern "C" int dt_dev_downslope_walk_route_w(dt_ctx *c, const dt_window *win, const float *dem, const uint8_t *fdr,
                                             double px, double dz, int64_t n, void *rec, void *work, int64_t work_bytes,
                                             float *out, const int32_t *row_starts, int32_t ty,
                                             const int32_t *col_starts, int32_t tx, void *send, int32_t *counts,
                                             int32_t *scratch) {
  DT_CTX(c);
  DtWin w;
  DT_TRY(dt_convert_window(win, &w));
  DT_REQUIRE(n >= 0, "negative count");
  DT_REQUIRE(ty >= 1 && tx >= 1 && row_starts && col_starts && counts, "layout / counts missing");
  DT_REQUIRE(n == 0 || (dem && fdr && rec && out && send && scratch), "NULL pointer");
  DT_REQUIRE(work == nullptr || work_bytes >= (int64_t)dt_downslope_lift_bytes_w(w), "downslope workspace too small");
  DT_TRY(dt_launch_ds_walk(c->stream, w, dem, fdr, px, dz, n, rec, work, out));
  DT_TRY(dt_launch_ds_route(c->stream, n, rec, row_starts, ty, col_starts, tx, send, counts, scratch));
  DT_HIP(hipGetLastError());
  return DT_OK;
}
// records of walkers at their start cells (core coordinates ys / xs of n cells), no move made: for cells that are
// marked -50 without a record (emission buffer too small, or a tile without one)
extern "C" int dt_dev_downslope_walk_seed_w(dt_ctx *c, const dt_window *win, const float *dem, int64_t n,
                                            const int32_t *ys, const int32_t *xs, void *rec) {
  DT_CTX(c);
  DtWin w;
  DT_TRY(dt_convert_window(win, &w));
  DT_REQUIRE(n >= 0, "negative count");
  DT_REQUIRE(n == 0 || (dem && ys && xs && rec), "NULL pointer");
  DT_TRY(dt_launch_ds_walk_seed(c->stream, w, dem, n, ys, xs, rec));
  DT_HIP(hipGetLastError());
  return DT_OK;
}

extern "C" int dt_dev_flowacc_local_w(dt_ctx *c, const dt_window *win, const uint8_t *fdr, int32_t *acc32,
                                      int64_t *A_perim, int32_t *xr_perim, uint8_t *code_perim) {
  DT_CTX(c);
  DtWin w;
  DT_TRY(dt_convert_window(win, &w));
  DT_REQUIRE(fdr && A_perim && xr_perim && code_perim, "NULL pointer");  // acc32 is not touched by phase 1
  // HAND's workspace is reserved beside flow accumulation's, so that phase 2 can run fused with HAND's phase 1
  // (dt_dev_flowacc_finish_flowhand_local_w) without disturbing the state this call leaves
  size_t need = dt_flowacc_tiled_scratch(w.H, w.W), need2 = dt_flowhand_tiled_scratch(w.H, w.W);
  DT_TRY(dt_scratch_reset(c, need + need2 + 512));
  void *scr = dt_scratch_take(c, need), *scr2 = dt_scratch_take(c, need2);
  DT_REQUIRE(scr && scr2, "scratch reservation failed");
  DT_TRY(dt_launch_fa_local(c->stream, w, fdr, scr, need, acc32, 1));
  DT_TRY(dt_launch_fa_summary(c->stream, w, scr, A_perim, xr_perim, code_perim));
  DT_HIP(hipGetLastError());
  c->scratch_owner = 1;
  c->owner_h = w.H;
  c->owner_w = w.W;
  c->owner_ptr = (char *)scr;
  c->owner_ptr2 = (char *)scr2;
  return DT_OK;
}

// must follow dt_dev_flowacc_local_w on the same context with no other scratch-using call in between
static int dev_flowacc_finish_w(dt_ctx *c, const dt_window *win, const uint8_t *fdr, const float *dem,
                                const uint64_t *ext_perim, int64_t threshold, void *acc, int acc64, int8_t *river) {
  DT_CTX(c);
  DtWin w;
  DT_TRY(dt_convert_window(win, &w));
  DT_REQUIRE(fdr && acc, "NULL raster");
  DT_REQUIRE(c->scratch && c->scratch_owner == 1 && c->owner_h == w.H && c->owner_w == w.W,
             "dt_dev_flowacc_finish_w without a matching dt_dev_flowacc_local_w on this context (another call has "
             "used the context's scratch in between)");
  DT_TRY(dt_launch_fa_finish(c->stream, w, fdr, dem, c->owner_ptr, (const unsigned long long *)ext_perim, threshold,
                             acc, acc64, river, c->status));
  DT_HIP(hipGetLastError());
  return DT_OK;
}

// phase 2 of flow accumulation and phase 1 of HAND in one call (the last accumulation tile pass and HAND's first
// share the tile's codes and the river mask: one kernel in the common form, dt_launch_fa_finish_fh_local); leaves the
// context in the state dt_dev_flowhand_local_w leaves it in
static int dev_flowacc_finish_fh_local_w(dt_ctx *c, const dt_window *win, const uint8_t *fdr, const float *dem,
                                         const uint64_t *ext_perim, int64_t threshold, void *acc, int acc64,
                                         int8_t *river, uint8_t *kind, int32_t *ref, int32_t *nc, int32_t *nd,
                                         float *zr, int64_t *ar) {
  DT_CTX(c);
  DtWin w;
  DT_TRY(dt_convert_window(win, &w));
  DT_REQUIRE(fdr && acc && river && kind && ref && nc && nd && zr && ar, "NULL pointer");
  DT_REQUIRE(c->scratch && c->scratch_owner == 1 && c->owner_h == w.H && c->owner_w == w.W && c->owner_ptr2,
             "dt_dev_flowacc_finish_flowhand_local_w without a matching dt_dev_flowacc_local_w on this context (another "
             "call has used the context's scratch in between)");
  DT_TRY(dt_launch_fa_finish_fh_local(c->stream, w, fdr, dem, c->owner_ptr, c->owner_ptr2,
                                      dt_flowhand_tiled_scratch(w.H, w.W), (const unsigned long long *)ext_perim, threshold,
                                      acc, acc64, river, c->status));
  DT_TRY(dt_launch_fh_summary(c->stream, w, c->owner_ptr2, dem, acc, acc64, kind, ref, nc, nd, zr, (long long *)ar));
  DT_HIP(hipGetLastError());
  c->scratch_owner = 2;
  c->owner_ptr = c->owner_ptr2;
  c->owner_ptr2 = nullptr;
  return DT_OK;
}
extern "C" int dt_dev_flowacc_finish_flowhand_local_w(dt_ctx *c, const dt_window *win, const uint8_t *fdr,
                                                      const float *dem, const uint64_t *ext_perim, int64_t threshold,
                                                      int32_t *acc32, int8_t *river, uint8_t *kind, int32_t *ref,
                                                      int32_t *nc, int32_t *nd, float *zr, int64_t *ar) {
  return dev_flowacc_finish_fh_local_w(c, win, fdr, dem, ext_perim, threshold, acc32, 0, river, kind, ref, nc, nd, zr, ar);
}
extern "C" int dt_dev_flowacc_finish_flowhand_local_w_a64(dt_ctx *c, const dt_window *win, const uint8_t *fdr,
                                                          const float *dem, const uint64_t *ext_perim,
                                                          int64_t threshold, int64_t *acc64, int8_t *river,
                                                          uint8_t *kind, int32_t *ref, int32_t *nc, int32_t *nd,
                                                          float *zr, int64_t *ar) {
  return dev_flowacc_finish_fh_local_w(c, win, fdr, dem, ext_perim, threshold, acc64, 1, river, kind, ref, nc, nd, zr, ar);
}

// single raster: flow accumulation (all phases), river mask and HAND's phase 1; dt_dev_flowhand_finish_w /
// dt_dev_flowhand_gfi_finish_w with the whole raster as the window follow
static int dev_flowacc_river_flowhand_local(dt_ctx *c, const uint8_t *fdr, const float *dem, const uint8_t *nod4,
                                           int64_t H, int64_t W, int64_t threshold, int32_t *acc32, int8_t *river) {
  DT_CTX(c);
  DT_TRY(dt_check_hw(H, W));
  DT_REQUIRE((fdr && acc32 && river) || H * W == 0, "NULL raster");
  size_t need = dt_flowacc_tiled_scratch(H, W), need2 = dt_flowhand_tiled_scratch(H, W);
  DT_TRY(dt_scratch_reset(c, need + need2 + 512));
  void *scr = dt_scratch_take(c, need), *scr2 = dt_scratch_take(c, need2);
  DT_REQUIRE(scr && scr2, "scratch reservation failed");
  DtWin w = dt_full_window(H, W);
  DT_TRY(dt_launch_fa_local(c->stream, w, fdr, scr, need, acc32, 0));
  DT_TRY(dt_launch_fa_finish_fh_local(c->stream, w, fdr, dem, scr, scr2, need2, nullptr, threshold, acc32, 0, river,
                                      c->status, nod4, dt_nodata4_ld(W)));
  DT_HIP(hipGetLastError());
  c->scratch_owner = 2;
  c->owner_h = H;
  c->owner_w = W;
  c->owner_ptr = (char *)scr2;
  c->owner_ptr2 = nullptr;
  return DT_OK;
}
extern "C" int dt_dev_flowacc_river_flowhand_local(dt_ctx *c, const uint8_t *fdr, const float *dem, int64_t H, int64_t W,
                                                   int64_t threshold, int32_t *acc32, int8_t *river) {
  return dev_flowacc_river_flowhand_local(c, fdr, dem, nullptr, H, W, threshold, acc32, river);
}
// ... with the nodata mask the D8 kernel wrote (dt_dev_slope_d8_m): the pass reads 0.125 instead of 4 bytes per cell to
// learn which cells are nodata.  `dem` is still required (it serves the raster shapes the fused kernel does not take).
extern "C" int dt_dev_flowacc_river_flowhand_local_m(dt_ctx *c, const uint8_t *fdr, const float *dem,
                                                     const uint8_t *nodata4, int64_t H, int64_t W, int64_t threshold,
                                                     int32_t *acc32, int8_t *river) {
  DT_REQUIRE((dem && nodata4) || H * W == 0, "dem and the nodata mask are both required");
  return dev_flowacc_river_flowhand_local(c, fdr, dem, nodata4, H, W, threshold, acc32, river);
}
extern "C" int64_t dt_nodata_mask_bytes(int64_t H, int64_t W) {
  return (H < 0 || W < 0) ? -1 : (int64_t)dt_nodata4_bytes(H, W);
}
// D8 codes (the hot / cold kernel pair) and, on the way, the nodata mask: one 16-bit word per 4 x 4 patch of cells
// (bit 4 j + k = cell (4 r + j, 4 i + k) holds the sentinel, z <= -100)
extern "C" int dt_dev_slope_d8_m(dt_ctx *c, const float *dem, int64_t H, int64_t W, double px, uint8_t *fdr,
                                 uint8_t *nodata4) {
  DT_CTX(c);
  DT_TRY(dt_check_hw(H, W));
  DT_REQUIRE((dem && fdr && nodata4) || H * W == 0, "NULL raster");
  DT_TRY(dt_side_reserve(c, &c->aux, &c->aux_bytes, dt_stencil_aux_bytes(H, W)));
  DT_TRY(dt_launch_stencil(c->stream, dt_full_window(H, W), dem, px, nullptr, fdr, nullptr, nullptr, 0, 0.0, nullptr,
                           nullptr, c->aux, nodata4, dt_nodata4_ld(W)));
  DT_HIP(hipGetLastError());
  return DT_OK;
}
extern "C" int dt_dev_flowacc_finish_w(dt_ctx *c, const dt_window *win, const uint8_t *fdr, const float *dem,
                                       const uint64_t *ext_perim, int64_t threshold, int32_t *acc32,
                                       int8_t *river) {
  return dev_flowacc_finish_w(c, win, fdr, dem, ext_perim, threshold, acc32, 0, river);
}
extern "C" int dt_dev_flowacc_finish_w_a64(dt_ctx *c, const dt_window *win, const uint8_t *fdr, const float *dem,
                                           const uint64_t *ext_perim, int64_t threshold, int64_t *acc64,
                                           int8_t *river) {
  return dev_flowacc_finish_w(c, win, fdr, dem, ext_perim, threshold, acc64, 1, river);
}

static int dev_flowhand_local_w(dt_ctx *c, const dt_window *win, const float *dem, const uint8_t *fdr,
                                const int8_t *river, const void *acc, int acc64, uint8_t *kind, int32_t *ref,
                                int32_t *nc, int32_t *nd, float *zr, int64_t *ar) {
  DT_CTX(c);
  DtWin w;
  DT_TRY(dt_convert_window(win, &w));
  DT_REQUIRE(fdr && river && kind && ref && nc && nd && zr && ar, "NULL pointer");
  size_t need = dt_flowhand_tiled_scratch(w.H, w.W);
  DT_TRY(dt_scratch_reset(c, need));
  void *scr = dt_scratch_take(c, need);
  DT_TRY(dt_launch_fh_local(c->stream, w, fdr, river, scr, need));
  DT_TRY(dt_launch_fh_summary(c->stream, w, scr, dem, acc, acc64, kind, ref, nc, nd, zr, (long long *)ar));
  DT_HIP(hipGetLastError());
  c->scratch_owner = 2;
  c->owner_h = w.H;
  c->owner_w = w.W;
  c->owner_ptr = (char *)scr;
  c->owner_ptr2 = nullptr;
  return DT_OK;
}
extern "C" int dt_dev_flowhand_local_w(dt_ctx *c, const dt_window *win, const float *dem, const uint8_t *fdr,
                                       const int8_t *river, const int32_t *acc32, uint8_t *kind, int32_t *ref,
                                       int32_t *nc, int32_t *nd, float *zr, int64_t *ar) {
  return dev_flowhand_local_w(c, win, dem, fdr, river, acc32, 0, kind, ref, nc, nd, zr, ar);
}
extern "C" int dt_dev_flowhand_local_w_a64(dt_ctx *c, const dt_window *win, const float *dem, const uint8_t *fdr,
                                           const int8_t *river, const int64_t *acc64, uint8_t *kind, int32_t *ref,
                                           int32_t *nc, int32_t *nd, float *zr, int64_t *ar) {
  return dev_flowhand_local_w(c, win, dem, fdr, river, acc64, 1, kind, ref, nc, nd, zr, ar);
}

// must follow dt_dev_flowhand_local_w on the same context with no other scratch-using call in between; `fused`: with
// the GFI / ln(hl/H) epilogue (see dt_dev_flowhand_gfi)
static int dev_flowhand_finish_w(dt_ctx *c, const dt_window *win, const float *dem, const uint8_t *fdr,
                                 const int8_t *river, const void *acc, int acc64, double px, double n_gfi, double b,
                                 const uint8_t *res_ok, const int32_t *res_nc, const int32_t *res_nd,
                                 const int64_t *rem_gidx, const float *rem_zr, const int64_t *rem_ar, float *fdist,
                                 int32_t *idx32, int64_t *idx64, float *hand, void *a_river, float *gfi,
                                 float *lnhlh, bool fused) {
  DT_CTX(c);
  DtWin w;
  DT_TRY(dt_convert_window(win, &w));
  DT_REQUIRE(fdr && river, "NULL raster");
  DT_REQUIRE(!hand || dem, "hand needs dem");
  DT_REQUIRE(!a_river || acc, "a_river needs the accumulation raster");
  DT_REQUIRE(!fused || (dem && acc && gfi && lnhlh), "NULL raster");
  DT_REQUIRE(!res_ok || (res_nc && res_nd && rem_gidx && rem_zr && rem_ar), "incomplete rank-exit results");
  DT_REQUIRE(c->scratch && c->scratch_owner == 2 && c->owner_h == w.H && c->owner_w == w.W,
             "flowhand finish without a matching dt_dev_flowhand_local_w on this context (another call has used the "
             "context's scratch in between)");
  DT_TRY(dt_launch_fh_finish(c->stream, w, dem, fdr, river, acc, acc64, px, c->owner_ptr, res_ok, res_nc, res_nd,
                             (const long long *)rem_gidx, rem_zr, (const long long *)rem_ar, fdist, idx32,
                             (long long *)idx64, hand, a_river, fused ? gfi : nullptr, fused ? lnhlh : nullptr, n_gfi, b,
                             px));
  DT_HIP(hipGetLastError());
  return DT_OK;
}
extern "C" int dt_dev_flowhand_finish_w(dt_ctx *c, const dt_window *win, const float *dem, const uint8_t *fdr,
                                        const int8_t *river, const int32_t *acc32, double px,
                                        const uint8_t *res_ok, const int32_t *res_nc, const int32_t *res_nd,
                                        const int64_t *rem_gidx, const float *rem_zr, const int64_t *rem_ar,
                                        float *fdist, int32_t *idx32, int64_t *idx64, float *hand,
                                        int32_t *a_river) {
  return dev_flowhand_finish_w(c, win, dem, fdr, river, acc32, 0, px, 0.0, 1.0, res_ok, res_nc, res_nd, rem_gidx, rem_zr,
                               rem_ar, fdist, idx32, idx64, hand, a_river, nullptr, nullptr, false);
}
extern "C" int dt_dev_flowhand_finish_w_a64(dt_ctx *c, const dt_window *win, const float *dem, const uint8_t *fdr,
                                            const int8_t *river, const int64_t *acc64, double px,
                                            const uint8_t *res_ok, const int32_t *res_nc, const int32_t *res_nd,
                                            const int64_t *rem_gidx, const float *rem_zr, const int64_t *rem_ar,
                                            float *fdist, int32_t *idx32, int64_t *idx64, float *hand,
                                            int64_t *a_river64) {
  return dev_flowhand_finish_w(c, win, dem, fdr, river, acc64, 1, px, 0.0, 1.0, res_ok, res_nc, res_nd, rem_gidx, rem_zr,
                               rem_ar, fdist, idx32, idx64, hand, a_river64, nullptr, nullptr, false);
}
extern "C" int dt_dev_flowhand_gfi_finish_w(dt_ctx *c, const dt_window *win, const float *dem, const uint8_t *fdr,
                                            const int8_t *river, const int32_t *acc32, double px, double n_gfi,
                                            double b, const uint8_t *res_ok, const int32_t *res_nc,
                                            const int32_t *res_nd, const int64_t *rem_gidx, const float *rem_zr,
                                            const int64_t *rem_ar, float *fdist, int32_t *idx32, int64_t *idx64,
                                            float *hand, int32_t *a_river, float *gfi, float *lnhlh) {
  return dev_flowhand_finish_w(c, win, dem, fdr, river, acc32, 0, px, n_gfi, b, res_ok, res_nc, res_nd, rem_gidx, rem_zr,
                               rem_ar, fdist, idx32, idx64, hand, a_river, gfi, lnhlh, true);
}
extern "C" int dt_dev_flowhand_gfi_finish_w_a64(dt_ctx *c, const dt_window *win, const float *dem,
                                                const uint8_t *fdr, const int8_t *river, const int64_t *acc64,
                                                double px, double n_gfi, double b, const uint8_t *res_ok,
                                                const int32_t *res_nc, const int32_t *res_nd,
                                                const int64_t *rem_gidx, const float *rem_zr, const int64_t *rem_ar,
                                                float *fdist, int32_t *idx32, int64_t *idx64, float *hand,
                                                int64_t *a_river64, float *gfi, float *lnhlh) {
  return dev_flowhand_finish_w(c, win, dem, fdr, river, acc64, 1, px, n_gfi, b, res_ok, res_nc, res_nd, rem_gidx, rem_zr,
                               rem_ar, fdist, idx32, idx64, hand, a_river64, gfi, lnhlh, true);
}

static int dt_scratch2_reserve(dt_ctx *c, size_t bytes) {
  return dt_side_reserve(c, &c->scratch2, &c->scratch2_bytes, bytes);
}

extern "C" int dt_dev_rank_solve_flowacc(dt_ctx *c, int ty, int tx, const int64_t *heights, const int64_t *widths,
                                         int64_t Pmax, const void *rows_dev, int64_t rowbytes,
                                         const int64_t *field_offsets3, int rank, int64_t P_rank,
                                         uint64_t *ext_out_dev) {
  DT_CTX(c);
  DT_REQUIRE(heights && widths && rows_dev && field_offsets3 && ext_out_dev, "NULL pointer");
  DT_REQUIRE(rank >= 0 && rank < ty * tx && P_rank >= 0 && P_rank <= Pmax, "bad rank / ring size");
  DT_TRY(dt_scratch2_reserve(c, dt_rank_solve_scratch(ty * tx, Pmax)));
  DT_TRY(dt_launch_rank_solve_flowacc(c->stream, ty, tx, heights, widths, Pmax, rows_dev, rowbytes, field_offsets3,
                                      rank, P_rank, c->scratch2, (unsigned long long *)ext_out_dev));
  DT_HIP(hipGetLastError());
  return DT_OK;
}

extern "C" int dt_dev_rank_solve_flowhand(dt_ctx *c, int ty, int tx, const int64_t *heights, const int64_t *widths,
                                          int64_t Pmax, const void *rows_dev, int64_t rowbytes,
                                          const int64_t *field_offsets7, int rank, int64_t P_rank,
                                          uint8_t *res_ok, int32_t *res_nc, int32_t *res_nd, int64_t *rem_gidx,
                                          float *rem_zr, int64_t *rem_ar) {
  DT_CTX(c);
  DT_REQUIRE(heights && widths && rows_dev && field_offsets7 && res_ok && res_nc && res_nd && rem_gidx && rem_zr &&
                 rem_ar, "NULL pointer");
  DT_REQUIRE(rank >= 0 && rank < ty * tx && P_rank >= 0 && P_rank <= Pmax, "bad rank / ring size");
  DT_TRY(dt_scratch2_reserve(c, dt_rank_solve_scratch(ty * tx, Pmax)));
  DT_TRY(dt_launch_rank_solve_flowhand(c->stream, ty, tx, heights, widths, Pmax, rows_dev, rowbytes,
                                       field_offsets7, rank, P_rank, c->scratch2, res_ok, res_nc, res_nd,
                                       (long long *)rem_gidx, rem_zr, (long long *)rem_ar));
  DT_HIP(hipGetLastError());
  return DT_OK;
}

// ---- host tier: H2D, kernels, D2H on a process-wide default context ----------------------------
static dt_ctx *g_host_ctx = nullptr;
static std::mutex g_host_mu;  // whole-call granularity, as SURVEY.md 8b (threading) allows

static int host_ctx(dt_ctx **out) {
  if (!g_host_ctx) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    DT_TRY(dt_ctx_create(dev, nullptr, &g_host_ctx));
  }
  *out = g_host_ctx;
  return hipSetDevice(g_host_ctx->device) == hipSuccess ? DT_OK : DT_EHIP;
}

// Device blocks of the host tier, cached per process: a drop-in caller makes one dt_<op> call per descriptor and
// raster-sized hipMalloc / hipFree pairs cost more than the kernels.  Blocks are handed out best-fit (within 25 %
// of the request) and kept when released; dt_host_trim() frees the idle ones.  Guarded by g_host_mu.
struct HostBlock {
  void *p;
  size_t bytes;
  bool busy;
};
static std::vector<HostBlock> g_blocks;
static void *host_block_take(size_t bytes) {
  int best = -1;
  for (size_t i = 0; i < g_blocks.size(); i++)
    if (!g_blocks[i].busy && g_blocks[i].bytes >= bytes && g_blocks[i].bytes <= bytes + bytes / 4 + 4096 &&
        (best < 0 || g_blocks[i].bytes < g_blocks[(size_t)best].bytes))
      best = (int)i;
  if (best >= 0) {
    g_blocks[(size_t)best].busy = true;
    return g_blocks[(size_t)best].p;
  }
  void *p = nullptr;
  if (hipMalloc(&p, bytes) != hipSuccess) {
    // out of memory: drop every idle block and try once more
    for (size_t i = 0; i < g_blocks.size();) {
      if (!g_blocks[i].busy) {
        (void)hipFree(g_blocks[i].p);
        g_blocks.erase(g_blocks.begin() + (long)i);
      } else {
        i++;
      }
    }
    (void)hipGetLastError();
    if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
  }
  g_blocks.push_back(HostBlock{p, bytes, true});
  return p;
}
// idle blocks kept beyond this many bytes are freed on release (DT_HOST_CACHE_MB, default 4096; 0 = keep nothing)
static size_t host_cache_budget() {
  static long long mb = -1;
  if (mb < 0) {
    const char *e = getenv("DT_HOST_CACHE_MB");
    mb = e ? atoll(e) : 4096;
    if (mb < 0) mb = 0;
  }
  return (size_t)mb << 20;
}
static void host_blocks_drop_idle(size_t keep_bytes) {
  // largest idle blocks go first; the host tier's stream is idle whenever a block is released (every dt_<op> ends
  // with a synchronisation before its DevBufs go out of scope)
  for (;;) {
    size_t idle = 0;
    int big = -1;
    for (size_t i = 0; i < g_blocks.size(); i++)
      if (!g_blocks[i].busy) {
        idle += g_blocks[i].bytes;
        if (big < 0 || g_blocks[i].bytes > g_blocks[(size_t)big].bytes) big = (int)i;
      }
    if (idle <= keep_bytes || big < 0) return;
    (void)hipFree(g_blocks[(size_t)big].p);
    g_blocks.erase(g_blocks.begin() + big);
  }
}
static void host_block_release(void *p) {
  for (auto &b : g_blocks)
    if (b.p == p) b.busy = false;
  host_blocks_drop_idle(host_cache_budget());
}
// RAII device buffer of the host tier
struct DevBuf {
  void *p = nullptr;
  ~DevBuf() {
    if (p) host_block_release(p);
  }
  int alloc(size_t bytes) {
    if (bytes == 0) bytes = 16;
    p = host_block_take(bytes);
    if (!p) {
      dt_set_error("hipMalloc of %zu bytes failed", bytes);
      return DT_ENOMEM;
    }
    return DT_OK;
  }
  template <typename T>
  T *as() {
    return (T *)p;
  }
};
extern "C" int dt_host_trim(void) {
  std::lock_guard<std::mutex> lk(g_host_mu);
  if (g_host_ctx) DT_HIP(hipStreamSynchronize(g_host_ctx->stream));
  for (size_t i = 0; i < g_blocks.size();) {
    if (!g_blocks[i].busy) {
      DT_HIP(hipFree(g_blocks[i].p));
      g_blocks.erase(g_blocks.begin() + (long)i);
    } else {
      i++;
    }
  }
  return DT_OK;
}
// page-locked host memory for rasters that cross PCIe at full rate (the Python package keeps a pool of these
// behind the arrays chain.run_host returns)
// Page-locked host memory for rasters.  Large blocks are anonymous mappings on transparent huge pages, first touched
// by a few threads, then registered with the runtime: 9-15 ms per GiB on the MI355X host against 125 ms for
// hipHostMalloc (which faults and pins 4 KiB pages one thread at a time), the same 57 GB/s device-to-host
// (tools/micro/host_alloc.cpp, profiles/r3/host_alloc.txt).  Falls back to hipHostMalloc when the mapping or the
// registration is refused.
struct HostMap {
  void *map;
  size_t map_bytes;
};
static std::mutex g_hostmap_mu;
static std::vector<std::pair<void *, HostMap>> g_hostmaps;  // registered blocks by their aligned address
static const size_t DT_HUGE = (size_t)2 << 20;
// Huge pages are asked for until one such block takes longer to map and touch than plain pages would (a host whose
// memory is fragmented compacts it inside the page faults: seconds per GiB have been seen); 4 KiB pages from then on.
static std::atomic<bool> g_host_thp{true};

extern "C" int dt_host_alloc(int64_t bytes, void **out) {
  DT_REQUIRE(out && bytes >= 0, "bad arguments");
  *out = nullptr;
  if ((size_t)bytes >= 16 * DT_HUGE) {
    const size_t n = ((size_t)bytes + DT_HUGE - 1) & ~(DT_HUGE - 1);
    void *m = mmap(nullptr, n + DT_HUGE, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (m != MAP_FAILED) {
      char *a = (char *)(((uintptr_t)m + DT_HUGE - 1) & ~(uintptr_t)(DT_HUGE - 1));
      const bool thp = g_host_thp.load();
      if (thp) (void)madvise(a, n, MADV_HUGEPAGE);
      const auto t_touch = std::chrono::steady_clock::now();
      unsigned hc = std::thread::hardware_concurrency();
      const int threads = (int)(hc == 0 ? 1 : (hc > 8 ? 8 : hc));
      const size_t per = ((n / (size_t)threads) + DT_HUGE - 1) & ~(DT_HUGE - 1);
      std::vector<std::thread> th;
      auto touch = [=](int t) {
        for (size_t o = per * (size_t)t; o < n && o < per * (size_t)(t + 1); o += 4096) a[o] = 0;
      };
      for (int t = 0; t < threads; t++) {
        try {
          th.emplace_back(touch, t);
        } catch (const std::system_error &) {  // no thread to be had: this one does the work (nothing may be thrown
          touch(t);                             // through extern "C")
        }
      }
      for (auto &t : th) t.join();
      const double touch_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_touch).count();
      if (thp && touch_s > 0.1 * ((double)n / (double)(1ull << 30)) + 0.02) g_host_thp.store(false);
      if (hipHostRegister(a, n, hipHostRegisterPortable) == hipSuccess) {
        std::lock_guard<std::mutex> lk(g_hostmap_mu);
        g_hostmaps.push_back({(void *)a, HostMap{m, n + DT_HUGE}});
        *out = a;
        return DT_OK;
      }
      (void)hipGetLastError();
      munmap(m, n + DT_HUGE);
    }
  }
  DT_HIP(hipHostMalloc(out, bytes > 0 ? (size_t)bytes : 16, hipHostMallocDefault));
  return DT_OK;
}
extern "C" int dt_host_free(void *p) {
  if (!p) return DT_OK;
  HostMap hm{nullptr, 0};
  {
    std::lock_guard<std::mutex> lk(g_hostmap_mu);
    for (size_t i = 0; i < g_hostmaps.size(); i++)
      if (g_hostmaps[i].first == p) {
        hm = g_hostmaps[i].second;
        g_hostmaps.erase(g_hostmaps.begin() + (long)i);
        break;
      }
  }
  if (hm.map) {
    DT_HIP(hipHostUnregister(p));
    munmap(hm.map, hm.map_bytes);
    return DT_OK;
  }
  DT_HIP(hipHostFree(p));
  return DT_OK;
}
// float32 -> float64 on the host with a few threads (the reference's containers are float64 rasters holding float32
// values: numpy's single-threaded astype over a 16384^2 raster costs more than the kernel and both copies together)
extern "C" int dt_host_f32_to_f64(const float *src, double *dst, int64_t n) {
  DT_REQUIRE((src && dst) || n == 0, "NULL pointer");
  DT_REQUIRE(n >= 0, "negative size");
  unsigned hc = std::thread::hardware_concurrency();
  int threads = (int)(hc == 0 ? 1 : (hc > 8 ? 8 : hc));
  if (n < (int64_t)1 << 22) threads = 1;
  const int64_t per = (n + threads - 1) / threads;
  auto work = [=](int t) {
    const int64_t a = per * t, b = a + per < n ? a + per : n;
    for (int64_t i = a; i < b; i++) dst[i] = (double)src[i];
  };
  if (threads == 1) {
    work(0);
    return DT_OK;
  }
  std::vector<std::thread> th;
  for (int t = 0; t < threads; t++) {
    try {
      th.emplace_back(work, t);
    } catch (const std::system_error &) {  // no thread to be had: this one does the work
      work(t);
    }
  }
  for (auto &t : th) t.join();
  return DT_OK;
}
#define H2D(dst, src, bytes, c) DT_HIP(hipMemcpyAsync((dst).p, (src), (bytes), hipMemcpyHostToDevice, (c)->stream))
#define D2H(dst, src, bytes, c) DT_HIP(hipMemcpyAsync((dst), (src).p, (bytes), hipMemcpyDeviceToHost, (c)->stream))

extern "C" int dt_synth_dem(uint32_t seed, int64_t Hg, int64_t Wg, int64_t y0, int64_t x0, int64_t h,
                            int64_t w, int nodata_pct, float *out) {
  std::lock_guard<std::mutex> lk(g_host_mu);
  dt_ctx *c;
  DT_TRY(host_ctx(&c));
  size_t n = (size_t)h * w;
  DevBuf d;
  DT_TRY(d.alloc(n * 4));
  DT_TRY(dt_dev_synth_dem(c, seed, Hg, Wg, y0, x0, h, w, nodata_pct, d.as<float>()));
  D2H(out, d, n * 4, c);
  return dt_ctx_sync(c);
}

static int host_slope_d8(const float *dem, int64_t H, int64_t W, double px, float *slope, uint8_t *fdr) {
  std::lock_guard<std::mutex> lk(g_host_mu);
  dt_ctx *c;
  DT_TRY(host_ctx(&c));
  DT_TRY(dt_check_hw(H, W));
  DT_REQUIRE(dem || H * W == 0, "dem is NULL");
  size_t n = (size_t)H * W;
  if (n == 0) return DT_OK;
  DevBuf d_dem, d_sl, d_f;
  DT_TRY(d_dem.alloc(n * 4));
  if (slope) DT_TRY(d_sl.alloc(n * 4));
  if (fdr) DT_TRY(d_f.alloc(n));
  H2D(d_dem, dem, n * 4, c);
  DT_TRY(dt_dev_slope_d8(c, d_dem.as<float>(), H, W, px, slope ? d_sl.as<float>() : nullptr,
                         fdr ? d_f.as<uint8_t>() : nullptr, nullptr));
  if (slope) D2H(slope, d_sl, n * 4, c);
  if (fdr) D2H(fdr, d_f, n, c);
  return dt_ctx_sync(c);
}

extern "C" int dt_slope_f32(const float *dem, int64_t H, int64_t W, double px, float *slope) {
  DT_REQUIRE(slope || H * W == 0, "slope is NULL");
  return host_slope_d8(dem, H, W, px, slope, nullptr);
}
extern "C" int dt_d8_f32(const float *dem, int64_t H, int64_t W, double px, uint8_t *fdr, float *slope) {
  DT_REQUIRE(fdr || H * W == 0, "fdr is NULL");
  return host_slope_d8(dem, H, W, px, slope, fdr);
}

extern "C" int dt_flowacc_u8(const uint8_t *fdr, const float *dem, int64_t H, int64_t W, int64_t *acc) {
  std::lock_guard<std::mutex> lk(g_host_mu);
  dt_ctx *c;
  DT_TRY(host_ctx(&c));
  DT_TRY(dt_check_hw(H, W));
  size_t n = (size_t)H * W;
  if (n == 0) return DT_OK;
  DT_REQUIRE(fdr && acc, "NULL raster");
  DevBuf d_f, d_dem, d_a32, d_a64;
  DT_TRY(d_f.alloc(n));
  DT_TRY(d_a32.alloc(n * 4));
  DT_TRY(d_a64.alloc(n * 8));
  H2D(d_f, fdr, n, c);
  if (dem) {
    DT_TRY(d_dem.alloc(n * 4));
    H2D(d_dem, dem, n * 4, c);
  }
  DT_TRY(dt_dev_flowacc(c, d_f.as<uint8_t>(), dem ? d_dem.as<float>() : nullptr, H, W, d_a32.as<int32_t>()));
  DT_TRY(dt_dev_i32_to_i64(c, d_a32.as<int32_t>(), (int64_t)n, d_a64.as<int64_t>()));
  D2H(acc, d_a64, n * 8, c);
  return dt_ctx_sync(c);
}

extern "C" int dt_flowhand(const float *dem, const uint8_t *fdr, const int8_t *river, int64_t H, int64_t W,
                           double px, float *fdist, int64_t *idx, float *hand) {
  std::lock_guard<std::mutex> lk(g_host_mu);
  dt_ctx *c;
  DT_TRY(host_ctx(&c));
  DT_TRY(dt_check_hw(H, W));
  size_t n = (size_t)H * W;
  if (n == 0) return DT_OK;
  DT_REQUIRE(fdr && river, "NULL raster");
  DT_REQUIRE(!hand || dem, "hand needs dem");
  DevBuf d_dem, d_f, d_r, d_fd, d_i32, d_i64, d_h;
  DT_TRY(d_f.alloc(n));
  DT_TRY(d_r.alloc(n));
  H2D(d_f, fdr, n, c);
  H2D(d_r, river, n, c);
  if (hand) {
    DT_TRY(d_dem.alloc(n * 4));
    H2D(d_dem, dem, n * 4, c);
    DT_TRY(d_h.alloc(n * 4));
  }
  if (fdist) DT_TRY(d_fd.alloc(n * 4));
  if (idx) {
    DT_TRY(d_i32.alloc(n * 4));
    DT_TRY(d_i64.alloc(n * 8));
  }
  DT_TRY(dt_dev_flowhand(c, hand ? d_dem.as<float>() : nullptr, d_f.as<uint8_t>(), d_r.as<int8_t>(), nullptr, H,
                         W, px, fdist ? d_fd.as<float>() : nullptr, idx ? d_i32.as<int32_t>() : nullptr,
                         hand ? d_h.as<float>() : nullptr, nullptr));
  if (idx) {
    DT_TRY(dt_dev_i32_to_i64(c, d_i32.as<int32_t>(), (int64_t)n, d_i64.as<int64_t>()));
    D2H(idx, d_i64, n * 8, c);
  }
  if (fdist) D2H(fdist, d_fd, n * 4, c);
  if (hand) D2H(hand, d_h, n * 4, c);
  return dt_ctx_sync(c);
}

extern "C" int dt_twi(const int64_t *fac, const float *slope_rad, int64_t N, double px, double n_top,
                      float *ti, float *mti) {
  std::lock_guard<std::mutex> lk(g_host_mu);
  dt_ctx *c;
  DT_TRY(host_ctx(&c));
  DT_REQUIRE(N >= 0, "negative size");
  if (N == 0) return DT_OK;
  DT_REQUIRE(fac && slope_rad && ti && mti, "NULL raster");
  size_t n = (size_t)N;
  DevBuf d_f, d_s, d_t, d_m;
  DT_TRY(d_f.alloc(n * 8));
  DT_TRY(d_s.alloc(n * 4));
  DT_TRY(d_t.alloc(n * 4));
  DT_TRY(d_m.alloc(n * 4));
  H2D(d_f, fac, n * 8, c);
  H2D(d_s, slope_rad, n * 4, c);
  DT_TRY(dt_launch_twi_i64(c->stream, d_f.as<int64_t>(), d_s.as<float>(), N, px, n_top, d_t.as<float>(),
                           d_m.as<float>()));
  DT_HIP(hipGetLastError());
  D2H(ti, d_t, n * 4, c);
  D2H(mti, d_m, n * 4, c);
  return dt_ctx_sync(c);
}

extern "C" int dt_river_accumulation(const int64_t *fac, const int64_t *idx, int64_t N, int64_t *out) {
  std::lock_guard<std::mutex> lk(g_host_mu);
  dt_ctx *c;
  DT_TRY(host_ctx(&c));
  DT_REQUIRE(N >= 0, "negative size");
  if (N == 0) return DT_OK;
  DT_REQUIRE(fac && idx && out, "NULL raster");
  size_t n = (size_t)N;
  DevBuf d_f, d_i, d_a;
  DT_TRY(d_f.alloc(n * 8));
  DT_TRY(d_i.alloc(n * 8));
  DT_TRY(d_a.alloc(n * 8));
  H2D(d_f, fac, n * 8, c);
  H2D(d_i, idx, n * 8, c);
  DT_TRY(dt_launch_river_acc_i64(c->stream, d_f.as<int64_t>(), d_i.as<int64_t>(), N, d_a.as<int64_t>()));
  DT_HIP(hipGetLastError());
  D2H(out, d_a, n * 8, c);
  return dt_ctx_sync(c);
}

static int host_gfi(const float *hand, const int64_t *fac, const int64_t *idx, int64_t N, double n_gfi,
                    double b, double size, float *out, int own_cell) {
  std::lock_guard<std::mutex> lk(g_host_mu);
  dt_ctx *c;
  DT_TRY(host_ctx(&c));
  DT_REQUIRE(N >= 0, "negative size");
  if (N == 0) return DT_OK;
  DT_REQUIRE(hand && fac && out, "NULL raster");
  size_t n = (size_t)N;
  DevBuf d_h, d_f, d_i, d_a, d_o;
  DT_TRY(d_h.alloc(n * 4));
  DT_TRY(d_f.alloc(n * 8));
  DT_TRY(d_o.alloc(n * 4));
  H2D(d_h, hand, n * 4, c);
  H2D(d_f, fac, n * 8, c);
  const int64_t *area = d_f.as<int64_t>();
  if (idx) {
    DT_TRY(d_i.alloc(n * 8));
    DT_TRY(d_a.alloc(n * 8));
    H2D(d_i, idx, n * 8, c);
    DT_TRY(dt_launch_river_acc_i64(c->stream, d_f.as<int64_t>(), d_i.as<int64_t>(), N, d_a.as<int64_t>()));
    area = d_a.as<int64_t>();
  }
  DT_TRY(dt_launch_gfi_i64(c->stream, d_h.as<float>(), area, N, n_gfi, b, size, d_o.as<float>(), own_cell));
  DT_HIP(hipGetLastError());
  D2H(out, d_o, n * 4, c);
  return dt_ctx_sync(c);
}
extern "C" int dt_gfi(const float *hand, const int64_t *fac, const int64_t *idx, int64_t N, double n_gfi,
                      double b, double size, float *gfi) {
  DT_REQUIRE(idx || N == 0, "idx is NULL");
  return host_gfi(hand, fac, idx, N, n_gfi, b, size, gfi, 0);
}
extern "C" int dt_lnhlh(const float *hand, const int64_t *fac, int64_t N, double n_gfi, double b, double size,
                        float *out) {
  return host_gfi(hand, fac, nullptr, N, n_gfi, b, size, out, 1);
}
extern "C" int dt_gfi_area(const float *hand, const int64_t *area, int64_t N, double n_gfi, double b,
                           double size, int zero_guard, float *out) {
  return host_gfi(hand, area, nullptr, N, n_gfi, b, size, out, zero_guard ? 1 : 0);
}

extern "C" int dt_hand_f32(const float *dem, const int64_t *idx, int64_t N, float *hand) {
  std::lock_guard<std::mutex> lk(g_host_mu);
  dt_ctx *c;
  DT_TRY(host_ctx(&c));
  DT_REQUIRE(N >= 0, "negative size");
  if (N == 0) return DT_OK;
  DT_REQUIRE(dem && idx && hand, "NULL raster");
  size_t n = (size_t)N;
  DevBuf d_d, d_i, d_h;
  DT_TRY(d_d.alloc(n * 4));
  DT_TRY(d_i.alloc(n * 8));
  DT_TRY(d_h.alloc(n * 4));
  H2D(d_d, dem, n * 4, c);
  H2D(d_i, idx, n * 8, c);
  DT_TRY(dt_launch_hand_i64(c->stream, d_d.as<float>(), d_i.as<int64_t>(), N, d_h.as<float>()));
  DT_HIP(hipGetLastError());
  D2H(hand, d_h, n * 4, c);
  return dt_ctx_sync(c);
}

// ---- heights in float64 (dt_wide.hip): see include/descriptools_hip.h --------------------------------------------
extern "C" int dt_slope_f64(const double *dem, int64_t H, int64_t W, double px, float *slope) {
  std::lock_guard<std::mutex> lk(g_host_mu);
  dt_ctx *c;
  DT_TRY(host_ctx(&c));
  DT_TRY(dt_check_hw(H, W));
  const size_t n = (size_t)H * W;
  if (n == 0) return DT_OK;
  DT_REQUIRE(dem && slope, "NULL raster");
  DevBuf d_d, d_s;
  DT_TRY(d_d.alloc(n * 8));
  DT_TRY(d_s.alloc(n * 4));
  H2D(d_d, dem, n * 8, c);
  DT_TRY(dt_launch_slope_f64(c->stream, d_d.as<double>(), H, W, px, d_s.as<float>()));
  DT_HIP(hipGetLastError());
  D2H(slope, d_s, n * 4, c);
  return dt_ctx_sync(c);
}
extern "C" int dt_hand_f64(const double *dem, const int64_t *idx, int64_t N, double *hand) {
  std::lock_guard<std::mutex> lk(g_host_mu);
  dt_ctx *c;
  DT_TRY(host_ctx(&c));
  DT_REQUIRE(N >= 0, "negative size");
  if (N == 0) return DT_OK;
  DT_REQUIRE(dem && idx && hand, "NULL raster");
  const size_t n = (size_t)N;
  DevBuf d_d, d_i, d_h;
  DT_TRY(d_d.alloc(n * 8));
  DT_TRY(d_i.alloc(n * 8));
  DT_TRY(d_h.alloc(n * 8));
  H2D(d_d, dem, n * 8, c);
  H2D(d_i, idx, n * 8, c);
  DT_TRY(dt_launch_hand_f64(c->stream, d_d.as<double>(), d_i.as<int64_t>(), N, d_h.as<double>()));
  DT_HIP(hipGetLastError());
  D2H(hand, d_h, n * 8, c);
  return dt_ctx_sync(c);
}
extern "C" int dt_downslope_f64(const double *dem, const uint8_t *fdr, int64_t H, int64_t W, double px, double dz,
                                int raw, float *out) {
  std::lock_guard<std::mutex> lk(g_host_mu);
  dt_ctx *c;
  DT_TRY(host_ctx(&c));
  DT_TRY(dt_check_hw(H, W));
  const size_t n = (size_t)H * W;
  if (n == 0) return DT_OK;
  DT_REQUIRE(dem && fdr && out, "NULL raster");
  DevBuf d_d, d_f, d_o;
  DT_TRY(d_d.alloc(n * 8));
  DT_TRY(d_f.alloc(n));
  DT_TRY(d_o.alloc(n * 4));
  H2D(d_d, dem, n * 8, c);
  H2D(d_f, fdr, n, c);
  DT_TRY(dt_launch_downslope_f64(c->stream, d_d.as<double>(), d_f.as<uint8_t>(), H, W, px, dz, raw, d_o.as<float>()));
  DT_HIP(hipGetLastError());
  D2H(out, d_o, n * 4, c);
  return dt_ctx_sync(c);
}
extern "C" int dt_gfi_f64h(const double *hand, const int64_t *fac, const int64_t *idx, int64_t N, double n_gfi,
                           double b, double size, int own_area, float *out) {
  std::lock_guard<std::mutex> lk(g_host_mu);
  dt_ctx *c;
  DT_TRY(host_ctx(&c));
  DT_REQUIRE(N >= 0, "negative size");
  if (N == 0) return DT_OK;
  DT_REQUIRE(hand && fac && out && (own_area || idx), "NULL raster");
  const size_t n = (size_t)N;
  DevBuf d_h, d_f, d_i, d_o;
  DT_TRY(d_h.alloc(n * 8));
  DT_TRY(d_f.alloc(n * 8));
  DT_TRY(d_o.alloc(n * 4));
  H2D(d_h, hand, n * 8, c);
  H2D(d_f, fac, n * 8, c);
  if (!own_area) {
    DT_TRY(d_i.alloc(n * 8));
    H2D(d_i, idx, n * 8, c);
  }
  DT_TRY(dt_launch_gfi_f64h(c->stream, d_h.as<double>(), d_f.as<int64_t>(), own_area ? nullptr : d_i.as<int64_t>(), N,
                            n_gfi, b, size, own_area, d_o.as<float>()));
  DT_HIP(hipGetLastError());
  D2H(out, d_o, n * 4, c);
  return dt_ctx_sync(c);
}

extern "C" int dt_downslope(const float *dem, const uint8_t *fdr, int64_t H, int64_t W, double px, double dz,
                            int raw, float *out) {
  std::lock_guard<std::mutex> lk(g_host_mu);
  dt_ctx *c;
  DT_TRY(host_ctx(&c));
  DT_TRY(dt_check_hw(H, W));
  size_t n = (size_t)H * W;
  if (n == 0) return DT_OK;
  DT_REQUIRE(dem && fdr && out, "NULL raster");
  DevBuf d_dem, d_f, d_o, d_w;
  DT_TRY(d_dem.alloc(n * 4));
  DT_TRY(d_f.alloc(n));
  DT_TRY(d_o.alloc(n * 4));
  H2D(d_dem, dem, n * 4, c);
  H2D(d_f, fdr, n, c);
  // Long walks (real, conditioned rasters walk thousands of moves through flats and along valley floors) are queued;
  // this call is synchronous anyway, so it looks at the queue and allocates the skip tables only for a raster that
  // has enough of them.  The plain kernel when the device has no room for the queue.
  const size_t qb = dt_flow_impl() == 1 ? 0 : (size_t)dt_downslope_queue_workspace(H, W);
  if (qb && d_w.alloc(qb) == DT_OK) {
    DT_TRY(dt_dev_downslope_queue(c, d_dem.as<float>(), d_f.as<uint8_t>(), H, W, px, dz, raw, d_o.as<float>(), d_w.p,
                                  (int64_t)qb));
    int64_t queued = 0;
    DT_TRY(dt_dev_downslope_queued(c, d_w.p, &queued));
    if (queued > 0) {
      DevBuf d_t;
      const size_t tb = (size_t)dt_downslope_tables_workspace(H, W);
      const bool tables = queued >= dt_downslope_tables_threshold(H, W) && d_t.alloc(tb) == DT_OK;
      DT_TRY(dt_dev_downslope_finish(c, d_dem.as<float>(), d_f.as<uint8_t>(), H, W, px, dz, raw, d_o.as<float>(), d_w.p,
                                     (int64_t)qb, tables ? d_t.p : nullptr, tables ? (int64_t)tb : 0));
      D2H(out, d_o, n * 4, c);
      return dt_ctx_sync(c);  // (d_t lives until here)
    }
  } else {
    DT_TRY(dt_dev_downslope(c, d_dem.as<float>(), d_f.as<uint8_t>(), H, W, px, dz, raw, d_o.as<float>()));
  }
  D2H(out, d_o, n * 4, c);
  return dt_ctx_sync(c);
}

extern "C" int dt_confusion_multi(const double *desc, const int8_t *flood, int64_t N, double nodata_value,
                                  const double *th, int nth, int under, int64_t *counts4) {
  std::lock_guard<std::mutex> lk(g_host_mu);
  dt_ctx *c;
  DT_TRY(host_ctx(&c));
  DT_REQUIRE(N >= 0 && th && counts4 && nth >= 1, "bad arguments");
  size_t n = (size_t)N;
  DevBuf d_d, d_f, d_c;
  DT_TRY(d_d.alloc(n * 8));
  DT_TRY(d_f.alloc(n));
  DT_TRY(d_c.alloc(sizeof(int64_t) * 4 * 24));
  if (n) {
    H2D(d_d, desc, n * 8, c);
    H2D(d_f, flood, n, c);
  }
  // more than 24 thresholds: several passes over the resident rasters
  for (int t0 = 0; t0 < nth; t0 += 24) {
    int k = nth - t0 < 24 ? nth - t0 : 24;
    DT_TRY(dt_dev_confusion_multi(c, d_d.as<double>(), d_f.as<int8_t>(), N, nodata_value, th + t0, k, under,
                                  d_c.as<int64_t>()));
    DT_HIP(hipMemcpyAsync(counts4 + (size_t)t0 * 4, d_c.p, sizeof(int64_t) * 4 * k, hipMemcpyDeviceToHost,
                          c->stream));
    DT_HIP(hipStreamSynchronize(c->stream));
  }
  return DT_OK;
}

// ---- evaluation.minMaxScale / binary_map / avaliacao, host tier --------------------------------------
extern "C" int dt_minmax_scale(const void *x, int is_f32, int64_t N, double mn, double mx, double nodata, void *out) {
  std::lock_guard<std::mutex> lk(g_host_mu);
  dt_ctx *c;
  DT_TRY(host_ctx(&c));
  DT_REQUIRE(N >= 0, "negative size");
  if (N == 0) return DT_OK;
  DT_REQUIRE(x && out, "NULL raster");
  const size_t es = is_f32 ? 4 : 8, n = (size_t)N;
  DevBuf d_x, d_o;
  DT_TRY(d_x.alloc(n * es));
  DT_TRY(d_o.alloc(n * es));
  H2D(d_x, x, n * es, c);
  if (is_f32) DT_TRY(dt_launch_minmax_scale_f32f32(c->stream, d_x.as<float>(), N, (float)mn, (float)mx, (float)nodata,
                                                   d_o.as<float>()));
  else DT_TRY(dt_launch_minmax_scale_f64(c->stream, d_x.as<double>(), N, mn, mx, nodata, d_o.as<double>()));
  DT_HIP(hipGetLastError());
  D2H(out, d_o, n * es, c);
  return dt_ctx_sync(c);
}

extern "C" int dt_binary_map(const void *desc, int is_f32, int64_t N, double nodata_value, double threshold, int under,
                             uint8_t *binary) {
  std::lock_guard<std::mutex> lk(g_host_mu);
  dt_ctx *c;
  DT_TRY(host_ctx(&c));
  DT_REQUIRE(N >= 0, "negative size");
  if (N == 0) return DT_OK;
  DT_REQUIRE(desc && binary, "NULL raster");
  const size_t es = is_f32 ? 4 : 8, n = (size_t)N;
  DevBuf d_d, d_f, d_b, d_c;
  DT_TRY(d_d.alloc(n * es));
  DT_TRY(d_f.alloc(n));
  DT_TRY(d_b.alloc(n));
  DT_TRY(d_c.alloc(4 * sizeof(int64_t)));
  H2D(d_d, desc, n * es, c);
  DT_HIP(hipMemsetAsync(d_f.p, 0, n, c->stream));
  if (is_f32) DT_TRY(dt_launch_classify_f32(c->stream, d_d.as<float>(), nullptr, d_f.as<int8_t>(), N, (float)nodata_value,
                                            (float)threshold, under, 0, d_b.as<uint8_t>(), nullptr,
                                            d_c.as<unsigned long long>()));
  else DT_TRY(dt_launch_classify_f64(c->stream, d_d.as<double>(), nullptr, d_f.as<int8_t>(), N, nodata_value, threshold,
                                     under, 0, d_b.as<uint8_t>(), nullptr, d_c.as<unsigned long long>()));
  DT_HIP(hipGetLastError());
  D2H(binary, d_b, n, c);
  return dt_ctx_sync(c);
}

extern "C" int dt_avaliacao(const int32_t *binary, int8_t *flood, int64_t N, int32_t *klass, int64_t *counts4) {
  std::lock_guard<std::mutex> lk(g_host_mu);
  dt_ctx *c;
  DT_TRY(host_ctx(&c));
  DT_REQUIRE(N >= 0 && counts4, "bad arguments");
  const size_t n = (size_t)N;
  DevBuf d_b, d_f, d_k, d_c;
  DT_TRY(d_b.alloc(n * 4));
  DT_TRY(d_f.alloc(n));
  DT_TRY(d_k.alloc(n * 4));
  DT_TRY(d_c.alloc(4 * sizeof(int64_t)));
  if (n) {
    DT_REQUIRE(binary && flood, "NULL raster");
    H2D(d_b, binary, n * 4, c);
    H2D(d_f, flood, n, c);
  }
  DT_TRY(dt_launch_classify_f64(c->stream, nullptr, d_b.as<int32_t>(), d_f.as<int8_t>(), N, 0.0, 0.0, 0, 1, nullptr,
                                klass ? d_k.as<int32_t>() : nullptr, d_c.as<unsigned long long>()));
  DT_HIP(hipGetLastError());
  if (n) {
    D2H(flood, d_f, n, c);  // the benchmark map comes back remapped (evaluation.py:149-150 mutates it)
    if (klass) D2H(klass, d_k, n * 4, c);
  }
  D2H(counts4, d_c, 4 * sizeof(int64_t), c);
  return dt_ctx_sync(c);
}

// conditioned D8, host tier: dem in, D8 codes (flats resolved on the filled surface) and optionally the filled
// surface out
extern "C" int dt_d8_conditioned_f32(const float *dem, int64_t H, int64_t W, double px, uint8_t *fdr, float *filled,
                                     int32_t *info3) {
  std::lock_guard<std::mutex> lk(g_host_mu);
  dt_ctx *c;
  DT_TRY(host_ctx(&c));
  DT_TRY(dt_check_hw(H, W));
  size_t n = (size_t)H * W;
  if (n == 0) return DT_OK;
  DT_REQUIRE(dem && fdr, "NULL raster");
  DevBuf d_dem, d_w, d_f;
  DT_TRY(d_dem.alloc(n * 4));
  DT_TRY(d_w.alloc(n * 4));
  DT_TRY(d_f.alloc(n));
  H2D(d_dem, dem, n * 4, c);
  DT_TRY(dt_dev_condition_d8(c, d_dem.as<float>(), H, W, px, d_w.as<float>(), d_f.as<uint8_t>(), info3));
  D2H(fdr, d_f, n, c);
  if (filled) D2H(filled, d_w, n * 4, c);
  return dt_ctx_sync(c);
}
