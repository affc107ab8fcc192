"""ctypes binding of libdescriptools_hip.so (include/descriptools_hip.h).

The HIP library is the only compute path: if it cannot be loaded (or built) every descriptor call
raises -- there is no CPU fallback.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libdescriptools_hip.so")
_lib = None

c_f32p = C.POINTER(C.c_float)
c_f64p = C.POINTER(C.c_double)
c_u8p = C.POINTER(C.c_uint8)
c_i8p = C.POINTER(C.c_int8)
c_i32p = C.POINTER(C.c_int32)
c_i64p = C.POINTER(C.c_int64)
i64, f64, u32, ci = C.c_int64, C.c_double, C.c_uint32, C.c_int
vp = C.c_void_p

_SIGS = {
    "dt_last_error": (C.c_char_p, []),
    "dt_version": (C.c_char_p, []),
    "dt_device_count": (ci, []),
    "dt_set_flow_impl": (ci, [ci]),
    "dt_debug_set": (ci, [ci, ci]),
    "dt_host_trim": (ci, []),
    "dt_host_alloc": (ci, [i64, C.POINTER(vp)]),
    "dt_host_free": (ci, [vp]),
    "dt_host_f32_to_f64": (ci, [c_f32p, C.POINTER(C.c_double), i64]),
    "dt_ctx_create": (ci, [ci, vp, C.POINTER(vp)]),
    "dt_ctx_destroy": (ci, [vp]),
    "dt_ctx_set_stream": (ci, [vp, vp]),
    "dt_ctx_stream": (vp, [vp]),
    "dt_ctx_sync": (ci, [vp]),
    "dt_ctx_capture_begin": (ci, [vp]),
    "dt_ctx_capture_end": (ci, [vp, C.POINTER(vp)]),
    "dt_graph_launch": (ci, [vp, vp]),
    "dt_graph_destroy": (ci, [vp]),
    "dt_ctx_status": (ci, [vp, c_i32p]),
    "dt_ctx_scratch_bytes": (i64, [vp]),
    # host tier
    "dt_slope_f32": (ci, [c_f32p, i64, i64, f64, c_f32p]),
    "dt_d8_f32": (ci, [c_f32p, i64, i64, f64, c_u8p, c_f32p]),
    "dt_flowacc_u8": (ci, [c_u8p, c_f32p, i64, i64, c_i64p]),
    "dt_d8_conditioned_f32": (ci, [c_f32p, i64, i64, f64, c_u8p, c_f32p, c_i32p]),
    "dt_dev_condition_d8": (ci, [vp, vp, i64, i64, f64, vp, vp, c_i32p]),
    "dt_dev_condition_d8_async": (ci, [vp, vp, i64, i64, f64, vp, vp, ci]),
    "dt_dev_condition_stage_w": (ci, [vp, vp, ci, ci, vp, vp, vp, vp, vp]),
    "dt_dev_condition_stage_m_w": (ci, [vp, vp, ci, ci, vp, vp, vp, vp, vp, vp]),
    "dt_flowhand": (ci, [c_f32p, c_u8p, c_i8p, i64, i64, f64, c_f32p, c_i64p, c_f32p]),
    "dt_hand_f32": (ci, [c_f32p, c_i64p, i64, c_f32p]),
    "dt_twi": (ci, [c_i64p, c_f32p, i64, f64, f64, c_f32p, c_f32p]),
    "dt_river_accumulation": (ci, [c_i64p, c_i64p, i64, c_i64p]),
    "dt_gfi_area": (ci, [c_f32p, c_i64p, i64, f64, f64, f64, ci, c_f32p]),
    "dt_gfi": (ci, [c_f32p, c_i64p, c_i64p, i64, f64, f64, f64, c_f32p]),
    "dt_lnhlh": (ci, [c_f32p, c_i64p, i64, f64, f64, f64, c_f32p]),
    "dt_downslope": (ci, [c_f32p, c_u8p, i64, i64, f64, f64, ci, c_f32p]),
    # heights in float64 (a DEM / HAND that float32 cannot hold)
    "dt_slope_f64": (ci, [c_f64p, i64, i64, f64, c_f32p]),
    "dt_hand_f64": (ci, [c_f64p, c_i64p, i64, c_f64p]),
    "dt_downslope_f64": (ci, [c_f64p, c_u8p, i64, i64, f64, f64, ci, c_f32p]),
    "dt_gfi_f64h": (ci, [c_f64p, c_i64p, c_i64p, i64, f64, f64, f64, ci, c_f32p]),
    "dt_confusion_multi": (ci, [c_f64p, c_i8p, i64, f64, c_f64p, ci, ci, c_i64p]),
    "dt_synth_dem": (ci, [u32, i64, i64, i64, i64, i64, i64, ci, c_f32p]),
    # device tier
    "dt_dev_malloc": (ci, [vp, i64, C.POINTER(vp)]),
    "dt_dev_free": (ci, [vp, vp]),
    "dt_dev_h2d": (ci, [vp, vp, vp, i64]),
    "dt_dev_d2h": (ci, [vp, vp, vp, i64]),
    "dt_dev_d2h_async": (ci, [vp, vp, vp, i64]),
    "dt_dev_synth_dem": (ci, [vp, u32, i64, i64, i64, i64, i64, i64, ci, vp]),
    "dt_dev_slope_d8": (ci, [vp, vp, i64, i64, f64, vp, vp, vp]),
    "dt_dev_slope_twi": (ci, [vp, vp, vp, i64, i64, f64, f64, vp, vp, vp, vp]),
    "dt_dev_flowacc": (ci, [vp, vp, vp, i64, i64, vp]),
    "dt_dev_river_mask": (ci, [vp, vp, i64, i64, vp]),
    "dt_dev_flowacc_river": (ci, [vp, vp, vp, i64, i64, i64, vp, vp]),
    "dt_dev_gfi_lnhlh": (ci, [vp, vp, vp, vp, i64, f64, f64, f64, vp, vp]),
    "dt_dev_flowhand": (ci, [vp, vp, vp, vp, vp, i64, i64, f64, vp, vp, vp, vp]),
    "dt_dev_twi": (ci, [vp, vp, vp, i64, f64, f64, vp, vp]),
    "dt_dev_gfi": (ci, [vp, vp, vp, i64, f64, f64, f64, vp]),
    "dt_dev_lnhlh": (ci, [vp, vp, vp, i64, f64, f64, f64, vp]),
    "dt_dev_downslope": (ci, [vp, vp, vp, i64, i64, f64, f64, ci, vp]),
    "dt_downslope_lift_workspace": (i64, [i64, i64]),
    "dt_dev_downslope_lift": (ci, [vp, vp, vp, i64, i64, f64, f64, ci, vp, vp, i64]),
    "dt_downslope_queue_workspace": (i64, [i64, i64]),
    "dt_downslope_tables_workspace": (i64, [i64, i64]),
    "dt_downslope_tables_threshold": (i64, [i64, i64]),
    "dt_dev_downslope_queue": (ci, [vp, vp, vp, i64, i64, f64, f64, ci, vp, vp, i64]),
    "dt_dev_downslope_queued": (ci, [vp, vp, C.POINTER(i64)]),
    "dt_dev_downslope_finish": (ci, [vp, vp, vp, i64, i64, f64, f64, ci, vp, vp, i64, vp, i64]),
    "dt_dev_confusion_multi": (ci, [vp, vp, vp, i64, f64, c_f64p, ci, ci, vp]),
    "dt_perim_cells": (i64, [i64, i64]),
    "dt_dev_slope_d8_w": (ci, [vp, vp, vp, f64, vp, vp, vp]),
    "dt_dev_slope_twi_w": (ci, [vp, vp, vp, vp, f64, f64, vp, vp, vp, vp]),
    "dt_dev_downslope_w": (ci, [vp, vp, vp, vp, f64, f64, ci, vp, vp]),
    "dt_downslope_lift_workspace_w": (i64, [vp]),
    "dt_dev_downslope_lift_w": (ci, [vp, vp, vp, vp, f64, f64, ci, vp, vp, vp, i64]),
    "dt_dev_downslope_emit_w": (ci, [vp, vp, vp, vp, f64, f64, ci, vp, vp, vp, i64, vp, i64]),
    "dt_dev_downslope_walk_w": (ci, [vp, vp, vp, vp, f64, f64, i64, vp, vp, i64]),
    "dt_dev_downslope_walk_route_w": (ci, [vp, vp, vp, vp, f64, f64, i64, vp, vp, i64, vp, vp, C.c_int32, vp, C.c_int32, vp, vp, vp]),
    "dt_dev_downslope_walk_seed_w": (ci, [vp, vp, vp, i64, vp, vp, vp]),
    "dt_dev_flowacc_local_w": (ci, [vp, vp, vp, vp, vp, vp, vp]),
    "dt_dev_flowacc_finish_w": (ci, [vp, vp, vp, vp, vp, i64, vp, vp]),
    "dt_dev_flowhand_local_w": (ci, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "dt_dev_flowhand_finish_w": (ci, [vp, vp, vp, vp, vp, vp, f64, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "dt_dev_unique_extremes_f32": (ci, [vp, vp, i64, vp]),
    "dt_dev_minmax_scale_f32": (ci, [vp, vp, i64, C.c_float, C.c_float, C.c_float, vp]),
    "dt_dev_membench_copy": (ci, [vp, vp, vp, i64, ci]),
    "dt_dev_membench_mix": (ci, [vp, vp, vp, vp, vp, vp, i64, ci, ci, ci]),
    "dt_dev_membench_mix_timed": (ci, [vp, vp, vp, vp, vp, vp, i64, ci, ci, ci, ci, c_f64p]),
    "dt_dev_mem_info": (ci, [vp, c_i64p, c_i64p]),
    "dt_dev_minmax_scale_f32_f64": (ci, [vp, vp, i64, f64, f64, f64, vp]),
    "dt_dev_classify": (ci, [vp, vp, vp, i64, f64, f64, ci, ci, vp, vp, vp]),
    "dt_minmax_scale": (ci, [vp, ci, i64, f64, f64, f64, vp]),
    "dt_binary_map": (ci, [vp, ci, i64, f64, f64, ci, c_u8p]),
    "dt_avaliacao": (ci, [c_i32p, c_i8p, i64, c_i32p, c_i64p]),
    "dt_dev_flowhand_gfi": (ci, [vp, vp, vp, vp, vp, i64, i64, f64, f64, f64, vp, vp, vp, vp, vp, vp]),
    "dt_dev_flowhand_gfi_finish_w": (ci, [vp, vp, vp, vp, vp, vp, f64, f64, f64, vp, vp, vp, vp, vp,
                                          vp, vp, vp, vp, vp, vp, vp, vp]),
    "dt_ctx_set_priority": (ci, [vp, ci]),
    "dt_ctx_fork": (ci, [vp, vp]),
    "dt_ctx_join": (ci, [vp, vp]),
    "dt_dev_rank_solve_flowacc": (ci, [vp, ci, ci, c_i64p, c_i64p, i64, vp, i64, c_i64p, ci, i64, vp]),
    "dt_dev_rank_solve_flowhand": (ci, [vp, ci, ci, c_i64p, c_i64p, i64, vp, i64, c_i64p, ci, i64, vp, vp, vp, vp,
                                        vp, vp]),
    # int64 accumulation rasters (multi-rank rasters of >= 2^31 cells)
    "dt_dev_flowacc_finish_w_a64": (ci, [vp, vp, vp, vp, vp, i64, vp, vp]),
    "dt_dev_slope_twi_w_a64": (ci, [vp, vp, vp, vp, f64, f64, vp, vp, vp, vp]),
    "dt_dev_flowhand_local_w_a64": (ci, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "dt_dev_flowhand_finish_w_a64": (ci, [vp, vp, vp, vp, vp, vp, f64, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "dt_dev_flowhand_gfi_finish_w_a64": (ci, [vp, vp, vp, vp, vp, vp, f64, f64, f64, vp, vp, vp, vp, vp,
                                              vp, vp, vp, vp, vp, vp, vp, vp]),
    "dt_dev_gfi_lnhlh_a64": (ci, [vp, vp, vp, vp, i64, f64, f64, f64, vp, vp]),
    "dt_dev_flowacc_river_flowhand_local": (ci, [vp, vp, vp, i64, i64, i64, vp, vp]),
    "dt_dev_flowacc_river_flowhand_local_m": (ci, [vp, vp, vp, vp, i64, i64, i64, vp, vp]),
    "dt_dev_slope_d8_m": (ci, [vp, vp, i64, i64, f64, vp, vp]),
    "dt_nodata_mask_bytes": (i64, [i64, i64]),
    "dt_dev_flowacc_finish_flowhand_local_w": (ci, [vp, vp, vp, vp, vp, i64, vp, vp, vp, vp, vp, vp, vp, vp]),
    "dt_dev_flowacc_finish_flowhand_local_w_a64": (ci, [vp, vp, vp, vp, vp, i64, vp, vp, vp, vp, vp, vp, vp, vp]),
    "dt_dev_i32_to_i64": (ci, [vp, vp, i64, vp]),
    "dt_dev_i64_to_i32": (ci, [vp, vp, i64, vp]),
}


class Window(C.Structure):
    """dt_window: one rank's core window of a global raster (see include/descriptools_hip.h)."""
    _fields_ = [(n, C.c_int64) for n in ("H", "W", "ld", "gy0", "gx0", "Hg", "Wg", "halo")]


def exported_symbols():
    return sorted(_SIGS)


def _preload_torch_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64.  Two HIP runtimes in one process do not
    share the GPU (whichever initialises second sees "no HIP GPUs"), so if torch is installed but not
    imported yet, load ITS runtime first: our library then binds to the same one, whatever the
    import order.  Without torch the system runtime (/opt/rocm) is used."""
    import sys
    if "torch" in sys.modules:
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:  # pragma: no cover - best effort
        pass


def lib():
    """Load (building in-tree if the .so is missing) the HIP library; raises if impossible."""
    global _lib
    if _lib is not None:
        return _lib
    _preload_torch_hip_runtime()
    if not os.path.exists(_SO):
        from . import build as _build  # hipcc cross-compiles; raises CalledProcessError on failure
        _build.build()
    try:
        L = C.CDLL(_SO)
    except OSError as e:  # pragma: no cover
        raise RuntimeError("descriptools_amd: cannot load %s (%s); the HIP library is required, "
                           "there is no CPU fallback" % (_SO, e))
    for name, (res, args) in _SIGS.items():
        fn = getattr(L, name)  # AttributeError here = header / library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


DT_ENOMEM = -3


def check(rc):
    """raise for a non-zero return code of the C ABI: MemoryError for DT_ENOMEM (a full device is something a caller
    may want to handle: placement.assign does), RuntimeError for everything else"""
    if rc != 0:
        msg = lib().dt_last_error().decode("utf-8", "replace")
        if rc == DT_ENOMEM:
            raise MemoryError("descriptools_hip error %d: %s" % (rc, msg))
        raise RuntimeError("descriptools_hip error %d: %s" % (rc, msg))


def ptr(a, ct):
    return a.ctypes.data_as(ct) if a is not None else None


def as_c(a, dtype):
    """C-contiguous array of `dtype` (no copy when already so)."""
    return np.ascontiguousarray(a, dtype=dtype)


_EXACT_IN_F32 = (np.float32, np.float16, np.int8, np.uint8, np.int16, np.uint16, np.bool_)


def _first_inexact(a, d32):
    """flat index of the first element of `a` that float32 cannot hold (d32 = a as float32), or -1; blockwise: no
    second full-size temporary"""
    flat, f32 = a.reshape(-1), d32.reshape(-1)
    step = 1 << 24
    for i in range(0, flat.size, step):
        blk = flat[i:i + step]
        back = f32[i:i + step].astype(a.dtype)
        same = (back == blk) | ((back != back) & (blk != blk)) if a.dtype.kind == "f" else (back == blk)
        if not same.all():
            return i + int(np.argmin(same))
    return -1


def heights(raster, what="DEM"):
    """A DEM / HAND at the boundary -> (array, wide).

    The reference takes height differences in the raster's OWN dtype (slope.py:244-258 under Numba typing,
    flowhand.py:436-438 `dem - dem[indices]`, downslope.py:468).  The tuned kernels take them in float32 -- the same
    arithmetic exactly when every height is a float32 value: int8 / int16 / float16 / float32 rasters always, wider
    dtypes (float64, int32, int64) when their values happen to be representable (a float64 array holding float32
    values, integer heights below 2^24).  Those come back as (float32 array, False).  Anything else -- a genuinely
    float64 DEM, integer heights beyond 2^24 -- comes back as (float64 array, True) and the caller uses the float64
    entry points (dt_slope_f64, dt_hand_f64, dt_downslope_f64, dt_gfi_f64h: csrc/dt_wide.hip), which evaluate the
    reference's expressions on the heights as they are.  Integer heights beyond 2^53 have no exact float64 either:
    ValueError.  DT_ALLOW_DEM_ROUNDING=1 rounds everything to float32 knowingly (the fast kernels)."""
    a = np.asarray(raster)
    d32 = np.ascontiguousarray(a, dtype=np.float32)
    if a.dtype.type in _EXACT_IN_F32 or os.environ.get("DT_ALLOW_DEM_ROUNDING") == "1":
        return d32, False
    if _first_inexact(a, d32) < 0:
        return d32, False
    d64 = np.ascontiguousarray(a, dtype=np.float64)
    if a.dtype.kind in "iu" and a.dtype.itemsize == 8:
        back = d64.astype(a.dtype)
        if not (back == a).all():
            k = int(np.argmin((back == a).reshape(-1)))
            raise ValueError("%s of dtype %s holds %r (flat index %d), which float64 cannot represent: the reference "
                             "would take differences of such heights in int64" % (what, a.dtype, a.reshape(-1)[k].item(), k))
    return d64, True


def dem_f32(dem, what="DEM"):
    """DEM / HAND for an entry point that exists in float32 only (the resident chain, the net-new D8 / conditioning):
    the float32 array when every height is a float32 value (see heights()), ValueError otherwise -- the drop-in
    descriptor functions (sloper, flow_hand_index, hand_calculator, downsloper, gfi_calculator, ln_hl_H_calculator)
    accept such a raster and compute in float64; set DT_ALLOW_DEM_ROUNDING=1 to round to float32 knowingly."""
    a = np.asarray(dem)
    d32 = np.ascontiguousarray(a, dtype=np.float32)
    if a.dtype.type in _EXACT_IN_F32 or os.environ.get("DT_ALLOW_DEM_ROUNDING") == "1":
        return d32
    k = _first_inexact(a, d32)
    if k >= 0:
        raise ValueError(
            "%s of dtype %s is not exactly representable in float32 (first at flat index %d: %r -> %r): the "
            "reference computes height differences in the raster's own dtype, this entry point in float32.  Use the "
            "drop-in descriptor functions (they take such a raster in float64), pass float32-exact heights, or set "
            "DT_ALLOW_DEM_ROUNDING=1 to accept the rounding"
            % (what, a.dtype, k, a.reshape(-1)[k].item(), d32.reshape(-1)[k].item()))
    return d32
