"""CPU (not gpu): the C-ABI library builds, loads without a GPU, and exports every symbol that
include/descriptools_hip.h declares; the Python binding table matches the header; the product path
fails loudly (no CPU fallback) when no GPU is visible."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "descriptools_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(dt_[a-z0-9_]+)\s*\(", txt)))


def test_library_builds_and_exports_header_symbols():
    from descriptools_amd import build
    so = build.build()
    lib = ctypes.CDLL(so)
    syms = header_symbols()
    assert len(syms) > 35
    for s in syms:
        assert hasattr(lib, s), "header declares %s but the library does not export it" % s


def test_binding_table_matches_header():
    from descriptools_amd import _lib
    bound = set(_lib.exported_symbols())
    declared = set(header_symbols())
    assert bound <= declared, "bound but not declared: %s" % sorted(bound - declared)
    # every host-tier entry point of the header is bound
    missing = {s for s in declared - bound if not s.startswith("dt_dev_") and not s.startswith("dt_ctx_")}
    assert not missing, "declared but not bound: %s" % sorted(missing)
    L = _lib.lib()
    assert L.dt_version().startswith(b"descriptools_hip")


def test_no_cpu_fallback_without_gpu():
    from descriptools_amd import _lib, slope, flowhand, topoindexes
    if _lib.lib().dt_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(RuntimeError, match="no HIP device"):
        slope.sloper(np.zeros((8, 8), np.float32), 10.0)
    with pytest.raises(RuntimeError):
        flowhand.flow_hand_index(np.zeros((8, 8), np.int16), np.ones((8, 8), np.uint8),
                                 np.zeros((8, 8), np.int8), 10.0)
    with pytest.raises(RuntimeError):
        topoindexes.topographic_index(np.ones((8, 8), np.int64), np.zeros((8, 8), np.float32), 10.0, 0.1)


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under descriptools_amd/ may reference it."""
    pkg = os.path.join(ROOT, "descriptools_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, f
                assert "dt_oracle" not in txt or f in ("dt_kernels.hip",), f


def test_helpers_divisor_matches_golden():
    from conftest import golden
    from descriptools_amd import helpers
    g = golden("edge")
    r, c = helpers.divisor(100, 37, 3, 2)
    assert np.array_equal(r, g["div_a"]) and np.array_equal(c, g["div_b"])
    r, c = helpers.divisor(10, 10, 0, 0)
    assert r.size == 0 and c.size == 0


def test_evaluation_has_no_cpu_path_either():
    """minMaxScale / binary_map / avaliacao are kernels behind the C ABI like every other descriptor (their
    goldens are checked on the GPU, tests/test_gpu_parity.py::test_golden_eval); the pure index formulas are not."""
    from descriptools_amd import _lib, evaluation
    assert evaluation.correctness(np.array([5, 1, 2, 6])) == 6 / 8
    assert evaluation.fit(np.array([5, 1, 2, 6])) == 6 / 9
    if _lib.lib().dt_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(RuntimeError):
        evaluation.minMaxScale(np.ones((4, 4), np.float32), 0, 1, -100)
    with pytest.raises(RuntimeError):
        evaluation.binary_map(np.ones((4, 4)), 0.5, 'under')
    with pytest.raises(RuntimeError):
        evaluation.avaliacao(np.ones((4, 4), np.int64), np.ones((4, 4), np.int8))


def test_python_api_mirrors_reference_names():
    """every public (non-kernel) function name of the reference's modules exists with the
    same positional parameters in descriptools_amd."""
    import inspect
    import importlib
    expected = {
        "slope": {"sloper": ["dem", "px", "division_column", "division_row"],
                  "slope_cpu": ["dem", "px", "extra", "blocks", "threads"],
                  "slope_sequential_jit": None, "slope_sequential": None},
        "flowhand": {"flow_hand_index": ["dem_raster", "flow_direction_matrix", "river_matrix", "px",
                                         "division_column", "division_row"],
                     "hand_calculator": ["dem", "indices"],
                     "index_calculator": ["river_indices", "row_start", "column_start", "column_size"],
                     "flow_distance_index_cpu": ["dem", "flow_direction", "river_matrix", "px",
                                                 "boundary_distance", "boundary_index", "out", "row_start",
                                                 "col_start", "matrix_columns", "blocks", "threads"],
                     "flow_distance_indexes_sequential": None, "fdist_indexes_sequential_jit": None},
        "topoindexes": {"topographic_index": ["flow_accumulation", "slope", "px", "n_top", "div_col", "div_row"],
                        "topographic_index_cpu": ["flow_accumulation", "slope", "px", "expoent", "blocks",
                                                  "threads"],
                        "topographic_index_sequential": None, "topographic_index_sequential_jit": None,
                        "modified_topographic_index_sequential": None,
                        "modified_topographic_index_sequential_jit": None},
        "gfi": {"gfi_calculator": ["hand", "flow_accumulation", "indices", "n_gfi", "scale_factor", "size",
                                   "division_column", "division_row"],
                "river_accumulation": ["flow_accumulation", "indices"],
                "geomorphic_flood_index_cpu": ["hand", "river_flow_accumulation", "expoent", "scale_factor",
                                               "size", "blocks", "threads"],
                "ln_hl_H_calculator": ["hand", "flow_accumulation", "n_gfi", "scale_factor", "size",
                                       "division_column", "division_row"],
                "ln_hl_H_cpu": ["hand", "flow_accumulation", "expoent", "scale_factor", "size", "blocks",
                                "threads"],
                "geomorphic_flood_index_sequential": None, "geomorphic_flood_index_sequential_jit": None,
                "ln_hl_H_sequential": None, "ln_hl_H_sequential_jit": None},
        "downslope": {"downsloper": ["dem", "flow_direction", "px", "elevation_difference", "column_division",
                                     "row_division"],
                      "downslope_cpu": ["dem", "flow_direction", "px", "elevation_difference", "blocks",
                                        "threads"],
                      "downslope_sequential_jit": None, "downslope_sequential": None},
        "evaluation": {"minMaxScale": ["mat", "mn", "mx", "nodata"],
                       "calibration": ["descriptor_matrix", "comparison_matrix", "under"],
                       "binary_map": ["descriptor_matrix", "threshold", "under"],
                       "avaliacao": ["descriptor_flood_map", "comparison_flood_map"],
                       "correctness": ["count"], "fit": ["count"]},
        "helpers": {"divisor": ["row_length", "column_length", "row_division", "column_division"]},
    }
    for mod, funcs in expected.items():
        m = importlib.import_module("descriptools_amd." + mod)
        for name, params in funcs.items():
            assert hasattr(m, name), "%s.%s missing" % (mod, name)
            if params is not None:
                got = list(inspect.signature(getattr(m, name)).parameters)
                assert got == params, "%s.%s%s != %s" % (mod, name, got, params)


def test_reference_import_block_runs_unchanged():
    """Callers of the reference import `descriptools.<module>` (Example/example.py:11-16): the alias package at
    the repository root serves those names from descriptools_amd, function for function."""
    ns = {}
    exec("import descriptools.topoindexes as topoindexes\nimport descriptools.downslope as downslope\n"
         "import descriptools.slope as slope\nimport descriptools.flowhand as flowhand\n"
         "import descriptools.gfi as gfi\nimport descriptools.evaluation as evaluation\n", ns)
    import descriptools
    import descriptools.helpers
    import descriptools_amd.slope, descriptools_amd.evaluation, descriptools_amd.flowhand  # noqa: E401
    assert open(descriptools.__file__).read() == "", "descriptools/__init__.py is empty like the reference's"
    assert ns["slope"].sloper is descriptools_amd.slope.sloper
    assert ns["evaluation"].calibration is descriptools_amd.evaluation.calibration
    assert ns["flowhand"].flow_hand_index is descriptools_amd.flowhand.flow_hand_index
    assert descriptools.helpers.divisor(10, 10, 1, 1)[0][0] == 5
