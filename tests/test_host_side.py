"""CPU (not gpu): host-side helpers that need no device -- the threaded float32 -> float64 container fill, and the
placement search's logic (spacers between candidates, neighbours of a block that adds a class) against the
allocation pattern measured on an MI355X box (profiles/r3/placement_classes.txt), with a stand-in classifier."""
import ctypes as C

import numpy as np
import pytest


@pytest.mark.parametrize("n", [0, 1, 1000, (1 << 22) + 12345])
def test_threaded_f32_to_f64_equals_astype(n):
    from descriptools_amd import _lib
    rng = np.random.default_rng(n)
    a = (rng.standard_normal(n) * 1e3).astype(np.float32)
    if n > 10:
        a[:4] = [np.nan, np.inf, -np.inf, -100.0]
    out = np.full(n, 7.0, np.float64)
    _lib.check(_lib.lib().dt_host_f32_to_f64(a.ctypes.data_as(C.POINTER(C.c_float)),
                                             out.ctypes.data_as(C.POINTER(C.c_double)), n))
    assert np.array_equal(out, a.astype(np.float64), equal_nan=True)


PATTERN = ("ABBAABBBBBBBBBBBBAAAAAAAABBBBBBBBBBBBBBBBCCCCCCCCCCCCCCCCAAAAAAAAAAAAAAAABBBBBBBBBBBBBBBBBBBBBBBBBBBBBBBBAAAAAAAAAAAAAAA"
           * 2)  # 1-GiB allocations of one process in order, by conflict class


class _ByPosition:
    """the class of a block is the pattern's letter at its position in allocation order (pointer = position)"""

    def __init__(self, ctx, nbytes):
        self.reps, self.single_ms, self.names = [], 0.167, {}

    def usable(self):
        return True

    def label(self, p):
        c = PATTERN[int(p)]
        if c not in self.names:
            self.names[c] = len(self.names)
            self.reps.append(int(p))
        return self.names[c]


@pytest.mark.parametrize("start", [0, 5, 17, 25, 41, 57, 73, 80, 100])
def test_placement_search_finds_other_classes_behind_long_runs(monkeypatch, start):
    from descriptools_amd import chain, placement
    monkeypatch.setattr(placement, "WriteClassifier", _ByPosition)
    pos = [start + 12]
    released, spacers = [], []

    def extra_alloc():
        pos[0] += 1
        return pos[0] - 1

    def spacer_alloc(nbytes):
        h = (pos[0], nbytes >> 30)
        pos[0] += nbytes >> 30
        spacers.append(h)
        return h
    groups = [list(g) for g in chain.WRITE_GROUPS]
    roles, info = placement.assign(None, 1 << 30, list(range(start, start + 12)), groups, extra_alloc, released.append,
                                   spacer_alloc=spacer_alloc, spacer_release=spacers.remove)
    assert not spacers, "every spacer is released"
    assert len(set(roles.values())) == sum(len(g) for g in groups)           # one block per role, no block twice
    assert not (set(released) & set(roles.values()))                          # nothing kept is released
    assert info["tuned"] and info["candidates_tried"] <= 24 and info["spacer_GiB"] <= 160
    # what the search is for: no group of rasters written together sits in one class
    for g in groups:
        if len(g) > 1:
            assert len({info["classes"][r] for r in g}) >= 2, (start, g, info)


def test_placement_search_without_spacers_is_the_old_back_to_back_search(monkeypatch):
    from descriptools_amd import chain, placement
    monkeypatch.setattr(placement, "WriteClassifier", _ByPosition)
    pos = [73 + 12]

    def extra_alloc():
        pos[0] += 1
        return pos[0] - 1
    roles, info = placement.assign(None, 1 << 30, list(range(73, 85)), [list(g) for g in chain.WRITE_GROUPS], extra_alloc,
                                   lambda q: None, budget=16)
    assert info["spacer_GiB"] == 0 and info["candidates_tried"] == 16 and info["n_classes"] == 1  # the run is 32 long


class _MixedFirstBlock:
    """conflict-based stand-in: two blocks conflict when their letters are equal -- and block 0 (a block that straddles
    two runs) conflicts with everybody"""
    letters = "XBBBBBBBBBBBCCCCAAAABBBBCCCCAAAAAAAA"

    def __init__(self, ctx, nbytes):
        self.reps, self.single_ms, self.ratios = [], 0.167, []

    def usable(self):
        return True

    def label(self, p):
        p = int(p)
        for k, r in enumerate(self.reps):
            conflict = r == 0 or p == 0 or self.letters[r] == self.letters[p]
            self.ratios.append(2.07 if conflict else 1.8)
            if conflict:
                return k
        self.reps.append(p)
        return len(self.reps) - 1


def test_a_representative_that_conflicts_with_everybody_is_replaced(monkeypatch):
    from descriptools_amd import chain, placement
    monkeypatch.setattr(placement, "WriteClassifier", _MixedFirstBlock)
    pos = [12]

    def extra_alloc():
        pos[0] += 1
        return pos[0] - 1
    groups = [list(g) for g in chain.WRITE_GROUPS]
    roles, info = placement.assign(None, 1 << 30, list(range(12)), groups, extra_alloc, lambda q: None,
                                   spacer_alloc=None, spacer_release=None)
    assert info["relabelled"] and info["n_classes"] >= 2
    for g in groups:
        if len(g) > 1:
            assert len({info["classes"][r] for r in g}) >= 2, (g, info)


def test_bench_gpus8_defaults_to_config5():
    """`bench.py --gpus 8` (what the driver's SCALE run launches) measures BASELINE.json configs[4]: the 65536^2 raster
    as 2 x 4 rank tiles of 32768 x 16384; an explicit --size keeps the weak-scaling series; N = 4 is configs[3]"""
    import importlib
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    bench = importlib.import_module("bench")
    size, glob = bench.default_workload(8, None, None)
    layout, series = bench.tiled_layout(8, size, glob)
    assert (layout.ty, layout.tx) == (2, 4) and layout.shape(0) == (32768, 16384)
    assert (layout.Hg, layout.Wg) == (65536, 65536) and "configs[4]" in series
    assert layout.Hg * layout.Wg > 2 ** 31  # -> int64 accumulation and river index (RankTile defaults)
    size, glob = bench.default_workload(8, 16384, None)
    layout, series = bench.tiled_layout(8, size, glob)
    assert layout.shape(0) == (16384, 16384) and (layout.Hg, layout.Wg) == (32768, 65536) and "weak" in series
    size, glob = bench.default_workload(4, None, None)
    layout, _ = bench.tiled_layout(4, size, glob)
    assert (layout.Hg, layout.Wg) == (32768, 32768) and layout.shape(3) == (16384, 16384)
    assert bench.default_workload(1, None, None) == (16384, None)
    layout, _ = bench.tiled_layout(2, 0, "1024x2048")
    assert (layout.ty, layout.tx) == (1, 2) and layout.shape(1) == (1024, 1024)


def test_pinned_pool_size_classes_and_cap(monkeypatch):
    """a 1.01 GiB raster must not lock 2 GiB, and the page-locked bytes in callers' hands are capped
    (pure bookkeeping: no allocation happens in this test)"""
    from descriptools_amd import device
    k = device.PinnedPool._klass
    assert k(100) == 4096 and k((1 << 26)) == 1 << 26
    big = int(1.01 * (1 << 30))
    assert big <= k(big) < big + (2 << 20) and k(big) % (2 << 20) == 0
    pool = device.PinnedPool()
    pool.max_live = 1 << 20
    pool._live = 1 << 20       # the cap is reached: the next array is pageable numpy, no library call
    a = pool.empty((1 << 18,), "float32")
    assert a.shape == (1 << 18,) and pool._live == 1 << 20
