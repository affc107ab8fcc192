"""GPU (-m gpu): placement-aware assignment of the chain's rasters (descriptools_amd/placement.py).  Which block of
device memory serves which raster must not change a single bit of the results; the assignment itself is checked
for what it promises: no group of rasters written by one kernel in a single conflict class when the blocks at
hand allow it."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_spread_prefers_distinct_classes():
    from descriptools_amd import chain, placement
    labels = {i: (0 if i == 0 else (2 if i in (10, 11) else 1)) for i in range(12)}
    roles, left = placement.spread(labels, [list(g) for g in chain.WRITE_GROUPS])
    assert not left and len(set(roles.values())) == 12
    assert {labels[roles[r]] for r in ("slope", "ti", "mti")} == {0, 1, 2}
    assert len({labels[roles[r]] for r in ("fdist", "idx", "hand", "gfi", "lnhlh")}) >= 2
    # one class only: still a valid assignment
    roles, left = placement.spread({i: 0 for i in range(12)}, [list(g) for g in chain.WRITE_GROUPS])
    assert len(roles) == 12 and not left


def test_tuned_chain_gives_identical_rasters():
    import torch
    from descriptools_amd import _lib, chain
    from descriptools_amd.device import Context
    n, px = 4096, 10.0   # 64 MiB rasters: the smallest size the tuning measures
    ctx = Context()
    L = _lib.lib()
    dem = ctx.empty((n, n), np.float32)
    _lib.check(L.dt_dev_synth_dem(ctx.h, 3, n, n, 0, 0, n, n, 1, dem.ptr))
    outs = []
    for tune in (False, True):
        ch = chain.Chain(n, n, ctx=ctx, px=px, tune_placement=tune)
        ch.run(dem.ptr)
        ctx.sync()
        outs.append({k: ch.buf[k].to_host() for k, _ in chain.OUTPUTS})
        info = ch.placement
        ch.free()
    assert info["tuned"] and set(info["classes"]) == {r for g in chain.WRITE_GROUPS for r in g}
    assert info["n_classes"] >= 1 and info["blocks"] >= 12
    for k, dt in chain.OUTPUTS:
        assert outs[0][k].dtype == np.dtype(dt) == outs[1][k].dtype
        assert np.array_equal(outs[0][k], outs[1][k], equal_nan=True), k
    dem.free()
    ctx.close()
    torch.cuda.empty_cache()


class _RunOfOneClass:
    """stands in for placement.WriteClassifier: the first `run` distinct blocks it sees are class 0 (a long run of one
    class in allocation order, as on some boxes: profiles/r3/placement_classes.txt), later ones alternate 1 / 2"""
    run = 14   # the chain's twelve blocks and the first two candidates

    def __init__(self, ctx, nbytes):
        self.seen, self.reps, self.single_ms = {}, [], 0.1

    def usable(self):
        return True

    def label(self, p):
        p = int(p)
        if p not in self.seen:
            i = len(self.seen)
            self.seen[p] = 0 if i < self.run else 1 + (i % 2)
            if self.seen[p] not in [self.seen[q] for q in self.reps]:
                self.reps.append(p)
        return self.seen[p]


def test_search_with_spacers_when_the_first_blocks_are_all_alike(monkeypatch):
    """12 rasters + 24 back-to-back candidates of one class used to leave the chain untuned; with spacers between the
    candidates the search moves on, finds the other classes and releases every spacer and unused candidate -- for the
    chain's own allocator, for a caller's alloc / release pair, and for a rank tile's torch tensors"""
    import torch
    from descriptools_amd import _lib, chain, placement, tiling
    from descriptools_amd.device import Context
    monkeypatch.setattr(placement, "WriteClassifier", _RunOfOneClass)
    n = 2048
    ctx = Context()
    L = _lib.lib()
    dem = ctx.empty((n, n), np.float32)
    _lib.check(L.dt_dev_synth_dem(ctx.h, 3, n, n, 0, 0, n, n, 1, dem.ptr))
    ref = chain.Chain(n, n, ctx=ctx, tune_placement=False)
    ref.run(dem.ptr)
    ctx.sync()
    want = {k: ref.buf[k].to_host() for k, _ in chain.OUTPUTS}
    ref.free()
    # the library's default works with the blocks at hand: nothing else is allocated
    ch = chain.Chain(n, n, ctx=ctx)
    assert ch.placement["tuned"] and ch.placement["candidates_tried"] == 0 and ch.placement["spacer_GiB"] == 0
    assert ch.placement["mode"] == "own blocks" and ch.placement["setup_s"] >= 0
    ch.free()
    # (a) the chain's own allocator, search opted in
    ch = chain.Chain(n, n, ctx=ctx, tune_placement="search")
    info = ch.placement
    assert info["tuned"] and info["n_classes"] == 3 and info["spacer_GiB"] >= 16 and info["mode"] == "search"
    assert info["spacer_GiB"] <= info["spacer_budget_GiB"]
    assert len({info["classes"][r] for r in ("slope", "ti", "mti")}) >= 2
    ch.run(dem.ptr)
    ctx.sync()
    for k, _ in chain.OUTPUTS:
        assert np.array_equal(ch.buf[k].to_host(), want[k], equal_nan=True), k
    ch.free()
    # (b) a caller's alloc / release pair (torch): everything not kept comes back
    held = {}

    def alloc(shape, dt):
        t = torch.empty(shape, dtype={np.float32: torch.float32, np.uint8: torch.uint8, np.int8: torch.int8,
                                      np.int32: torch.int32}[dt], device="cuda")
        held[t.data_ptr()] = t
        return t.data_ptr()

    def release(q):
        del held[q]
    ch = chain.Chain(n, n, ctx=ctx, alloc=alloc, release=release, tune_placement="search")
    assert ch.placement["spacer_GiB"] >= 16 and len(held) == len(chain.OUTPUTS)
    ch.run(dem.ptr)
    ctx.sync()
    ch.free()
    held.clear()
    # (c) a rank tile
    layout = tiling.Layout([n], [n])
    tl = tiling.RankTile(layout, 0, device=0, tune_placement="search")
    assert tl.placement["tuned"] and tl.placement["spacer_GiB"] >= 16
    tl.free()
    dem.free()
    ctx.close()
    torch.cuda.empty_cache()


def test_a_full_device_during_the_search_is_not_a_fault_and_leaves_no_stale_error(monkeypatch):
    """the search uses "the device is full" as control flow.  A candidate / spacer allocation that fails with
    out-of-memory must end the search quietly (MemoryError from the C ABI's DT_ENOMEM, torch's OutOfMemoryError), must
    not leave a sticky HIP error behind for the next entry point -- and any OTHER error must propagate."""
    from descriptools_amd import _lib, chain, placement
    from descriptools_amd.device import Context
    monkeypatch.setattr(placement, "WriteClassifier", _RunOfOneClass)
    n = 2048
    ctx = Context()
    L = _lib.lib()
    dem = ctx.empty((n, n), np.float32)
    _lib.check(L.dt_dev_synth_dem(ctx.h, 3, n, n, 0, 0, n, n, 1, dem.ptr))
    # a real out-of-memory from the C ABI: MemoryError, and the next kernel launch does not see its ghost
    with pytest.raises(MemoryError):
        ctx.empty((1 << 50,), np.uint8)
    ch = chain.Chain(n, n, ctx=ctx, tune_placement=False)
    ch.run(dem.ptr)
    ctx.sync()
    want = {k: ch.buf[k].to_host() for k, _ in chain.OUTPUTS}
    ch.free()
    # spacers that cannot be had: the search carries on back to back and then works with what it found
    real_empty = Context.empty

    def stingy(self, shape, dtype):
        if int(np.prod(shape)) * np.dtype(dtype).itemsize >= (1 << 30):
            return real_empty(self, (1 << 50,), np.uint8)  # -> DT_ENOMEM -> MemoryError
        return real_empty(self, shape, dtype)
    monkeypatch.setattr(Context, "empty", stingy)
    ch = chain.Chain(n, n, ctx=ctx, tune_placement="search")
    assert ch.placement["tuned"] and ch.placement["spacer_GiB"] == 0 and ch.placement["candidates_tried"] >= 1
    assert ch.placement["n_classes"] == 3  # (back to back the candidates leave the stand-in's run after two)
    ch.run(dem.ptr)
    ctx.sync()
    for k, _ in chain.OUTPUTS:
        assert np.array_equal(ch.buf[k].to_host(), want[k], equal_nan=True), k
    ch.free()
    # a fault that is not out-of-memory is not swallowed

    def broken(self, shape, dtype):
        if int(np.prod(shape)) * np.dtype(dtype).itemsize >= (1 << 30):
            raise RuntimeError("descriptools_hip error -2: injected fault")
        return real_empty(self, shape, dtype)
    monkeypatch.setattr(Context, "empty", broken)
    with pytest.raises(RuntimeError, match="injected fault"):
        chain.Chain(n, n, ctx=ctx, tune_placement="search")
    monkeypatch.setattr(Context, "empty", real_empty)
    dem.free()
    ctx.close()
