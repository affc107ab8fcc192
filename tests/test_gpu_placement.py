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
