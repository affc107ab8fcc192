"""GPU (-m gpu): the chain captured as a HIP graph (Chain.capture, dt_ctx_capture_begin / _end, dt_graph_launch)
replays to exactly the rasters of the kernel-by-kernel run, on new DEM contents in the same buffer, with and without
the downslope side branch."""
import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("overlap", [False, True])
def test_captured_chain_equals_direct_run(overlap):
    from descriptools_amd import chain
    from descriptools_amd.device import Context
    H, W = 700, 900
    ctx = Context()
    ch = chain.Chain(H, W, ctx=ctx, px=10.0, river_threshold=H * W // 512, overlap=overlap)
    dems = [oracle.synth_dem(s, 2048, 2048, 100, 50, H, W, 2) for s in (3, 4)]
    d_dem = ctx.to_device(dems[0])
    g = ch.capture(d_dem.ptr)  # captured on the first DEM ...
    try:
        for dem in dems[::-1]:  # ... replayed on both: the graph holds pointers, not contents
            d_dem.copy_from(dem)
            ch.run(d_dem.ptr)
            ctx.sync()
            want = {k: ch.buf[k].to_host().copy() for k, _ in chain.OUTPUTS}
            for k, dt in chain.OUTPUTS:  # wipe the outputs: the replay must write them all again
                ch.buf[k].copy_from(np.full((H, W), 77, dt))
            g.launch()
            ctx.sync()
            for k, _ in chain.OUTPUTS:
                got = ch.buf[k].to_host()
                assert np.array_equal(got, want[k], equal_nan=True), k
    finally:
        g.free()
        d_dem.free()
        ch.free()
        ctx.close()


def test_stale_graph_is_refused():
    """a captured graph holds the raw addresses of the context's grow-only workspaces.  After a call that
    made the context reallocate one (here: a larger raster's flow accumulation), after freeing the chain, and after
    destroying the capturing context, a replay must fail loudly instead of writing into freed memory."""
    import ctypes as C
    from descriptools_amd import _lib, chain
    from descriptools_amd.device import Context
    L = _lib.lib()
    H, W = 256, 320
    ctx = Context()
    ch = chain.Chain(H, W, ctx=ctx, px=10.0, river_threshold=H * W // 512)
    d_dem = ctx.to_device(oracle.synth_dem(3, 2048, 2048, 0, 0, H, W, 0))
    g = ch.capture(d_dem.ptr)
    g.launch()
    ctx.sync()
    before = int(L.dt_ctx_scratch_bytes(ctx.h))
    # a larger raster on the same context: the scratch grows, the old block is freed
    big = chain.Chain(4 * H, 4 * W, ctx=ctx, px=10.0)
    d_big = ctx.to_device(oracle.synth_dem(4, 2048, 2048, 0, 0, 4 * H, 4 * W, 0))
    big.run(d_big.ptr)
    ctx.sync()
    assert int(L.dt_ctx_scratch_bytes(ctx.h)) > before
    with pytest.raises(RuntimeError, match="reallocated after the capture"):
        g.launch()
    g.free()
    # a fresh capture works again (the workspaces are large enough now) ...
    g2 = ch.capture(d_dem.ptr)
    g2.launch()
    ctx.sync()
    # ... until the chain whose buffers it addresses is freed
    ch.free()
    with pytest.raises(RuntimeError, match="has been freed"):
        g2.launch()
    # the C ABI's own guard: the capturing context is gone
    ctx2 = Context()
    ch2 = chain.Chain(H, W, ctx=ctx2, px=10.0)
    d2 = ctx2.to_device(oracle.synth_dem(5, 2048, 2048, 0, 0, H, W, 0))
    g3 = ch2.capture(d2.ptr)
    handle = g3.h
    d2.free()
    ch2.free()
    ctx2.close()
    assert L.dt_graph_launch(handle, ctx.h) != 0
    assert b"destroyed" in L.dt_last_error()
    L.dt_graph_destroy(handle)
    g2.free()
    big.free()
    d_big.free()
    d_dem.free()
    ctx.close()
