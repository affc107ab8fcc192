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
