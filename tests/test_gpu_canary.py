"""GPU (-m gpu): no kernel of the chain writes outside its rasters, whatever the raster's tile count.

Round 3 lost a work-in-progress defect (gpurun_out/call2.log: an abort at the first host synchronisation after the
untiled chain on a 384 x 640 raster -- the first raster of that suite whose 64 x 64 tile count, 60, is not a multiple of
16; every raster before it had 48 tiles; the committed code passed the same test sixteen minutes later and the tree that
crashed was not kept).  This test makes that class of defect deterministic instead of a matter of what
happens to lie behind a raster: every output raster of the chain sits between guard bands of a known pattern inside
one allocation, the chain runs over shapes whose tile counts cover every residue mod 16 (and ragged last rows /
columns), and the bands must come back untouched."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GUARD = 1 << 16  # bytes on either side of every raster


@pytest.mark.parametrize("long_walks", [False, True])
def test_guard_bands_around_every_raster_survive_every_tile_count(long_walks):
    import torch
    from descriptools_amd import _lib, chain
    from descriptools_amd.device import Context
    L = _lib.lib()
    ctx = Context()
    # (tile rows, tile columns, cells missing from the last tile row / column): tile counts 16, 17, 2, 3, ... cover
    # every residue mod 16; the 6 x 10 raster is the one of round 3's abort
    shapes = [(64 * ty - ry, 64 * tx - rx) for ty, tx, ry, rx in
              [(2, 8, 0, 0), (1, 17, 0, 5), (2, 1, 0, 0), (3, 1, 1, 0), (2, 2, 30, 17), (5, 1, 0, 0), (2, 3, 0, 63),
               (7, 1, 0, 0), (2, 4, 3, 3), (3, 3, 0, 0), (2, 5, 0, 0), (3, 9, 63, 1), (2, 6, 0, 0), (3, 15, 0, 0),
               (2, 7, 0, 9), (3, 5, 0, 0), (6, 10, 0, 0), (1, 1, 0, 0), (1, 1, 40, 50)]]
    assert len({((h + 63) // 64) * ((w + 63) // 64) % 16 for h, w in shapes}) == 16  # every residue of the tile count
    for H, W in shapes:
        sizes = {name: H * W * np.dtype(dt).itemsize for name, dt in chain.OUTPUTS}
        offs, total = {}, GUARD
        for name, nb in sizes.items():
            offs[name] = total
            total += (nb + 255) // 256 * 256 + GUARD
        arena = torch.full((total,), 0xA5, dtype=torch.uint8, device="cuda")
        base = arena.data_ptr()
        assert base % 256 == 0
        names = iter([n for n, _ in chain.OUTPUTS])
        dem = ctx.empty((H, W), np.float32)
        _lib.check(L.dt_dev_synth_dem(ctx.h, 7, H, W, 0, 0, H, W, 2, dem.ptr))
        ch = chain.Chain(H, W, ctx=ctx, px=10.0, alloc=lambda shape, dt: base + offs[next(names)], tune_placement=False,
                         long_walks=long_walks)
        for _ in range(2):
            ch.run(dem.ptr)
        ctx.sync()
        ch.free()
        host = arena.cpu().numpy()
        inside = np.zeros(total, bool)
        for name, nb in sizes.items():
            inside[offs[name]:offs[name] + nb] = True
        bad = np.flatnonzero(~inside & (host != 0xA5))
        assert bad.size == 0, "raster %dx%d: %d guard bytes overwritten, first at %d (rasters at %s)" % (
            H, W, bad.size, int(bad[0]), offs)
        dem.free()
        del arena
    ctx.close()
    torch.cuda.empty_cache()
