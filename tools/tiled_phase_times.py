"""Times of the multi-rank phases on ONE GPU: 2 x 2 logical ranks of S x S tiles (inflow across ranks is real,
unlike bench.py --tiled at world 1), per-phase HIP-event times of rank 3 (the most downstream one)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from descriptools_amd import tiling
S = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
layout = tiling.Layout([S, S], [S, S])
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
tiles = []
for r in range(4):
    t = tiling.RankTile(layout, r, device=0, stream=st.cuda_stream, px=10.0, river_threshold=(layout.Hg * layout.Wg) // 512)
    t.synth_dem(1); tiles.append(t)
def ev(): return torch.cuda.Event(enable_timing=True)
for rep in range(3):
    for t in tiles:
        t.d8(); t.fa_local(sync=False); t.fill_ring_codes()
    rows = torch.cat([t.fa_row for t in tiles])
    times = {}
    for t in tiles:
        a, b = ev(), ev(); a.record(st); t.fa_solve_finish(rows); b.record(st); times[("fa_finish", t.rank)] = (a, b)
    for t in tiles:
        a, b = ev(), ev(); a.record(st); t.fh_local(sync=False); b.record(st); times[("fh_local", t.rank)] = (a, b)
    rows = torch.cat([t.fh_row for t in tiles])
    for t in tiles:
        a, b = ev(), ev(); a.record(st); t.fh_solve_finish(rows, fuse_gfi=True, want_a_river=False); b.record(st); times[("fh_finish", t.rank)] = (a, b)
    torch.cuda.synchronize()
print({k: round(a.elapsed_time(b), 3) for k, (a, b) in times.items()})
