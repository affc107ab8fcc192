"""Fourth probe: a pool of P rasters, the fused stencil timed on every triple of them as its three outputs."""
import itertools, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from descriptools_amd import _lib
from descriptools_amd.device import Context
L = _lib.lib()
S = 16384
P = int(sys.argv[1]) if len(sys.argv) > 1 else 7
ballast = int(sys.argv[2]) if len(sys.argv) > 2 else 0   # GiB allocated first (a bench-like heap)
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
ctx = Context(0, st.cuda_stream)
n = S * S
def timed(fn, reps=4):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps): fn()
    e1.record(st); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
hold = [torch.empty(n, dtype=torch.float32, device="cuda") for _ in range(ballast)]
dem = torch.empty(n, dtype=torch.float32, device="cuda")
fac = torch.zeros(n, dtype=torch.int32, device="cuda")
_lib.check(L.dt_dev_synth_dem(ctx.h, 1, S, S, 0, 0, S, S, 0, dem.data_ptr()))
pool = [torch.empty(n, dtype=torch.float32, device="cuda") for _ in range(P)]
res = []
for a, b, c in itertools.combinations(range(P), 3):
    ms = timed(lambda: L.dt_dev_slope_twi(ctx.h, dem.data_ptr(), fac.data_ptr(), S, S, 10.0, 0.1, pool[a].data_ptr(), None, pool[b].data_ptr(), pool[c].data_ptr()))
    res.append((ms, (a, b, c)))
res.sort()
print("pool %d, ballast %d GiB: best %s ... worst %s" % (P, ballast, ["%.3f %s" % r for r in res[:5]], ["%.3f %s" % r for r in res[-3:]]))
import collections
cnt = collections.Counter()
for ms, t in res[:max(len(res) // 5, 1)]:
    cnt.update(t)
print("members of the fastest fifth:", sorted(cnt.items()))
print("single-raster effect: mean time of the triples containing raster i:", ["%d: %.3f" % (i, sum(m for m, t in res if i in t) / sum(1 for m, t in res if i in t)) for i in range(P)])
