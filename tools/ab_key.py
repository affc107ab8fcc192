"""A/B of a library debug key (dt_debug_set) on the serial chain at size^2: per-op times by events, and the rasters of
both settings compared bit for bit.   python tools/ab_key.py KEY VALUE [size] [steps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from descriptools_amd import _lib, chain  # noqa: E402
from descriptools_amd._lib import check  # noqa: E402
from descriptools_amd.device import Context  # noqa: E402

key, val = int(sys.argv[1]), int(sys.argv[2])
S = int(sys.argv[3]) if len(sys.argv) > 3 else 16384
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
L = _lib.lib()
st = torch.cuda.Stream()
torch.cuda.set_stream(st)
ctx = Context(0, st.cuda_stream)
dem = torch.empty((S, S), dtype=torch.float32, device="cuda")
check(L.dt_dev_synth_dem(ctx.h, 1, S, S, 0, 0, S, S, 0, dem.data_ptr()))
TD = {np.float32: torch.float32, np.uint8: torch.uint8, np.int8: torch.int8, np.int32: torch.int32}
keep = {}


def alloc(shape, dt):
    t = torch.empty(shape, dtype=TD[dt], device="cuda")
    keep[t.data_ptr()] = t
    return t.data_ptr()


ch = chain.Chain(S, S, ctx=ctx, alloc=alloc, overlap=False)


def timed():
    ops = ch.ops(dem.data_ptr(), serial=True)
    for _, _, call in ops:
        check(call())
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(len(ops) + 1)] for _ in range(steps)]
    for k in range(steps):
        ev[k][0].record(st)
        for i, (_, _, call) in enumerate(ops):
            check(call())
            ev[k][i + 1].record(st)
    torch.cuda.synchronize()
    out = {}
    for i, (name, _, _) in enumerate(ops):
        out[name] = float(np.median([ev[k][i].elapsed_time(ev[k][i + 1]) for k in range(steps)]))
    out["step"] = float(np.median([ev[k][0].elapsed_time(ev[k][-1]) for k in range(steps)]))
    return out


def snapshot():
    return {name: keep[int(ch.p(name))].view(TD[dt]).clone() for name, dt in chain.OUTPUTS if S <= 8192 or name in
            ("fac", "hand", "idx", "down")}


res = {}
for v in (0, val, 0, val):
    L.dt_debug_set(key, v)
    t = timed()
    res.setdefault(v, []).append(t)
    print("key %d = %d: " % (key, v) + "  ".join("%s %.3f" % kv for kv in t.items()), flush=True)
L.dt_debug_set(key, 0)
timed()
a = snapshot()
L.dt_debug_set(key, val)
timed()
b = snapshot()
L.dt_debug_set(key, 0)
for name in a:
    x, y = a[name], b[name]
    same = torch.equal(x, y) if x.dtype != torch.float32 else torch.equal(x.view(torch.int32), y.view(torch.int32))
    print("  %-8s %s" % (name, "identical" if same else "DIFFERENT (%d cells)" % int((x != y).sum())))
