"""Interpretive stand-in for ``numba.cuda`` (see package docstring). Test infrastructure only."""
import numpy as _np

_tid = [0]

# Optional subset of flat thread ids to execute (test-infrastructure knob used by
# oracle/gen_golden.py for the 20000-move / 5000-iteration cap cases, where running every
# thread interpretively would take hours).  None = run every thread.
THREAD_FILTER = None

# Numba compiles a read of a never-assigned local into an undefined value, it does not raise.  The reference
# relies on that in slope_gpu (slope.py:234-259: `aux` on border cells of a tile whose ring is real data; the
# garbage only reaches the ring slope_cpu strips, slope.py:202-205).  With this knob set a thread that hits
# UnboundLocalError is dropped where Numba would have written garbage -- used by oracle/gen_golden.py for the
# tiled slope cases only.
TOLERATE_UNBOUND = False


def grid(ndim):
    assert ndim == 1
    return _tid[0]


class _DeviceArray(_np.ndarray):
    def copy_to_host(self):
        return _np.array(self, copy=True).view(_np.ndarray)


def to_device(a):
    return _np.ascontiguousarray(_np.array(a, copy=True)).view(_DeviceArray)


class _Launcher:
    def __init__(self, f, blocks, threads):
        self.f, self.n = f, int(blocks) * int(threads)

    def __call__(self, *args):
        f = self.f
        ids = range(self.n) if THREAD_FILTER is None else [t for t in THREAD_FILTER if t < self.n]
        for t in ids:
            _tid[0] = t
            try:
                f(*args)
            except UnboundLocalError:
                if not TOLERATE_UNBOUND:
                    raise


class _Kernel:
    def __init__(self, f):
        self.f = f

    def __getitem__(self, cfg):
        blocks, threads = cfg
        return _Launcher(self.f, blocks, threads)


def jit(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return _Kernel(args[0])

    def deco(f):
        return _Kernel(f)

    return deco
