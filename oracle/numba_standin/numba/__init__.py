"""Stand-in ``numba`` namespace -- TEST INFRASTRUCTURE ONLY (used by oracle/gen_golden.py).

Numba is not installed in this image and there is no network.  The reference's
modules only need four names from numba (``cuda``, ``jit``, ``float32``,
``int32``); this package supplies interpretive equivalents so that the
reference's *unmodified* source under /root/reference can be imported and run,
in this container only, to generate the golden vectors committed under
tests/golden/.  Nothing in the product (descriptools_amd/) imports this.

Semantics:
  jit(f)            -> f                       (plain Python execution)
  float32 / int32   -> numpy dtypes
  cuda.jit(f)       -> launcher; ``k[blocks, threads](*args)`` runs ``f`` once per
                       flat thread id t in [0, blocks*threads), ``cuda.grid(1)`` -> t
  cuda.to_device(a) -> contiguous ndarray copy with ``.copy_to_host()``

Fidelity caveat (documented in DESIGN.md): numpy-2 scalar arithmetic keeps
``f32 (op) python-float`` in f32 where Numba promotes to f64, so gen_golden.py
feeds float64 copies of float32 inputs to reproduce Numba's promote-then-compute
arithmetic.
"""
import numpy as _np

from . import cuda  # noqa: F401

float32 = _np.float32
int32 = _np.int32


def jit(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]

    def deco(f):
        return f

    return deco
