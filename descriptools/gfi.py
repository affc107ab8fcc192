"""descriptools.gfi -- the reference's import name for descriptools_amd.gfi (the MI355X implementation): a caller
written for the reference (`import descriptools.gfi as gfi`, Example/example.py:11-16) runs unchanged."""
from descriptools_amd.gfi import *  # noqa: F401,F403
from descriptools_amd import gfi as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("_")]
