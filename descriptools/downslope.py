"""descriptools.downslope -- the reference's import name for descriptools_amd.downslope (the MI355X implementation): a caller
written for the reference (`import descriptools.downslope as downslope`, Example/example.py:11-16) runs unchanged."""
from descriptools_amd.downslope import *  # noqa: F401,F403
from descriptools_amd import downslope as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("_")]
