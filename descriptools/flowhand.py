"""descriptools.flowhand -- the reference's import name for descriptools_amd.flowhand (the MI355X implementation): a caller
written for the reference (`import descriptools.flowhand as flowhand`, Example/example.py:11-16) runs unchanged."""
from descriptools_amd.flowhand import *  # noqa: F401,F403
from descriptools_amd import flowhand as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("_")]
