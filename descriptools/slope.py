"""descriptools.slope -- the reference's import name for descriptools_amd.slope (the MI355X implementation): a caller
written for the reference (`import descriptools.slope as slope`, Example/example.py:11-16) runs unchanged."""
from descriptools_amd.slope import *  # noqa: F401,F403
from descriptools_amd import slope as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("_")]
