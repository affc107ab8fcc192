"""descriptools.helpers -- the reference's import name for descriptools_amd.helpers (the MI355X implementation): a caller
written for the reference (`import descriptools.helpers as helpers`, Example/example.py:11-16) runs unchanged."""
from descriptools_amd.helpers import *  # noqa: F401,F403
from descriptools_amd import helpers as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("_")]
