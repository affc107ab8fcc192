"""descriptools.topoindexes -- the reference's import name for descriptools_amd.topoindexes (the MI355X implementation): a caller
written for the reference (`import descriptools.topoindexes as topoindexes`, Example/example.py:11-16) runs unchanged."""
from descriptools_amd.topoindexes import *  # noqa: F401,F403
from descriptools_amd import topoindexes as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("_")]
