"""descriptools.evaluation -- the reference's import name for descriptools_amd.evaluation (the MI355X implementation): a caller
written for the reference (`import descriptools.evaluation as evaluation`, Example/example.py:11-16) runs unchanged."""
from descriptools_amd.evaluation import *  # noqa: F401,F403
from descriptools_amd import evaluation as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("_")]
