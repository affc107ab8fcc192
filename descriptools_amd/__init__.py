"""descriptools_amd -- MI355X-native drop-in for the descriptools terrain-descriptor hot path.

Same module / function names as the reference package (callers import the submodules, as
Example/example.py:11-16 does): slope, flowhand, topoindexes, gfi, downslope, evaluation, helpers;
net-new: flowdir (D8), flowacc, chain (device-resident full chain), tiling (multi-GPU).
All compute goes through libdescriptools_hip.so (include/descriptools_hip.h); there is no CPU path.
"""
__version__ = "0.1.0"
