"""one synthetic 16384^2 DEM, D8, then the downslope kernel a few times (for rocprofv3 --pmc runs)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from descriptools_amd import _lib
from descriptools_amd.device import Context
L = _lib.lib()
S = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
dz = float(sys.argv[2]) if len(sys.argv) > 2 else 5.0
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
ctx = Context(0, st.cuda_stream)
dem = torch.empty((S, S), dtype=torch.float32, device='cuda'); fdr = torch.empty((S, S), dtype=torch.uint8, device='cuda')
out = torch.empty((S, S), dtype=torch.float32, device='cuda')
_lib.check(L.dt_dev_synth_dem(ctx.h, 1, S, S, 0, 0, S, S, 0, dem.data_ptr()))
_lib.check(L.dt_dev_slope_d8(ctx.h, dem.data_ptr(), S, S, 10.0, None, fdr.data_ptr(), None))
for _ in range(2):
    _lib.check(L.dt_dev_downslope(ctx.h, dem.data_ptr(), fdr.data_ptr(), S, S, 10.0, dz, 0, out.data_ptr()))
torch.cuda.synchronize()
