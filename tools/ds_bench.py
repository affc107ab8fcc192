"""downslope kernel time vs elevation difference (= walk length): fixed (staging) cost vs per-move cost"""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from descriptools_amd import _lib
from descriptools_amd.device import Context
L=_lib.lib()
S=16384
st=torch.cuda.Stream(); torch.cuda.set_stream(st)
ctx=Context(0, st.cuda_stream)
dem=torch.empty((S,S),dtype=torch.float32,device='cuda'); fdr=torch.empty((S,S),dtype=torch.uint8,device='cuda'); out=torch.empty((S,S),dtype=torch.float32,device='cuda')
_lib.check(L.dt_dev_synth_dem(ctx.h,1,S,S,0,0,S,S,0,dem.data_ptr()))
_lib.check(L.dt_dev_slope_d8(ctx.h,dem.data_ptr(),S,S,10.0,None,fdr.data_ptr(),None))
for dz in (0.001, 0.5, 1.0, 2.5, 5.0, 7.5):
    for it in range(3):
        e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        e0.record(st); _lib.check(L.dt_dev_downslope(ctx.h,dem.data_ptr(),fdr.data_ptr(),S,S,10.0,dz,0,out.data_ptr())); e1.record(st); torch.cuda.synchronize()
    print('dz',dz,'ms %.3f'%e0.elapsed_time(e1), flush=True)
