"""Timing of RecToolsDIRCuPy.FBP.  usage: python tools/fbp_bench.py [n] [angles] [nz]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tomobar_amd.methodsDIR_CuPy import RecToolsDIRCuPy

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
na = int(sys.argv[2]) if len(sys.argv) > 2 else 900
nz = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
rt = RecToolsDIRCuPy(n, 0, nz, 0.0, np.linspace(0, np.pi, na, endpoint=False), n, device_projector=0)
data = torch.rand((na, nz, n), device="cuda")
rt.FBP(data)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3):
    rec = rt.FBP(data)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 3
print(f"FBP n={n} angles={na} nz={nz}: {ms:.1f} ms ({nz / ms * 1e3:.0f} slices/s)")
