"""Timing of RecToolsDIRCuPy.FOURIER_INV on synthetic data.  usage: python tools/fourier_bench.py [n] [nproj] [nz] [reps]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tomobar_amd.methodsDIR_CuPy import RecToolsDIRCuPy

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
nproj = int(sys.argv[2]) if len(sys.argv) > 2 else 900
nz = int(sys.argv[3]) if len(sys.argv) > 3 else 256
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 2
angles = np.linspace(0, np.pi, nproj, endpoint=False)
rt = RecToolsDIRCuPy(n, 0, nz, 0.0, angles, n, device_projector=0)
data = torch.rand((nz, nproj, n), device="cuda")
rec = rt.FOURIER_INV(data)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    rec = rt.FOURIER_INV(data)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print(f"FOURIER_INV n={n} nproj={nproj} nz={nz}: {ms:.1f} ms  ({nz / ms * 1e3:.0f} slices/s), finite={bool(torch.isfinite(rec).all())}")
