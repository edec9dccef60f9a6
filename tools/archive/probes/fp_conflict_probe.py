"""FP timing for angle sets with different detector-axis strides (LDS bank-conflict exposure of the tap reads).
usage: python tools/archive/probes/fp_conflict_probe.py [N] [NZ]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import numpy as np
import torch
from tomobar_amd.projector import HipTools3D

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
NZ = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
NA = 72
vol = torch.rand((NZ, N, N), device="cuda")


def run(name, angles):
    H = HipTools3D(N, 0, NZ, angles, 0.0, N, "gpu", 0, None)
    out = torch.empty((NZ, len(angles), N), device="cuda")
    for _ in range(2):
        H.forward(vol, None, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        H.forward(vol, None, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    print(f"{name:42s} {ms:8.3f} ms  {NZ*N*N*len(angles)/ms/1e6:8.1f} GUPS")


deg = np.pi / 180
run("72 angles in +-9 deg (stride 1.00-1.01)", np.linspace(-9 * deg, 9 * deg, NA))
run("72 angles in 36..44 deg (stride 1.24-1.39)", np.linspace(36 * deg, 44 * deg, NA))
run("72 angles in 20..38 deg (stride 1.06-1.27)", np.linspace(20 * deg, 38 * deg, NA))
run("72 angles over 180 deg (subset-like)", np.linspace(0, np.pi, NA, endpoint=False))
