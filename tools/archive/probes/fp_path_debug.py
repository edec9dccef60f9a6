import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from tomobar_amd.projector import HipTools3D
n, nz, na = 2560, 4, 1800
H = HipTools3D(n, 0, nz, np.linspace(0, np.pi, na, endpoint=False), 0.0, n, "gpu", 0, 12)
v = torch.rand((nz, n, n), device="cuda")
for sub in (0, 5):
    H.forward(v, sub); print(sub, H.kernel_path("fp"))
