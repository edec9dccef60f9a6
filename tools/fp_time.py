"""Forward-projection time at the BASELINE shapes (HIP events, median / min of `reps` calls after a warm-up call); used under
tools/run_ab.sh for same-box A/B of library builds.  usage: python tools/fp_time.py [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from tomobar_amd.projector import HipTools3D

REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 5
CASES = [("configs[3] share 2048^2 x 256, 1500 angles", 2048, 256, 1500, None), ("configs[2] 1024^3, subset of 75 of 900", 1024, 1024, 900, 12),
         ("configs[4] share 2560^2 x 270, subset of 150 of 1800", 2560, 270, 1800, 12), ("512^3, 360 angles", 512, 512, 360, None)]
for name, n, nz, na, os_n in CASES:
    H = HipTools3D(n, 0, nz, np.linspace(0, np.pi, na, endpoint=False), 0.0, n, "gpu", 0, os_n)
    vol = torch.rand((nz, n, n), device="cuda")
    sub = 3 if os_n else None
    out = H.forward(vol, sub)
    torch.cuda.synchronize()
    ts = []
    for _ in range(REPS):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        H.forward(vol, sub, out=out)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    print(f"{name:52s}: median {ts[len(ts) // 2]:9.3f} min {ts[0]:9.3f} ms  checksum {float(out.double().sum()):.6e}  {H.kernel_path('fp')}")
    del H, vol, out
