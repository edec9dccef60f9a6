"""Back-projection time (fused FISTA gradient step on the quad-interleaved residual, as the loop runs it, and plain tomo_bp3d) at the
BASELINE shapes (HIP events, median / min of `reps` calls); used under tools/run_ab.sh for same-box A/B of library builds.
usage: python tools/bp_time.py [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from tomobar_amd.projector import HipTools3D

REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 5
CASES = [("configs[2] 1024^3, subset of 75 of 900", 1024, 1024, 900, 12), ("configs[3] share 2048^2 x 256, 1500 angles", 2048, 256, 1500, None),
         ("configs[4] share 2560^2 x 270, subset of 150 of 1800", 2560, 270, 1800, 12), ("512^3, 360 angles", 512, 512, 360, None)]


def med(fn):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(REPS):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2], ts[0]


for name, n, nz, na, os_n in CASES:
    H = HipTools3D(n, 0, nz, np.linspace(0, np.pi, na, endpoint=False), 0.0, n, "gpu", 0, os_n)
    sub = 3 if os_n else None
    vol = torch.rand((nz, n, n), device="cuda")
    b = torch.rand(H.sino_shape(None), device="cuda")
    out = torch.empty_like(vol)
    H.set_residual_layout("zquad")
    res = H.residual_buffer(sub)
    H.residual(vol, b, None, "LS", sub, res)
    m1 = med(lambda: H.grad_step(res, vol, out, 1e-4, True, sub))
    chk = float(out.double().sum())
    H.set_residual_layout("planar")
    sino = torch.rand(H.sino_shape(sub), device="cuda")
    m2 = med(lambda: H.backward(sino, sub, out=out))
    print(f"{name:52s}: fused step median {m1[0]:9.3f} min {m1[1]:9.3f} ms | plain BP median {m2[0]:9.3f} min {m2[1]:9.3f} ms  checksum {chk:.6e}  {H.kernel_path('bp')}")
    del H, vol, b, out, res, sino
