"""A/B of the forward projector's per-angle lane multipliers (round 6): FP variant 0 (shipped) against variant 4 (the same
whole-row form with pixel = lane, the round-5 mapping), dev flavour, same process, interleaved; outputs must be identical.
usage: TOMO_MI355X_FLAVOUR=dev python tools/fp_mult_ab.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from tomobar_amd import _lib, ops
from tomobar_amd.projector import HipTools3D

assert _lib.flavour() == "dev", "run with TOMO_MI355X_FLAVOUR=dev"
CASES = [("configs[2] 1024^3, subset of 75 of 900", 1024, 1024, 900, 12), ("configs[4] share 2560^2 x 270, subset of 150 of 1800", 2560, 270, 1800, 12),
         ("configs[1] 256^3, 360 angles", 256, 256, 360, None), ("512^3, 360 angles", 512, 512, 360, None),
         ("640-wide, 128 slices, subset of 60 of 720", 640, 128, 720, 12)]
for name, n, nz, na, os_n in CASES:
    angles = np.linspace(0, np.pi, na, endpoint=False)
    H = HipTools3D(n, 0, nz, angles, 0.0, n, "gpu", 0, os_n)
    vol = torch.rand((nz, n, n), device="cuda")
    sub = 3 if os_n else None
    outs, times = {}, {0: [], 4: []}
    for rnd in range(4):
        for v in (0, 4):
            ops.set_variant("fp", v)
            out = H.forward(vol, sub)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                H.forward(vol, sub, out=out)
            e1.record()
            torch.cuda.synchronize()
            if rnd:
                times[v].append(e0.elapsed_time(e1) / 5)
            outs[v] = out
            path = H.kernel_path("fp")
    ops.set_variant("fp", 0)
    same = torch.equal(outs[0], outs[4])
    t0, t4 = min(times[0]), min(times[4])
    print(f"{name:52s}: multipliers {t0:8.3f} ms, pixel = lane {t4:8.3f} ms -> x {t4 / t0:.3f}; identical: {same}; {path}")
    del H, vol, outs
