"""Which FP form runs for a shape, and what it costs (dev flavour: variant 3 lifts the workgroup-count condition of the dense-angle
form).  usage: TOMO_MI355X_FLAVOUR=dev python tools/archive/probes/fp_form_probe.py N NZ NA"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import numpy as np, torch
from tomobar_amd import _lib, ops
from tomobar_amd.projector import HipTools3D
N, NZ, NA = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
angles = np.linspace(0, np.pi, NA, endpoint=False)
H = HipTools3D(N, 0, NZ, angles, 0.0, N, "gpu", 0, None)
vol = torch.rand((NZ, N, N), device="cuda"); out = torch.empty((NZ, NA, N), device="cuda")
for v in ((0, 3) if _lib.flavour() == "dev" else (0,)):
    ops.set_variant("fp", v)
    H.forward(vol, None, out=out); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): H.forward(vol, None, out=out)
    e1.record(); torch.cuda.synchronize()
    print(f"N={N} NZ={NZ} NA={NA} fp variant {v}: {e0.elapsed_time(e1)/3:8.3f} ms   path: {H.kernel_path('fp') if hasattr(H, 'kernel_path') else '?'}", flush=True)
ops.set_variant("fp", 0)
