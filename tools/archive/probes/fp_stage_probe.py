"""What the staging (global loads + LDS writes of the volume rows) costs the forward projector: the probe switch skips it
(the sampling loop, barriers and epilogue are unchanged; results are garbage); 32 skips only the LDS writes.  usage: python tools/archive/probes/fp_stage_probe.py [N] [NZ] [NA]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import statistics
import numpy as np
import torch
os.environ.setdefault("TOMO_MI355X_FLAVOUR", "dev")  # A/B variants and measurement switches live in libtomo_mi355x_dev.so
from tomobar_amd import ops
from tomobar_amd.projector import HipTools3D
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
NZ = int(sys.argv[2]) if len(sys.argv) > 2 else N
NA = int(sys.argv[3]) if len(sys.argv) > 3 else 900
OS = int(sys.argv[4]) if len(sys.argv) > 4 else 12
H = HipTools3D(N, 0, NZ, np.linspace(0, np.pi, NA, endpoint=False), 0.0, N, "gpu", 0, OS if OS > 1 else None)
vol = torch.rand((NZ, N, N), device="cuda")
sub = 3 if OS > 1 else None
out = torch.empty(H.sino_shape(sub), device="cuda")
res = {}
for rnd in range(4):
    for probe in (0, 32, 16):
        ops.set_variant("probe", probe)
        H.forward(vol, sub, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            H.forward(vol, sub, out=out)
        e1.record(); torch.cuda.synchronize()
        res.setdefault(probe, []).append(e0.elapsed_time(e1) / 3)
ops.set_variant("probe", 0)
print(f"N={N} NZ={NZ} angles {NA}/{OS}: {H.kernel_path('fp')}")
for probe, ts in res.items():
    print(f"probe {probe:2d}: median {statistics.median(ts):7.3f} min {min(ts):7.3f} ms per forward projection (incl. the 2 ms transpose)", flush=True)
