"""PD_TV launch time at 1024^3 for a 30-iteration prox (ten three-iteration launches), shipped library: default / exact / binary16.
usage: python tools/archive/probes/pd_time.py [N] [reps] [NZ]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
from tomobar_amd import ops
from tomobar_amd.regularisersCuPy import PD_TV_cupy
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 3
NZ = int(sys.argv[3]) if len(sys.argv) > 3 else N
vol = torch.rand((NZ, N, N), device="cuda")
out = torch.empty_like(vol)
cases = [("default f32", 0, False), ("exact f32 (22)", 22, False), ("f16 duals", 0, True)]
from tomobar_amd import _lib
if _lib.flavour() == "dev":   # TOMO_MI355X_FLAVOUR=dev: workgroup shapes of the relaxed float32 kernel as well
    cases += [("relaxed 1x4 waves (31)", 31, False), ("relaxed 4x1 waves (32)", 32, False)]
for name, variant, half in cases:
    ops.set_variant("pdtv", variant)
    PD_TV_cupy(vol, 0.01, 30, 0, 1, 12.0, 0, half, out=out); torch.cuda.synchronize()
    ts = []
    for _ in range(REPS):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); PD_TV_cupy(vol, 0.01, 30, 0, 1, 12.0, 0, half, out=out); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10)
    print(f"{name:24s} {min(ts):7.3f} ms per three-iteration launch (min of {REPS}), median {sorted(ts)[len(ts)//2]:7.3f}; {min(ts) * 1e6 / (NZ * N * N):6.3f} ns per voxel  [{NZ} x {N}^2]", flush=True)
ops.set_variant("pdtv", 0)
print("placement of the scratch arena:", ops.placement_last(), flush=True)
