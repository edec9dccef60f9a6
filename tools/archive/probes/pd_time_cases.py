"""PD_TV prox time (30 iterations, 1024^3) over the argument combinations that select different kernel instantiations:
nonneg x methodTV x {float32, binary16 duals} x {default, exact}.  usage: python tools/archive/probes/pd_time_cases.py [N] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
from tomobar_amd import ops
from tomobar_amd.regularisersCuPy import PD_TV_cupy
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 3
vol = torch.rand((N, N, N), device="cuda")
out = torch.empty_like(vol)
for variant in (0, 22):
    ops.set_variant("pdtv", variant)
    for half in (False, True):
        for methodTV in (0, 1):
            for nonneg in (0, 1):
                PD_TV_cupy(vol, 0.01, 30, methodTV, nonneg, 12.0, 0, half, out=out); torch.cuda.synchronize()
                ts = []
                for _ in range(REPS):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(); PD_TV_cupy(vol, 0.01, 30, methodTV, nonneg, 12.0, 0, half, out=out); e1.record(); torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1) / 10)
                print(f"variant {variant:2d} half {int(half)} methodTV {methodTV} nonneg {nonneg}: {min(ts):7.3f} ms per three-iteration launch", flush=True)
ops.set_variant("pdtv", 0)
print("placement of the scratch arena:", ops.placement_last(), flush=True)
