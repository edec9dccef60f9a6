"""With the library's scratch arena fixed (placed once), does the PD_TV prox time depend on where the CALLER's input / output
volumes lie?  Twelve candidate input volumes (torch allocations of 4.3 GB, held at once), the same prox on each; then the
same for the output.  usage: python tools/archive/probes/pd_input_probe.py [N] [count]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
from tomobar_amd import ops
from tomobar_amd.regularisersCuPy import PD_TV_cupy
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
K = int(sys.argv[2]) if len(sys.argv) > 2 else 12
src = torch.rand((N, N, N), device="cuda")
def t(vol, out):
    PD_TV_cupy(vol, 0.01, 30, 0, 1, 12.0, 0, False, out=out); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); PD_TV_cupy(vol, 0.01, 30, 0, 1, 12.0, 0, False, out=out); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10)
    return min(ts)
out0 = torch.empty_like(src)
print(f"reference: {t(src, out0):7.3f} ms; placement {ops.placement_last()}", flush=True)
cands = []
for k in range(K):
    c = torch.empty_like(src); c.copy_(src); cands.append(c)
for rep in range(2):
    print("input  candidates:", " ".join(f"{t(c, out0):6.3f}" for c in cands), flush=True)
for rep in range(2):
    print("output candidates:", " ".join(f"{t(src, c):6.3f}" for c in cands), flush=True)
