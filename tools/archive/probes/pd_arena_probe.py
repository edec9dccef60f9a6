"""Does the PD_TV launch time depend on WHERE the library's scratch arena was allocated?  The arena is per (device, stream):
the same prox on the same volume is run on several torch streams, each of which makes the library allocate its own 34 GB
arena, in order.  usage: python tools/archive/probes/pd_arena_probe.py [N] [streams]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
from tomobar_amd.regularisersCuPy import PD_TV_cupy
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
S = int(sys.argv[2]) if len(sys.argv) > 2 else 5
vol = torch.rand((N, N, N), device="cuda")
out = torch.empty_like(vol)
streams = [torch.cuda.Stream() for _ in range(S)]
def run(st):
    with torch.cuda.stream(st):
        PD_TV_cupy(vol, 0.01, 30, 0, 1, 12.0, 0, False, out=out); st.synchronize()
        ts = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st); PD_TV_cupy(vol, 0.01, 30, 0, 1, 12.0, 0, False, out=out); e1.record(st); st.synchronize()
            ts.append(e0.elapsed_time(e1) / 10)
    return min(ts)
for rep in range(2):
    for i, st in enumerate(streams):
        free, total = torch.cuda.mem_get_info()
        print(f"pass {rep} stream {i} (arena {i} of the process, {(total - free) / 1e9:6.1f} GB in use): {run(st):7.3f} ms per three-iteration launch", flush=True)
