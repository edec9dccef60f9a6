"""PD_TV through the slab driver on one GPU (world 1: no ghosts, no exchange), work arrays placed by the library or
allocated by torch.  usage: python tools/archive/probes/slab_pd_time.py [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
from tomobar_amd import ops, slab
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda", 0)
vol = torch.rand((N, N, N), device=dev)
out = torch.empty_like(vol)
comm = slab.SlabComm(0, 1, dev)
def run(placed):
    keep = slab._hip_alloc
    if not placed:
        slab._hip_alloc = lambda specs, device: [torch.empty(sh, dtype=dt, device=device) for sh, dt in specs]
    try:
        slab.pd_tv_slab(vol, comm, 0.01, 30, 0, 1, 12.0, False, out=out); torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); slab.pd_tv_slab(vol, comm, 0.01, 30, 0, 1, 12.0, False, out=out); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 10)
        return min(ts)
    finally:
        slab._hip_alloc = keep
for placed in (False, True, False, True):
    print(f"slab driver, work arrays {'placed by the library' if placed else 'from torch           '}: {run(placed):7.3f} ms per three-iteration launch (incl. the copy of the slab into its ghosted array); placement {ops.placement_last() if placed else None}", flush=True)
