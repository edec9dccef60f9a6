"""Sweep PD_TV kernel variants: python tools/pd_sweep.py [N] [variants...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tomobar_amd import ops
from tomobar_amd.regularisersCuPy import PD_TV_cupy
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
variants = [int(v) for v in sys.argv[2:]] or [0, 3, 10, 11]
vol = torch.rand((N, N, N), device="cuda")
out = torch.empty_like(vol)
IT = 8
for v in variants:
    ops.set_variant("pdtv", v)
    for half in (False, True):
        PD_TV_cupy(vol, 0.01, 2, 0, 1, 12.0, 0, half, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        PD_TV_cupy(vol, 0.01, IT, 0, 1, 12.0, 0, half, out=out)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / IT
        print(f"PD_TV v{v} half={int(half)}: {ms:7.3f} ms/iter {(24 if half else 36)*N**3/ms/1e6:7.1f} GB/s", flush=True)
