"""Interleaved A/B of PD_TV kernel variants (median and min over rounds): python tools/archive/probes/pd_sweep.py [N] [variants...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import statistics
import torch
os.environ.setdefault("TOMO_MI355X_FLAVOUR", "dev")  # A/B variants and measurement switches live in libtomo_mi355x_dev.so
from tomobar_amd import ops
from tomobar_amd.regularisersCuPy import PD_TV_cupy
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
variants = [int(v) for v in sys.argv[2:]] or [0, 2, 3, 21]
ROUNDS, IT = 5, int(os.environ.get("PD_IT", "12"))   # 12 = 4 launches of 3 (or 6 of 2)
DATA = os.environ.get("PD_DATA", "rand")
if DATA == "phantom":   # the bench's kind of volume: piecewise-constant phantom + mild noise
    import bench
    vol = bench.phantom_slab(N, N, 0, N, torch.device("cuda")) + 0.02 * torch.randn((N, N, N), device="cuda")
else:
    vol = torch.rand((N, N, N), device="cuda")
out = torch.empty_like(vol)
res = {}
for rnd in range(ROUNDS):
    for v in variants:
        ops.set_variant("pdtv", v)
        for half in (False, True):
            if rnd == 0:
                PD_TV_cupy(vol, 0.01, 2, 0, 1, 12.0, 0, half, out=out)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            PD_TV_cupy(vol, 5e-4 if DATA == "phantom" else 0.01, IT, 0, 1, 12.0, 0, half, out=out)
            e1.record(); torch.cuda.synchronize()
            res.setdefault((v, half), []).append(e0.elapsed_time(e1) / IT)
for (v, half), ts in res.items():
    med = statistics.median(ts)
    print(f"PD_TV v{v} half={int(half)}: median {med:6.3f} min {min(ts):6.3f} ms/iter  {(24 if half else 36)*N**3/med/1e6:7.1f} GB/s alg", flush=True)
