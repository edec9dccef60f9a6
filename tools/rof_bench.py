"""ROF_TV iteration time at 1024^3 (variants): python tools/rof_bench.py [N] [variants...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import statistics
import torch
os.environ.setdefault("TOMO_MI355X_FLAVOUR", "dev")  # A/B variants and measurement switches live in libtomo_mi355x_dev.so
from tomobar_amd import ops
from tomobar_amd.regularisersCuPy import ROF_TV_cupy
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
variants = [int(v) for v in sys.argv[2:]] or [0, 3]
vol = torch.rand((N, N, N), device="cuda")
out = torch.empty_like(vol)
IT = 10
res = {}
for rnd in range(4):
    for v in variants:
        for half in (False, True):
            ops.set_variant("roftv", v)
            if rnd == 0:
                ROF_TV_cupy(vol, 0.01, 2, 0.001, 0, half, out=out)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); ROF_TV_cupy(vol, 0.01, IT, 0.001, 0, half, out=out); e1.record(); torch.cuda.synchronize()
            res.setdefault((v, half), []).append(e0.elapsed_time(e1) / IT)
ops.set_variant("roftv", 0)
for (v, half), ts in res.items():
    print(f"ROF_TV v{v} half={int(half)}: median {statistics.median(ts):6.3f} min {min(ts):6.3f} ms/iter", flush=True)
