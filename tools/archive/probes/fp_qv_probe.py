"""EXPERIMENT (round 5): forward projector staged by LDS-DMA from quad-interleaved volume copies (fp variants 4 = 8 angles, 5 = 16 angles
per 1024-thread workgroup; dev flavour) against the shipped forms.  The copies are made once per volume pointer (warm-up call), so the
timings are the projection kernels alone.  usage: python tools/archive/probes/fp_qv_probe.py N NZ NA OS"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
os.environ["TOMO_MI355X_FLAVOUR"] = "dev"
import statistics
import numpy as np, torch
from tomobar_amd import ops
from tomobar_amd.projector import HipTools3D
N, NZ, NA, OS = (int(v) for v in sys.argv[1:5])
H = HipTools3D(N, 0, NZ, np.linspace(0, np.pi, NA, endpoint=False), 0.0, N, "gpu", 0, OS if OS > 1 else None)
vol = torch.rand((NZ, N, N), device="cuda")
sub = 3 if OS > 1 else None
outs, res, paths = {}, {}, {}
for rnd in range(4):
    for v in (0, 4, 5):
        ops.set_variant("fp", v)
        try:
            out = H.forward(vol, sub)
        except ValueError as e:
            paths[v] = f"not available: {e}"
            continue
        torch.cuda.synchronize()
        outs[v], paths[v] = out, H.kernel_path("fp")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            H.forward(vol, sub, out=out)
        e1.record(); torch.cuda.synchronize()
        res.setdefault(v, []).append(e0.elapsed_time(e1) / 3)
ops.set_variant("fp", 0)
print(f"N={N} NZ={NZ} angles {NA}/{OS}")
for v in (0, 4, 5):
    if v in res:
        same = bool(torch.equal(outs[v], outs[0]))
        print(f"fp variant {v}: median {statistics.median(res[v]):8.3f} min {min(res[v]):8.3f} ms  bit-identical to variant 0: {same}   {paths[v]}", flush=True)
    else:
        print(f"fp variant {v}: {paths.get(v)}", flush=True)
