"""Upper bound on what any halo-sharing scheme could buy the K = 3 PD_TV kernel: the probe switch aliases the halo rows
(bit 1) / halo lanes (bit 2) onto the workgroup's own tile, so HBM sees the compulsory traffic only while the arithmetic
is unchanged (results are garbage); bit 4: every plane access goes to plane 0 (cache-resident: the kernel without HBM);
bit 8 (round 4): ONE workgroup per CU (72 KiB of dynamic LDS reserved on top of the kernel's 80 KiB), i.e. one wave per
SIMD instead of two -- alone, and with bit 4 (12) = the instruction-issue floor of a single wave per SIMD.
usage: python tools/archive/probes/pd_halo_probe.py [N] [iters] [variant]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import statistics
import torch
os.environ.setdefault("TOMO_MI355X_FLAVOUR", "dev")  # A/B variants and measurement switches live in libtomo_mi355x_dev.so
from tomobar_amd import ops
from tomobar_amd.regularisersCuPy import PD_TV_cupy
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
IT = int(sys.argv[2]) if len(sys.argv) > 2 else 30
ops.set_variant("pdtv", int(sys.argv[3]) if len(sys.argv) > 3 else 0)   # 0 = shipped default (relaxed float32), 22 = the reference's roundings
vol = torch.rand((N, N, N), device="cuda")
out = torch.empty_like(vol)
res = {}
for rnd in range(4):
    for probe in (0, 1, 2, 3, 4, 8, 12):
        ops.set_variant("probe", probe)
        if rnd == 0:
            PD_TV_cupy(vol, 0.01, 3, 0, 1, 12.0, 0, False, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        PD_TV_cupy(vol, 0.01, IT, 0, 1, 12.0, 0, False, out=out)
        e1.record(); torch.cuda.synchronize()
        res.setdefault(probe, []).append(e0.elapsed_time(e1) / IT)
ops.set_variant("probe", 0)
for probe, ts in res.items():
    print(f"probe {probe}: median {statistics.median(ts):6.3f} min {min(ts):6.3f} ms/iter", flush=True)
