"""Co-scheduling probe (round 6, review item 3): do PD_TV (bound on the HBM request path), the back projector (LDS pipe) and the
forward projector (LDS + VALU) overlap when launched on two streams over disjoint arrays?  Reports t_both / (t_A + t_B):
1.0 = the two kernels serialise, 0.5 = perfect overlap of equal-length work.  The same with a device-to-device copy of halo
size on stream B (what an RCCL send/recv of the TV ghost planes costs the interior launch, docs/multi_gpu.md).
usage: python tools/cosched_probe.py [N] [NZ] [NA_S]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from tomobar_amd.projector import HipTools3D
from tomobar_amd.regularisersCuPy import PD_TV_cupy

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
NZ = int(sys.argv[2]) if len(sys.argv) > 2 else N
NA = int(sys.argv[3]) if len(sys.argv) > 3 else 75
dev = torch.device("cuda:0")
angles = np.linspace(0, np.pi, NA, endpoint=False)
H = HipTools3D(N, 0, NZ, angles, 0.0, N, "gpu", 0, None)
vol_a = torch.rand((NZ, N, N), device=dev)
out_a = torch.empty_like(vol_a)
vol_b = torch.rand((NZ, N, N), device=dev)
out_vb = torch.empty_like(vol_b)
sino_b = torch.rand((NZ, NA, N), device=dev)
out_sb = torch.empty_like(sino_b)
halo_src = torch.rand((12, N, N), device=dev)
halo_dst = torch.empty_like(halo_src)
big_src = torch.rand((256, N, N), device=dev)
big_dst = torch.empty_like(big_src)
sA, sB = torch.cuda.Stream(dev), torch.cuda.Stream(dev)


def pd():
    PD_TV_cupy(vol_a, 0.01, 30, 0, 1, 12.0, 0, False, out=out_a)   # 10 fused launches of 3 iterations


def timed(fa, fb, reps=3):
    """wall time (ms) from a common start to the end of both streams; fa on stream A, fb on stream B (either may be None)"""
    best = 1e30
    for _ in range(reps + 1):
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True)
        ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(torch.cuda.current_stream(dev))
        sA.wait_event(e0)
        sB.wait_event(e0)
        # stream B first: its (short) launches are queued before the host blocks on nothing -- both queues fill immediately
        if fb is not None:
            with torch.cuda.stream(sB):
                fb()
        if fa is not None:
            with torch.cuda.stream(sA):
                fa()
        ea.record(sA)
        eb.record(sB)
        torch.cuda.synchronize()
        best = min(best, max(e0.elapsed_time(ea), e0.elapsed_time(eb)))
    return best


def rep(fn, k):
    def f():
        for _ in range(k):
            fn()
    return f


t_pd = timed(pd, None)
print(f"N={N} NZ={NZ} NA_s={NA}: PD_TV(30 iterations = 10 launches) alone on stream A: {t_pd:8.2f} ms")
cases = [
    ("BP", lambda: H.backward(sino_b, None, out=out_vb)),
    ("FP", lambda: H.forward(vol_b, None, out=out_sb)),
    ("copy 12 planes (halo)", lambda: halo_dst.copy_(halo_src)),
    ("copy 256 planes (1 GiB)", lambda: big_dst.copy_(big_src)),
]
for name, fn in cases:
    t1 = timed(None, fn)
    k = max(1, int(round(t_pd / t1)))
    tb = timed(None, rep(fn, k))
    both = timed(pd, rep(fn, k))
    print(f"{name:24s}: one call {t1:8.3f} ms; x{k:<4d} alone on stream B {tb:8.2f} ms; with PD_TV on stream A {both:8.2f} ms"
          f"  -> t_both / (t_A + t_B) = {both / (t_pd + tb):.3f}")
# the projector pair against itself (FP of sub-iteration s+1 beside BP of s would need independent data; here: disjoint arrays)
t_bp = timed(rep(cases[0][1], 8), None)
t_fp = timed(None, rep(cases[1][1], 4))
both = timed(rep(cases[0][1], 8), rep(cases[1][1], 4))
print(f"BP x8 on A {t_bp:8.2f} ms, FP x4 on B {t_fp:8.2f} ms, together {both:8.2f} ms -> {both / (t_bp + t_fp):.3f}")
