"""Times the single-iteration PD_TV kernel (one iteration per call), the two-iteration kernel and ROF_TV at N^3.
usage: python tools/archive/probes/tv_single_probe.py [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
os.environ.setdefault("TOMO_MI355X_FLAVOUR", "dev")  # A/B variants and measurement switches live in libtomo_mi355x_dev.so
from tomobar_amd import ops
from tomobar_amd.regularisersCuPy import PD_TV_cupy, ROF_TV_cupy
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
vol = torch.rand((N, N, N), device="cuda")
out = torch.empty_like(vol)


def t(fn, reps=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for half in (False, True):
    for it, label in ((1, "single"), (2, "pair"), (3, "triple")):
        ms = t(lambda: PD_TV_cupy(vol, 0.01, it, 0, 1, 12.0, 0, half, out=out))
        print(f"PD_TV {label:6s} half={int(half)}: {ms / it:6.3f} ms/iter", flush=True)
for v in [int(x) for x in os.environ.get("ROF_VARIANTS", "0").split(",")]:
    ops.set_variant("roftv", v)
    ms = t(lambda: ROF_TV_cupy(vol, 0.01, 12, 0.002, 0, False, out=out))
    print(f"ROF_TV v{v}: {ms / 12:6.3f} ms/iter")
