"""BP with its fused epilogues at the bench shape: ms per call.  usage: python tools/archive/probes/bp_epi_bench.py [N] [NZ] [NA]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import statistics
import numpy as np
import torch
os.environ.setdefault("TOMO_MI355X_FLAVOUR", "dev")  # A/B variants and measurement switches live in libtomo_mi355x_dev.so
from tomobar_amd.projector import HipTools3D
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
NZ = int(sys.argv[2]) if len(sys.argv) > 2 else N
NA = int(sys.argv[3]) if len(sys.argv) > 3 else 75
H = HipTools3D(N, 0, NZ, np.linspace(0, np.pi, NA, endpoint=False), 0.0, N, "gpu", 0, None)
res = torch.rand((NZ, NA, N), device="cuda")
xt = torch.rand((NZ, N, N), device="cuda")
xo = torch.rand_like(xt)
zu = torch.empty_like(xt)
u = torch.rand_like(xt)
out = torch.empty_like(xt)
H.set_residual_layout("zquad")
resq = H.residual_buffer(None)
resq.copy_(torch.rand_like(resq))
H.set_residual_layout("planar")


def quad(fn):
    def run():
        H.set_residual_layout("zquad")
        try:
            fn()
        finally:
            H.set_residual_layout("planar")
    return run


cases = {
    "plain": lambda: H.backward(res, None, out=out),
    "fista": lambda: H.grad_step(res, xt, out, 1e-4, True, None),
    "fista (quad residual)": quad(lambda: H.grad_step(resq, xt, out, 1e-4, True, None)),
    "fista+momentum": lambda: H.grad_step_momentum(res, xt, xo, 1e-4, 0.5, True, None),
    "fista+mom. (quad)": quad(lambda: H.grad_step_momentum(resq, xt, xo, 1e-4, 0.5, True, None)),
    "admm": lambda: H.admm_z_update(res, xo, xt, u, zu, 1e-4, 1.0, True, 0.2, 0.8, True, None),
    "admm (quad residual)": quad(lambda: H.admm_z_update(resq, xo, xt, u, zu, 1e-4, 1.0, True, 0.2, 0.8, True, None)),
}
ts = {k: [] for k in cases}
for rnd in range(5):
    for k, fn in cases.items():
        if rnd == 0:
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); fn(); e1.record(); torch.cuda.synchronize()
        ts[k].append(e0.elapsed_time(e1) / 2)
for k, v in ts.items():
    print(f"BP {k:22s}: median {statistics.median(v):7.3f} min {min(v):7.3f} ms", flush=True)
