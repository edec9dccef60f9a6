"""Is the two-iteration PD_TV kernel VALU- or HBM-bound?  Same launch on (a) random data (every wave takes the |p|>1
projection branch), (b) zeros (no wave takes it: ~45 % fewer VALU instructions, identical HBM traffic)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
os.environ.setdefault("TOMO_MI355X_FLAVOUR", "dev")  # A/B variants and measurement switches live in libtomo_mi355x_dev.so
from tomobar_amd.regularisersCuPy import PD_TV_cupy

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
for name, vol in (("random", torch.rand((N, N, N), device="cuda")), ("zeros", torch.zeros((N, N, N), device="cuda")),
                  ("smooth", torch.linspace(0, 1, N, device="cuda").view(1, 1, N).expand(N, N, N).contiguous())):
    out = torch.empty_like(vol)
    for lam in (0.01, 100.0):
        PD_TV_cupy(vol, lam, 4, 0, 1, 12.0, 0, False, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        PD_TV_cupy(vol, lam, 20, 0, 1, 12.0, 0, False, out=out)
        e1.record()
        torch.cuda.synchronize()
        print(f"{name:8s} lambda={lam:<6} {e0.elapsed_time(e1) / 20:7.3f} ms/iter")
