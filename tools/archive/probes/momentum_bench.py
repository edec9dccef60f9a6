"""Time of tomo_momentum_transposed (FISTA momentum + in-plane transposed copy) and of the forward projector's own transpose
pass at n^2 x nz, HIP events.  usage: python tools/archive/probes/momentum_bench.py [n nz]"""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from tomobar_amd.projector import HipTools3D

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
nz = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
H = HipTools3D(n, 0, nz, np.linspace(0, np.pi, 24, endpoint=False), 0.0, n, "gpu", 0, None)
x, xo, xt = (torch.rand((nz, n, n), device="cuda") for _ in range(3))


def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


V = nz * n * n * 4
ms = timeit(lambda: H.momentum(x, xo, xt, 0.37))
print(f"momentum + transposed copy {n}^2 x {nz}: {ms:7.3f} ms  {4 * V / ms / 1e6:7.1f} GB/s (2 reads + 2 writes)")
want = x + np.float32(0.37) * (x - xo)
assert torch.equal(xt, want), "momentum values"
