"""Per-kernel timing on the GPU (HIP events on the launch stream): algorithmic GB/s and GUPS.
usage: python tools/kernel_bench.py [N] [NZ] [NA_S] [reps]"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tomobar_amd import _lib, ops
from tomobar_amd.projector import HipTools3D
from tomobar_amd.regularisersCuPy import PD_TV_cupy, ROF_TV_cupy

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
NZ = int(sys.argv[2]) if len(sys.argv) > 2 else N
NA = int(sys.argv[3]) if len(sys.argv) > 3 else 75
REPS = int(sys.argv[4]) if len(sys.argv) > 4 else 3


def timeit(fn, reps=REPS, warm=1):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps  # ms


V = NZ * N * N
S = NZ * NA * N
print(f"N={N} NZ={NZ} NA={NA}: V={V*4/1e9:.2f} GB, S={S*4/1e9:.3f} GB")
angles = np.linspace(0, np.pi, NA, endpoint=False)
H = HipTools3D(N, 0, NZ, angles, 0.0, N, "gpu", 0, None)
vol = torch.rand((NZ, N, N), device="cuda")
sino = torch.rand((NZ, NA, N), device="cuda")
out_v = torch.empty_like(vol)
out_s = torch.empty_like(sino)
DEV = _lib.flavour() == "dev"   # TOMO_MI355X_FLAVOUR=dev python tools/kernel_bench.py ...: the A/B variants as well
print(f"library flavour: {_lib.flavour()}")
for variant in ((0, 2, 1) if DEV else (0,)):
    ops.set_variant("bp", variant)
    ms = timeit(lambda: H.backward(sino, None, out=out_v))
    print(f"BP  variant {variant}: {ms:8.3f} ms  {4*(S+V)/ms/1e6:8.1f} GB/s alg  {V*NA/ms/1e6:8.1f} GUPS")
ops.set_variant("bp", 0)
for variant in ((0, 2, 1) if DEV else (0,)):
    ops.set_variant("fp", variant)
    ms = timeit(lambda: H.forward(vol, None, out=out_s))
    print(f"FP  variant {variant}: {ms:8.3f} ms  {4*(S+V)/ms/1e6:8.1f} GB/s alg  {V*NA/ms/1e6:8.1f} GUPS")
ops.set_variant("fp", 0)
IT = 10
for variant in ((0, 22, 3, 2, 21, 1) if DEV else (0, 22)):   # 0 = shipped default, 22 = the reference's roundings for float32 duals too
    ops.set_variant("pdtv", variant)
    for half in (False, True):
        ms = timeit(lambda: PD_TV_cupy(vol, 0.01, IT, 0, 1, 12.0, 0, half, out=out_v)) / IT
        bpv = 24 if half else 36
        print(f"PD_TV v{variant} half={int(half)}: {ms:8.3f} ms/iter  {bpv*V/ms/1e6:8.1f} GB/s alg")
ops.set_variant("pdtv", 0)
for variant in ((0, 2, 3, 1) if DEV else (0,)):
    ops.set_variant("roftv", variant)
    ms = timeit(lambda: ROF_TV_cupy(vol, 0.01, IT, 0.001, 0, False, out=out_v)) / IT
    print(f"ROF_TV v{variant}     : {ms:8.3f} ms/iter  {12*V/ms/1e6:8.1f} GB/s alg")
ops.set_variant("roftv", 0)
x2 = torch.rand_like(vol)
ms = timeit(lambda: ops.momentum(vol, x2, out_v, 0.5))
print(f"momentum      : {ms:8.3f} ms  {12*V/ms/1e6:8.1f} GB/s")
