"""Small fixed workloads for rocprofv3 --pmc passes (one kernel class per invocation, few launches).
usage: python tools/pmc_probe.py <what> [N] [NZ] [NA]   what in {pdtv0,pdtv1,pdtv0h,roftv,bp0,bp1,bp3,bpp,bpq,fp,fp4,fpq,momentum,fourier}"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
if os.environ.get("PMC_PROBE", "0") != "0" or (len(sys.argv) > 1 and sys.argv[1].rstrip("h") in ("pdtv1", "pdtv2", "pdtv21", "bp1", "bp2", "bp3", "bpp", "fp4")):
    os.environ.setdefault("TOMO_MI355X_FLAVOUR", "dev")   # measurement switches / A-B variants: libtomo_mi355x_dev.so
from tomobar_amd import ops
from tomobar_amd.projector import HipTools3D
from tomobar_amd.regularisersCuPy import PD_TV_cupy, ROF_TV_cupy

what = sys.argv[1]
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
NZ = int(sys.argv[3]) if len(sys.argv) > 3 else N
NA = int(sys.argv[4]) if len(sys.argv) > 4 else 75
vol = torch.rand((NZ, N, N), device="cuda")
out = torch.empty_like(vol)
if os.environ.get("PMC_PROBE", "0") != "0":
    ops.set_variant("probe", int(os.environ["PMC_PROBE"]))   # measurement switches (tools/archive/probes/pd_halo_probe.py)
if what.startswith("pdtv"):
    ops.set_variant("pdtv", int(what[4:].rstrip("h")))
    PD_TV_cupy(vol, 0.01, int(os.environ.get("PMC_PD_ITERS", "6")), 0, 1, 12.0, 0, what.endswith("h"), out=out)  # 3 + 3 (shipped f32) or 2 + 2 + 2; 9 = first (zero duals) + middle + last (no dual stores) launch
elif what == "roftv":
    ROF_TV_cupy(vol, 0.01, 4, 0.001, 0, False, out=out)
elif what.startswith("stream"):   # stream<nin><nout>[v] e.g. stream54 (dword) / stream54v (float4)
    import ctypes as C
    from tomobar_amd import _lib
    lib = _lib.lib()
    lib.tomo_diag_stream.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_void_p), C.c_int, C.c_size_t, C.c_int, C.c_int, C.c_void_p]
    nin, nout = int(what[6]), int(what[7])
    ins = [torch.rand_like(vol) for _ in range(nin)]
    outs = [torch.empty_like(vol) for _ in range(nout)]
    pi = (C.c_void_p * 8)(*[t.data_ptr() for t in ins])
    po = (C.c_void_p * 8)(*[t.data_ptr() for t in outs])
    for _ in range(3):
        lib.tomo_diag_stream(pi, nin, po, nout, vol.numel(), 4 if what.endswith("v") else 1, 2048, C.c_void_p(torch.cuda.current_stream().cuda_stream))
elif what == "fourier":
    from tomobar_amd.methodsDIR_CuPy import RecToolsDIRCuPy
    rt = RecToolsDIRCuPy(N, 0, NZ, 0.0, np.linspace(0, np.pi, NA, endpoint=False), N, device_projector=0)
    rt.FOURIER_INV(torch.rand((NZ, NA, N), device="cuda"))
elif what == "momentum":
    for _ in range(4):
        ops.momentum(vol, out, out, 0.5)
else:
    H = HipTools3D(N, 0, NZ, np.linspace(0, np.pi, NA, endpoint=False), 0.0, N, "gpu", 0, None)
    sino = torch.rand((NZ, NA, N), device="cuda")
    if what == "bpp":   # the fused gradient step on the PLANAR residual with planar staging (bp variant 3, dev flavour): the round-4 path
        ops.set_variant("bp", 3)
        res = H.residual_buffer(None)
        H.residual(vol, sino, None, "LS", None, res)
        for _ in range(3):
            H.grad_step(res, vol, out, 1e-4, True, None)
    elif what in ("bpq", "fpq"):   # the fused pair as the FISTA loop runs it: residual in the quad-interleaved layout (round 5)
        H.set_residual_layout("zquad")
        res = H.residual_buffer(None)
        for _ in range(2 if what == "fpq" else 1):
            H.residual(vol, sino, None, "LS", None, res)
        for _ in range(3 if what == "bpq" else 0):
            H.grad_step(res, vol, out, 1e-4, True, None)
    elif what.startswith("bp"):
        ops.set_variant("bp", int(what[2]))
        for _ in range(3):
            H.backward(sino, None, out=out)
    else:
        if what == "fp4":   # dev flavour: the whole-row form without the per-angle lane multipliers (pixel = lane, round 5)
            ops.set_variant("fp", 4)
        for _ in range(2):
            H.forward(vol, None, out=sino)
torch.cuda.synchronize()
