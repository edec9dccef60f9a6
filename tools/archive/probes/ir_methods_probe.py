"""Per-iteration time of every iterative driver on one geometry (sanity: no hidden host round trips).
usage: python tools/archive/probes/ir_methods_probe.py [n] [nz] [angles]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import numpy as np
import torch
from tomobar_amd.methodsIR_CuPy import RecToolsIRCuPy

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
nz = int(sys.argv[2]) if len(sys.argv) > 2 else 512
na = int(sys.argv[3]) if len(sys.argv) > 3 else 360
angles = np.linspace(0, np.pi, na, endpoint=False)
rt = RecToolsIRCuPy(n, 0, nz, 0.0, angles, n, device_projector=0)
rt_os = RecToolsIRCuPy(n, 0, nz, 0.0, angles, n, device_projector=0, OS_number=6)
vol = torch.rand((nz, n, n), device="cuda") + 0.1
sino = rt.Atools.forward(vol)
data = {"projection_data": sino, "data_axes_labels_order": ["detY", "angles", "detX"]}


def timed(name, fn, iters):
    fn(1)
    torch.cuda.synchronize()
    t0 = time.time(); fn(1); torch.cuda.synchronize(); t1 = time.time(); fn(1 + iters); torch.cuda.synchronize(); t2 = time.time()
    print(f"{name:28s} {((t2 - t1) - (t1 - t0)) / iters * 1e3:8.1f} ms / iteration")


timed("Landweber", lambda k: rt.Landweber(data, {"iterations": k, "tau_step_lanweber": 1e-5}), 4)
timed("SIRT", lambda k: rt.SIRT(data, {"iterations": k}), 4)
timed("CGLS", lambda k: rt.CGLS(data, {"iterations": k}), 4)
timed("FISTA (no prox)", lambda k: rt.FISTA(data, {"iterations": k, "lipschitz_const": 5e4}), 4)
timed("FISTA-OS6 + ROF_TV(10)", lambda k: rt_os.FISTA(data, {"iterations": k, "lipschitz_const": 1e4},
                                                       {"method": "ROF_TV", "regul_param": 1e-4, "iterations": 10,
                                                        "time_marching_step": 1e-3}), 2)
timed("ADMM + PD_TV(10)", lambda k: rt.ADMM(data, {"iterations": k, "ADMM_rho_const": 1.0, "ADMM_relax_par": 1.7},
                                             {"method": "PD_TV", "regul_param": 1e-4, "iterations": 10}), 2)
timed("OSEM (OS6)", lambda k: rt_os.OSEM(data, {"iterations": k}), 2)
