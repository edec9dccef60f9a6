"""BASELINE configs[3] per-GPU share: ADMM + ROF_TV on a 256-slice slab of 2048^2, 1500 angles (one of 4 GPUs).
Reports seconds per outer iteration and checks the result is finite.  usage: python tools/archive/probes/admm_cfg3_probe.py [nz] [iters]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import numpy as np
import torch
from tomobar_amd.methodsIR_CuPy import RecToolsIRCuPy

nz = int(sys.argv[1]) if len(sys.argv) > 1 else 256
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
n, na = 2048, 1500
angles = np.linspace(0, np.pi, na, endpoint=False)
rt = RecToolsIRCuPy(DetectorsDimH=n, DetectorsDimH_pad=0, DetectorsDimV=nz, CenterRotOffset=0.0, AnglesVec=angles,
                    ObjSize=n, device_projector=0)
# synthetic phantom: a few ellipsoids, projected with the HIP forward projector
zz, yy, xx = torch.meshgrid(torch.linspace(-1, 1, nz, device="cuda"), torch.linspace(-1, 1, n, device="cuda"),
                            torch.linspace(-1, 1, n, device="cuda"), indexing="ij")
ph = ((xx / 0.7) ** 2 + (yy / 0.9) ** 2 + (zz / 0.95) ** 2 < 1).float()
ph += 0.5 * (((xx - 0.2) / 0.2) ** 2 + ((yy + 0.1) / 0.3) ** 2 + (zz / 0.6) ** 2 < 1).float()
del zz, yy, xx
sino = rt.Atools.forward(ph.contiguous())           # [detY, angles, detX]
data = {"projection_data": sino, "data_axes_labels_order": ["detY", "angles", "detX"], "data_fidelity": "LS"}
reg = {"method": "ROF_TV", "regul_param": 0.0005, "iterations": 20, "time_marching_step": 0.001, "device_regulariser": 0}
t = []
for it in (1, iters):
    torch.cuda.synchronize()
    t0 = time.time()
    rec = rt.ADMM(data, {"iterations": it, "ADMM_rho_const": 1.0, "ADMM_relax_par": 1.7, "initialise": None}, reg)
    torch.cuda.synchronize()
    t.append(time.time() - t0)
per = (t[1] - t[0]) / max(iters - 1, 1)
err = float(torch.linalg.norm(rec - ph) / torch.linalg.norm(ph))
print(f"ADMM+ROF_TV(20) {nz}x{n}^2, {na} angles: {per:.3f} s per outer iteration ({nz/per:.1f} slices/s); "
      f"finite={bool(torch.isfinite(rec).all())} rel.err vs phantom after {iters} it = {err:.3f}; "
      f"peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
