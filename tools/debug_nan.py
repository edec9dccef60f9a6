import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import tomo_oracle as O
from tomobar_amd.projector import HipTools3D
from tomobar_amd import ops
rng = np.random.default_rng(0)
for n, na, shift in ((280, 180, 0.0), (280, 180, 0.003), (256, 180, 0.0), (280, 60, 0.0), (300, 180, 0.0), (280, 90, 0.0)):
    angles = np.linspace(0, np.pi, na, endpoint=False) + shift
    H = HipTools3D(n, 0, 4, angles, 0.0, n, "gpu", 0, None)
    P = O.Projector(4, n, n, angles, 0.0, 1)
    vol = rng.standard_normal((4, n, n)).astype(np.float32)
    want = P.fp(vol)
    for v in (0, 2):
        ops.set_variant("fp", v)
        got = H.forward(torch.from_numpy(vol).cuda()).cpu().numpy()
        bad = np.argwhere(~(got == want))
        print(f"n={n} na={na} shift={shift} variant {v}: mismatches {len(bad)} angles {sorted(set(bad[:,1].tolist()))[:12]} u-range {bad[:,2].min() if len(bad) else None}..{bad[:,2].max() if len(bad) else None}")
