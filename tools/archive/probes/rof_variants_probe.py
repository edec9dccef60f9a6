"""ROF_TV normalisation arithmetic: error vs the oracle on the noise inputs of tests/test_gpu_ref_tv.py and time per iteration at
1024^3 for variant 0 (relaxed), 4 (refined rsq/rcp), 5 (Markstein-corrected = the reference's roundings), 2 (compiler IEEE)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import numpy as np, torch
from oracle import tomo_oracle as O
os.environ.setdefault("TOMO_MI355X_FLAVOUR", "dev")  # A/B variants and measurement switches live in libtomo_mi355x_dev.so
from tomobar_amd import ops
from tomobar_amd.regularisersCuPy import ROF_TV_cupy
SH = [(6, 9, 13), (1, 20, 17), (12, 1, 70), (10, 11, 1), (8, 8, 8), (3, 5, 131), (24, 19), (20, 70, 150), (9, 40, 70), (33, 90, 200)]
def rel(a, b):
    a = a.astype(np.float64).ravel(); b = b.astype(np.float64).ravel(); return np.linalg.norm(a - b) / np.linalg.norm(b)
for shape in SH:
    for seed in (6, 7):
        rng = np.random.default_rng(seed)
        x = (rng.random(shape) * 0.3 + (np.indices(shape)[-1] > shape[-1] // 2)).astype(np.float32)
        want = O.rof_tv(x, 0.05, 60, 0.005, False)
        row = []
        for v in (0, 4, 5, 2):
            ops.set_variant("roftv", v)
            got = ROF_TV_cupy(torch.from_numpy(x).cuda(), 0.05, 60, 0.005, 0, False).cpu().numpy()
            row.append(f"v{v} {rel(got, want):.2e}{'=' if np.array_equal(got, want) else ' '}")
        print(f"{str(shape):16s} seed {seed}: " + "  ".join(row), flush=True)
N = 1024
vol = torch.rand((N, N, N), device="cuda"); out = torch.empty_like(vol)
for v in (0, 4, 5, 2, 0, 4, 5, 2):
    ops.set_variant("roftv", v)
    ROF_TV_cupy(vol, 0.01, 2, 0.001, 0, False, out=out); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ROF_TV_cupy(vol, 0.01, 12, 0.001, 0, False, out=out); e1.record(); torch.cuda.synchronize()
    print(f"ROF_TV v{v}: {e0.elapsed_time(e1) / 12:6.3f} ms/iter", flush=True)
