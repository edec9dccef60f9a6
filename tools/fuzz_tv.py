import sys, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib
from oracle import tomo_oracle as oracle
os.environ.setdefault("TOMO_MI355X_FLAVOUR", "dev")  # A/B variants and measurement switches live in libtomo_mi355x_dev.so
from tomobar_amd import ops
from tomobar_amd.regularisersCuPy import PD_TV_cupy, ROF_TV_cupy
bad = 0
for seed in range(300):
    rng = np.random.default_rng(7000 + seed)
    shape = (int(rng.integers(2, 90)), int(rng.integers(2, 100)), int(rng.integers(2, 260)))
    x = (rng.random(shape) * 0.4 + (np.indices(shape)[-1] > shape[-1] // 3) - 0.3).astype(np.float32)
    iters = int(rng.integers(1, 14))
    half, mtv, nn = bool(rng.integers(0, 2)), int(rng.integers(0, 2)), int(rng.integers(0, 2))
    lam = float(rng.choice([0.01, 0.05, 0.3]))
    want = oracle.pd_tv(x, lam, iters, mtv, nn, 8.0, half)
    for v in (21, 2, 22):   # 22 = the shipped FMA-corrected roundings (same source in both flavours)
        ops.set_variant("pdtv", v)
        got = PD_TV_cupy(torch.from_numpy(x).cuda(), lam, iters, mtv, nn, 8.0, 0, half).cpu().numpy()
        if not np.array_equal(got, want):
            bad += 1; print("MISMATCH", v, shape, iters, half, mtv, nn, np.abs(got - want).max(), flush=True)
    ops.set_variant("pdtv", 0)   # the shipped default (float32 duals relaxed: tolerance; binary16 duals exact)
    got = PD_TV_cupy(torch.from_numpy(x).cuda(), lam, iters, mtv, nn, 8.0, 0, half).cpu().numpy()
    r = np.linalg.norm((got - want).ravel().astype(np.float64)) / max(np.linalg.norm(want.ravel().astype(np.float64)), 1e-30)
    if r > 1e-5 or (half and not np.array_equal(got, want)):
        bad += 1; print("DEFAULT", shape, iters, half, mtv, nn, r, flush=True)
    ops.set_variant("pdtv", 0)
    ops.set_variant("roftv", 0)   # the shipped ROF_TV reproduces the reference's roundings
    wr = oracle.rof_tv(x, lam, iters, 0.004, half)
    got = ROF_TV_cupy(torch.from_numpy(x).cuda(), lam, iters, 0.004, 0, half).cpu().numpy()
    if not np.array_equal(got, wr):
        bad += 1; print("ROF MISMATCH", shape, iters, half, np.abs(got - wr).max(), flush=True)
print("done, bad =", bad)
