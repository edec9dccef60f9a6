"""A few LARGE odd-shaped TV cases against the oracle, bit for bit (several z-chunks, hundreds of interior waves, ragged
edges in every direction): python tools/fuzz_tv_big.py [cases]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import tomo_oracle as oracle
os.environ.setdefault("TOMO_MI355X_FLAVOUR", "dev")  # A/B variants and measurement switches live in libtomo_mi355x_dev.so
from tomobar_amd import ops
from tomobar_amd.regularisersCuPy import PD_TV_cupy, ROF_TV_cupy
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 6
bad = 0
for seed in range(n_cases):
    rng = np.random.default_rng(9100 + seed)
    shape = (int(rng.integers(75, 230)), int(rng.integers(150, 420)), int(rng.integers(250, 700)))
    x = (rng.random(shape) * 0.4 + (np.indices(shape)[-1] > shape[-1] // 3) - 0.3).astype(np.float32)
    iters = int(rng.choice([3, 6, 7, 9]))
    half, mtv, nn = bool(seed & 1), int(rng.integers(0, 2)), int(rng.integers(0, 2))
    lam = float(rng.choice([0.01, 0.05]))
    want = oracle.pd_tv(x, lam, iters, mtv, nn, 8.0, half)
    xd = torch.from_numpy(x).cuda()
    for v in (22, 21):
        ops.set_variant("pdtv", v)
        got = PD_TV_cupy(xd, lam, iters, mtv, nn, 8.0, 0, half).cpu().numpy()
        ok = np.array_equal(got, want)
        bad += not ok
        print("PD ", "ok " if ok else "MISMATCH", v, shape, iters, half, mtv, nn, float(np.abs(got - want).max()), flush=True)
    ops.set_variant("pdtv", 0)
    got = PD_TV_cupy(xd, lam, iters, mtv, nn, 8.0, 0, half).cpu().numpy()
    r = np.linalg.norm((got - want).ravel().astype(np.float64)) / np.linalg.norm(want.ravel().astype(np.float64))
    bad += r > 1e-5
    print("PD  default rel", r, flush=True)
    ops.set_variant("pdtv", 0)
    ops.set_variant("roftv", 0)
    wr = oracle.rof_tv(x, lam, iters, 0.004, half)
    got = ROF_TV_cupy(xd, lam, iters, 0.004, 0, half).cpu().numpy()
    ok = np.array_equal(got, wr)
    bad += not ok
    print("ROF", "ok " if ok else "MISMATCH", shape, iters, half, float(np.abs(got - wr).max()), flush=True)
print("done, bad =", bad)
