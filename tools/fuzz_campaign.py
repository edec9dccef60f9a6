"""One-off fuzz campaign on the GPU: the seeded random-geometry parity tests of tests/test_gpu_parity.py driven over seeds
the suite does not hold, plus random geometries through the fused sub-iteration tests (planar and quad-interleaved
residual).  Every comparison is array_equal against the oracle, as in the suite.  TEST INFRASTRUCTURE (imports oracle/).

    python tools/fuzz_campaign.py --minutes 8 --seed0 100 > gpurun_out/<tag>/fuzz.txt
"""
import argparse
import os
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def random_geometry(rng, min_os=1):
    """(nz, n, nu, na, cor, os): sizes off every tile multiple, detector wider / narrower than the grid, scalar or per-angle
    offsets, subsets with trimmed tails."""
    kind = int(rng.integers(0, 4))
    if kind == 0:      # small
        nz, n, na = int(rng.integers(1, 12)), int(rng.integers(8, 80)), int(rng.integers(3, 40))
    elif kind == 1:    # whole-row projector form (wide detector), few slices
        nz, n, na = int(rng.integers(1, 7)), int(rng.integers(500, 900)), int(rng.integers(5, 24))
    elif kind == 2:    # many slices (several z-batches, ragged last quad)
        nz, n, na = int(rng.integers(13, 50)), int(rng.integers(20, 140)), int(rng.integers(8, 48))
    else:              # dense angle sets on a mid-sized grid
        nz, n, na = int(rng.integers(2, 9)), int(rng.integers(150, 330)), int(rng.integers(100, 300))
    nu = max(8, int(n * rng.uniform(0.8, 1.3)))
    os_n = int(rng.integers(min_os, 7))
    na = max(na, 2 * os_n + 1)
    cor = "vec" if rng.integers(0, 3) == 0 else float(np.round(rng.uniform(-0.05, 0.05) * nu, 2))
    return (nz, n, nu, na, cor, os_n)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--minutes", type=float, default=5.0)
    ap.add_argument("--seed0", type=int, default=100)
    ap.add_argument("--only-dense", action="store_true")
    ap.add_argument("--skip-dense", action="store_true")
    ap.add_argument("--slabs", action="store_true", help="only the z-slab TV drivers: random shapes, splits into 2..5 slabs, "
                    "iteration counts, schedules (lockstep / boundary-first), float32 and binary16 duals")
    args = ap.parse_args()
    import torch
    import test_gpu_parity as T
    from oracle import tomo_oracle as O
    from tomobar_amd import ops
    O.lib()
    t_end = time.time() + 60.0 * args.minutes
    counts, failures = {}, []
    seed = args.seed0

    def run(name, fn, *a):
        counts[name] = counts.get(name, 0) + 1
        try:
            fn(*a)
        except Exception as e:   # noqa: BLE001 -- the campaign records and goes on
            shown = a[2:] if (len(a) > 2 and a[0] is O) else a
            failures.append((name, shown, repr(e)[:300]))
            print(f"FAIL {name} {shown}: {e!r}"[:600], flush=True)
            traceback.print_exc(limit=2)
        finally:
            for k in ("bp", "fp", "roftv"):
                ops.set_variant(k, 0)
            ops.set_variant("pdtv", 0)
            torch.cuda.synchronize()

    while time.time() < t_end:
        rng = np.random.default_rng(77000 + seed)
        g = random_geometry(rng)
        g3 = random_geometry(rng, min_os=3)   # the fused tests walk subsets 0..2
        if args.slabs:
            import test_gpu_slab as S
            world = int(rng.integers(2, 6))
            nz = int(rng.integers(4 * world, 12 * world))
            shape = (nz, int(rng.integers(9, 80)), int(rng.integers(40, 260)))
            half = bool(rng.integers(0, 2))
            variant = [22, "ranges", 0][int(rng.integers(0, 3))]
            iters = (int(rng.integers(1, 14)),)
            run("pdtv_slabs", S.run_pdtv_slabs, world, half, variant, shape, iters, int(rng.integers(0, 2)), int(rng.integers(0, 2)), seed)
            run("roftv_slabs", S.run_roftv_slabs, world, half, shape, int(rng.integers(1, 9)), seed)
            seed += 1
            continue
        if args.only_dense:
            from tomobar_amd import _lib
            with _lib.use_flavour("dev"):
                run("dense_angle_form_random_geometries", T.test_forward_projection_dense_angle_form_random_geometries, O, ops, seed)
            seed += 1
            continue
        run("projector_pair_random_geometries", T.test_projector_pair_random_geometries, O, ops, seed, (0,))
        if not args.skip_dense:
            from tomobar_amd import _lib
            with _lib.use_flavour("dev"):   # the test forces the dense-angle form wherever it applies (variant 3 of the dev build)
                run("dense_angle_form_random_geometries", T.test_forward_projection_dense_angle_form_random_geometries, O, ops, seed)
        run("tv_random_shapes", T.test_tv_random_shapes, O, ops, seed, "shipped")
        run("fused_residual_and_gradient_steps", T.test_fused_residual_and_gradient_steps, O, ops, g3, (0,))
        run("quad_interleaved_residual_layout", T.test_quad_interleaved_residual_layout, O, ops, g, 0)
        run("backprojection_vs_oracle", T.test_backprojection_vs_oracle, O, ops, g, 0)
        run("forward_projection_vs_oracle", T.test_forward_projection_vs_oracle, O, ops, g, 0)
        seed += 1
    print(f"seeds {args.seed0} .. {seed - 1}: " + ", ".join(f"{k} x{v}" for k, v in counts.items()))
    print(f"failures: {len(failures)}")
    for f in failures:
        print("  ", f)
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
