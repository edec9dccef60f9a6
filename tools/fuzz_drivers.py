"""One-off fuzz campaign at the DRIVER level (RecToolsIRCuPy.FISTA / ADMM / OSEM through the reference's dictionaries):
random geometry, detector padding, axis order, rotation-axis offsets (scalar / per angle / with a vertical component),
ordered subsets, data fidelity, regulariser and its parameters, non-negativity, circular mask, warm start -- every result
array_equal to the oracle's restatement of the same loop (PD_TV with the reference's roundings, variant 22, so that the
comparison is bit for bit).  TEST INFRASTRUCTURE (imports oracle/).

    python tools/fuzz_drivers.py --minutes 8 --seed0 0 > gpurun_out/<tag>/fuzz_drivers.txt
"""
import argparse
import os
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one_case(seed, O, torch, RecToolsIRCuPy, ops, scale=1):
    rng = np.random.default_rng(424200 + seed)
    method = str(rng.choice(["FISTA", "FISTA", "ADMM", "OSEM"]))
    two_d = rng.integers(0, 6) == 0
    nz = 1 if two_d else int(rng.integers(1, 11))
    det = int(rng.integers(24, 130)) * scale
    pad = int(rng.choice([0, 0, rng.integers(1, 12)]))
    n = det + 2 * pad
    na = int(rng.integers(12, 64))
    os_n = int(rng.integers(1, 6))
    os_n = 1 if na < 3 * os_n else os_n
    angles = float(rng.uniform(-1, 1)) + np.linspace(0, np.pi, na, endpoint=False)
    kind = int(rng.integers(0, 4))
    if kind == 0:
        cor = 0.0
    elif kind == 1:
        cor = float(np.round(rng.uniform(-3, 3), 2))
    elif kind == 2:
        cor = rng.uniform(-2, 2, na)
    else:
        cor = np.stack([rng.uniform(-2, 2, na), (0.0 if two_d else 1.0) * rng.uniform(-1.5, 1.5, na)], axis=1)
    fid = str(rng.choice(["LS", "PWLS", "KL", "SWLS"])) if method == "FISTA" else ("KL" if method == "OSEM" else str(rng.choice(["LS", "PWLS"])))
    # round 6: the data terms of the reference's removed class in the mix (FISTA only): Group-Huber ring offsets, Huber / Student's t
    extra, okw = {}, {}
    if method == "FISTA" and fid in ("LS", "PWLS") and rng.integers(0, 4) == 0:
        lam, acc = float(rng.choice([1e-4, 1e-3])), float(rng.integers(2, 9))
        extra.update(ringGH_lambda=lam, ringGH_accelerate=acc)
        okw["ring"] = {"lambda": lam, "accelerate": acc}
    if method == "FISTA" and fid != "KL" and rng.integers(0, 3) == 0:
        if rng.integers(0, 2):
            dlt = float(np.round(rng.uniform(0.05, 1.0), 3))
            extra["huber_threshold"], okw["huber"] = dlt, dlt
        else:
            dlt = float(np.round(rng.uniform(0.5, 3.0), 3))
            extra["studentst_threshold"], okw["studentst"] = dlt, dlt
    if fid == "SWLS":
        extra["beta_SWLS"], okw["beta_swls"] = 0.2, 0.2
    regm = rng.choice(["none", "PD_TV", "ROF_TV"])
    reg, full_reg = None, None
    if regm != "none":
        reg = {"method": str(regm), "regul_param": float(rng.uniform(5e-4, 5e-3)), "iterations": int(rng.integers(1, 12))}
        if regm == "PD_TV":
            reg["methodTV"] = int(rng.integers(0, 2))
            reg["PD_LipschitzConstant"] = float(rng.choice([8.0, 12.0, 16.0]))
        else:
            reg["time_marching_step"] = float(rng.uniform(5e-4, 4e-3))
        if rng.integers(0, 4) == 0:
            reg["half_precision"] = True
        full_reg = {"regul_param": 0.001, "iterations": 150, "time_marching_step": 0.005, "PD_LipschitzConstant": 12.0,
                    "methodTV": 0, **reg}
    nonneg = bool(rng.integers(0, 2))
    iters = int(rng.integers(1, 4))
    mask = None if (pad > 0 or rng.integers(0, 3)) else float(rng.choice([0.8, 0.95, 1.0]))
    warm = method != "OSEM" and rng.integers(0, 4) == 0
    # data: positive (PWLS / KL take pre-log-like data)
    vol = rng.random((nz, det, det)).astype(np.float32)
    P0 = O.Projector(nz, det, det, angles, cor if np.ndim(cor) == 0 or np.asarray(cor).ndim == 1 else cor, 1)
    sino = (P0.fp(vol) / det + np.float32(0.05) + 0.01 * rng.random((nz, na, det)).astype(np.float32)).astype(np.float32)
    P = O.Projector(nz, n, n, angles, cor, os_n)
    Lc = float(O.power_method(P, rng.standard_normal((nz, n, n)).astype(np.float32)))
    x0 = (rng.random((nz, n, n)).astype(np.float32) * 0.01) if warm else None
    desc = dict(seed=seed, method=method, nz=nz, det=det, pad=pad, na=na, os=os_n, cor_kind=kind, fid=fid, reg=reg, nonneg=nonneg,
                iters=iters, mask=mask, warm=bool(warm), two_d=bool(two_d))
    b_pad = O.pad_detector(sino, pad)
    if method == "FISTA":
        want = O.fista(P, b_pad, iters, Lc, nonneg, full_reg, fid, x0=x0, **okw)
    elif method == "ADMM":
        want = O.admm(P, b_pad, iters, Lc, 1.0, 1.6, nonneg, full_reg, fid, x0=x0)
    else:
        want = O.osem(P, b_pad, iters, nonneg, full_reg)
    if pad > 0:
        want = O.crop_recon(want, det)
    elif mask is not None:
        want = O.circular_mask(want, mask)
    # ---- the product
    order = ["detY", "angles", "detX"]
    data = sino
    if two_d:
        data, order = sino[0], ["angles", "detX"]
        if rng.integers(0, 2):
            data, order = np.ascontiguousarray(data.T), ["detX", "angles"]
    elif rng.integers(0, 2):
        data, order = np.ascontiguousarray(np.swapaxes(sino, 0, 1)), ["angles", "detY", "detX"]
    desc["order"] = order
    ops.set_variant("pdtv", 22)
    rt = RecToolsIRCuPy(det, pad, None if two_d else nz, cor, angles, det, 0, os_n if os_n > 1 else None)
    d = {"projection_data": torch.from_numpy(np.ascontiguousarray(data)).cuda(), "data_axes_labels_order": order, "data_fidelity": fid,
         **extra}
    desc["extra"] = extra
    a = {"iterations": iters, "lipschitz_const": Lc, "nonnegativity": nonneg, "recon_mask_radius": mask}
    if warm:
        a["initialise"] = torch.from_numpy(x0).cuda()
    got = getattr(rt, method)(d, a, dict(reg) if reg else None)
    torch.cuda.synchronize()
    got = got.cpu().numpy()
    ok = got.shape == want.shape and np.array_equal(got, want)
    err = None if ok else (got.shape, want.shape, float(np.abs(got - want).max()) if got.shape == want.shape else None)
    return ok, desc, err


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--minutes", type=float, default=5.0)
    ap.add_argument("--seed0", type=int, default=0)
    ap.add_argument("--only", type=int, default=None)
    ap.add_argument("--scale", type=int, default=1, help="detector width multiplier (3: up to 390 pixels -- several tiles / bricks per row)")
    args = ap.parse_args()
    import torch
    from oracle import tomo_oracle as O
    from tomobar_amd import ops
    from tomobar_amd.methodsIR_CuPy import RecToolsIRCuPy
    O.lib()
    t_end = time.time() + 60.0 * args.minutes
    seed, n_ok, bad = args.seed0, 0, []
    by_method = {}
    while time.time() < t_end:
        if args.only is not None:
            seed = args.only
        try:
            ok, desc, err = one_case(seed, O, torch, RecToolsIRCuPy, ops, args.scale)
        except Exception as e:   # noqa: BLE001
            ok, desc, err = False, {"seed": seed}, repr(e)[:400]
            traceback.print_exc(limit=3)
        finally:
            ops.set_variant("pdtv", 0)
        by_method[desc.get("method", "?")] = by_method.get(desc.get("method", "?"), 0) + 1
        if ok:
            n_ok += 1
        else:
            bad.append((desc, err))
            print("MISMATCH", desc, err, flush=True)
        seed += 1
        if args.only is not None:
            break
    print(f"seeds {args.seed0} .. {seed - 1}: {n_ok} identical to the oracle, {len(bad)} not; by method {by_method}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
