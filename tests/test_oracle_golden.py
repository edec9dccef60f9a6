"""Pins the CPU oracle against the fixtures under tests/golden/ (generated from the REFERENCE's own sources by the
committed scripts make_tv_golden.py / make_outer_golden.py; the .npz files hold data only)."""
import os

import numpy as np
import pytest

TOL = 1e-5  # relative L2, the tolerance BASELINE.json's north_star states


def rel(a, b):
    return float(np.linalg.norm(a.ravel().astype(np.float64) - b.ravel()) / max(np.linalg.norm(b.ravel()), 1e-30))


@pytest.fixture(scope="module")
def tv(golden_dir):
    return np.load(os.path.join(golden_dir, "tv_golden.npz"))


@pytest.fixture(scope="module")
def outer(golden_dir):
    return np.load(os.path.join(golden_dir, "outer_golden.npz"))


def tv_cases(tv, kind):
    return sorted(int(k.split("_")[1]) for k in tv.files if k.startswith(kind + "_") and k.endswith("_meta"))


def test_pdtv_oracle_vs_reference_kernels(oracle, tv):
    worst = 0.0
    for cid in tv_cases(tv, "pd"):
        in_id, half, mtv, nn, iters, lam, lip = tv[f"pd_{cid}_meta"]
        x = tv[f"in_{int(in_id)}"]
        if nn:
            x = (x - 0.6).astype(np.float32)
        got = oracle.pd_tv(x, float(lam), int(iters), int(mtv), int(nn), float(lip), bool(half))
        for build in ("off", "fma"):
            r = rel(got, tv[f"pd_{cid}_{build}"])
            worst = max(worst, r)
            assert r < TOL, (cid, build, r)
        assert got.shape == tv[f"pd_{cid}_fma"].shape
    print("PD_TV worst rel-L2 vs reference kernels:", worst)


def test_roftv_oracle_vs_reference_kernels(oracle, tv):
    worst = 0.0
    for cid in tv_cases(tv, "rof"):
        in_id, half, iters, lam, tms = tv[f"rof_{cid}_meta"]
        x = tv[f"in_{int(in_id)}"]
        got = oracle.rof_tv(x, float(lam), int(iters), float(tms), bool(half))
        for build in ("off", "fma"):
            r = rel(got, tv[f"rof_{cid}_{build}"])
            worst = max(worst, r)
            assert r < TOL, (cid, build, r)
    print("ROF_TV worst rel-L2 vs reference kernels:", worst)


OUTER_CASES = {
    # name: (method, projector kwargs, data kwargs, algorithm kwargs, regularisation)
    "fista_plain": ("fista", {}, {}, dict(iterations=6, L="L_full"), None),
    "fista_os4_pdtv": ("fista", dict(os_number=4), {}, dict(iterations=3, L="L_os4", nonnegativity=True),
                       dict(method="PD_TV", regul_param=0.002, iterations=8)),
    "fista_os7_roftv": ("fista", dict(os_number=7), {}, dict(iterations=2, L="L_os7"),
                        dict(method="ROF_TV", regul_param=0.002, iterations=8, time_marching_step=0.002)),
    "fista_os4_pdtv_half_aniso": ("fista", dict(os_number=4), {}, dict(iterations=2, L="L_os4"),
                                  dict(method="PD_TV", regul_param=0.002, iterations=6, methodTV=1, half_precision=True)),
    "fista_pwls_os4": ("fista", dict(os_number=4), dict(fidelity="PWLS"), dict(iterations=3, L="L_os4"), None),
    "admm_plain": ("admm", {}, {}, dict(iterations=5, L="L_full"), None),
    "admm_pdtv": ("admm", {}, {}, dict(iterations=4, L="L_full", rho=2.0, relax=1.5, nonnegativity=True),
                  dict(method="PD_TV", regul_param=0.004, iterations=8)),
    "admm_os4_roftv": ("admm", dict(os_number=4), {}, dict(iterations=4, L="L_os4"),
                       dict(method="ROF_TV", regul_param=0.004, iterations=8, time_marching_step=0.002)),
}
REG_DEFAULTS = dict(regul_param=0.001, iterations=150, time_marching_step=0.005, PD_LipschitzConstant=12.0, methodTV=0)


@pytest.mark.parametrize("name", sorted(OUTER_CASES))
def test_outer_loops_oracle_vs_reference_python(oracle, outer, name):
    """The reference's FISTA / ADMM loops (run unmodified by make_outer_golden.py) vs the oracle's numpy restatement,
    both on the oracle projector: pins the outer-loop algebra."""
    method, pk, dk, ak, reg = OUTER_CASES[name]
    sino, angles = outer["sino"], outer["angles"]
    nz, _, n = sino.shape
    P = oracle.Projector(nz, n, n, angles, 0.0, pk.get("os_number", 1))
    if reg is not None:
        reg = {**REG_DEFAULTS, **reg}
    L = float(outer[ak["L"]])
    if method == "fista":
        got = oracle.fista(P, sino, ak["iterations"], L, ak.get("nonnegativity", False), reg, dk.get("fidelity", "LS"))
    else:
        got = oracle.admm(P, sino, ak["iterations"], L, ak.get("rho", 1.0), ak.get("relax", 1.6),
                          ak.get("nonnegativity", False), reg, dk.get("fidelity", "LS"))
    got = oracle.circular_mask(got, 1.0)
    assert rel(got, outer[name]) < TOL, rel(got, outer[name])


def test_outer_mask_crop_kl_warm(oracle, outer):
    sino, angles = outer["sino"], outer["angles"]
    nz, _, n = sino.shape
    P = oracle.Projector(nz, n, n, angles)
    L = float(outer["L_full"])
    got = oracle.circular_mask(oracle.fista(P, sino, 5, L, True), 0.85)
    assert rel(got, outer["fista_nonneg_mask"]) < TOL
    got = oracle.circular_mask(oracle.fista(P, outer["raw_kl"], 3, L, True, None, "KL", outer["x0_kl"]), 1.0)
    assert rel(got, outer["fista_kl"]) < TOL
    got = oracle.circular_mask(oracle.fista(P, sino, 2, L, False, None, "LS", outer["fista_plain"]), 1.0)
    assert rel(got, outer["fista_warm"]) < TOL
    # centre of rotation + axis permutation
    Pc = oracle.Projector(nz, n, n, angles, 1.5)
    got = oracle.circular_mask(oracle.fista(Pc, sino, 4, L), 1.0)
    assert rel(got, outer["fista_perm_cor"]) < TOL
    # padded detector: larger grid, cropped back, never masked
    Pp = oracle.Projector(nz, n + 8, n + 8, angles, 0.0, 4)
    got = oracle.crop_recon(oracle.fista(Pp, oracle.pad_detector(sino, 4), 3, float(outer["L_pad_os4"])), n)
    assert rel(got, outer["fista_pad_os4"]) < TOL
    # 2D input: one slice of the 3D geometry, TV runs the 2D kernels
    P2 = oracle.Projector(1, n, n, angles, 0.0, 4)
    reg = {**REG_DEFAULTS, "method": "PD_TV", "regul_param": 0.002, "iterations": 8}
    got = oracle.circular_mask(oracle.fista(P2, sino[1:2], 3, float(outer["L_os4"]), False, reg), 1.0)
    assert rel(got, outer["fista_2d_os4_pdtv"]) < TOL
    got = oracle.circular_mask(oracle.admm(Pp.__class__(nz, n, n, angles, 0.0, 4), sino, 3, float(outer["L_os4"]),
                                           fidelity="PWLS"), 0.9)
    assert rel(got, outer["admm_os4_pwls"]) < TOL


def test_power_method_vs_reference(oracle, outer):
    sino, angles = outer["sino"], outer["angles"]
    nz, _, n = sino.shape
    x = np.random.default_rng(3).standard_normal((nz, n, n)).astype(np.float32)
    for key, os_n in (("L_full", 1), ("L_os4", 4), ("L_os7", 7)):
        P = oracle.Projector(nz, n, n, angles, 0.0, os_n)
        np.testing.assert_allclose(oracle.power_method(P, x), float(outer[key]), rtol=2e-5)


OSEM_CASES = {
    # name: (os_number, iterations, mask radius, nonnegativity, regularisation)
    "mlem": (1, 3, 1.0, False, None),
    "osem_os4": (4, 2, 1.0, False, None),
    "osem_os7_mask": (7, 1, 0.9, False, None),
    "osem_os4_pdtv": (4, 2, 1.0, True, dict(method="PD_TV", regul_param=0.002, iterations=6)),
    "mlem_roftv": (1, 2, 1.0, False, dict(method="ROF_TV", regul_param=0.002, iterations=5, time_marching_step=0.002)),
}


@pytest.mark.parametrize("name", sorted(OSEM_CASES))
def test_osem_oracle_vs_reference_python(oracle, golden_dir, name):
    """The reference's OSEM loop (methodsIR_CuPy.py:587-667, run unmodified by make_osem_golden.py) vs the oracle's."""
    g = np.load(os.path.join(golden_dir, "osem_golden.npz"))
    os_n, iters, radius, nonneg, reg = OSEM_CASES[name]
    sino, angles = g["sino"], g["angles"]
    nz, _, n = sino.shape
    P = oracle.Projector(nz, n, n, angles, 0.0, os_n)
    if reg is not None:
        reg = {**REG_DEFAULTS, **reg}
    got = oracle.circular_mask(oracle.osem(P, sino, iters, nonneg, reg), radius)
    assert rel(got, g[name]) < TOL, rel(got, g[name])
