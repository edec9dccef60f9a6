"""Robust data terms (second half): ``_data_["huber_threshold"]`` and ``_data_["studentst_threshold"]``.

Keys of the reference's removed RecToolsIR class (Demos/methods_IR_legacy/DemoFISTA_artifacts2D.py:197,263,307,348; listed as
supported data fidelities in docs/source/introduction/about.rst:38).  This reference version has no implementation
(supp/dicts.py:85-88), so parity is FORMULA-LEVEL, UNPINNED: the oracle restates the models (gradient of the Huber function;
gradient of log(delta^2 + r^2)), the CPU tests check the restatement's defining properties, the GPU tests check the HIP path
(fused forward-projection epilogue, both residual layouts, and the element-wise kernel of the ring-term paths) against it
bit for bit."""
import numpy as np
import pytest


def _zinger_data(O, nz=3, n=48, na=60, seed=7):
    angles = np.linspace(0, np.pi, na, endpoint=False)
    clean = (O.shepp_logan_sino(n, nz, n, angles) / n).astype(np.float32)
    rng = np.random.default_rng(seed)
    hit = rng.random(clean.shape) < 0.004          # zingers: isolated, very bright detector events
    noisy = clean.copy()
    noisy[hit] += np.float32(25.0)
    return angles, clean, noisy.astype(np.float32)


def test_robust_weight_formulas(oracle):
    O = oracle
    r = np.linspace(-9, 9, 73, dtype=np.float32)
    h = O.robust_weight(r, huber=2.5)
    inside = np.abs(r) <= 2.5
    assert np.array_equal(h[inside], r[inside])                       # quadratic zone: the residual itself
    assert np.allclose(np.abs(h[~inside]), 2.5, rtol=1e-6) and np.array_equal(np.sign(h), np.sign(r))   # linear zone: clipped to +-delta
    s = O.robust_weight(r, studentst=2.5)
    assert np.allclose(s, 2 * r / (2.5 ** 2 + r.astype(np.float64) ** 2), rtol=1e-6)
    # gradient of log(delta^2 + r^2) by central differences
    f = lambda x: np.log(2.5 ** 2 + x ** 2)
    assert np.allclose(s, (f(r.astype(np.float64) + 1e-5) - f(r.astype(np.float64) - 1e-5)) / 2e-5, atol=1e-5)
    assert np.array_equal(O.robust_weight(r), r)


def test_huber_and_studentst_suppress_zingers(oracle):
    """outliers in the data pull a least-squares reconstruction away from the clean one; both robust terms pull it back"""
    O = oracle
    angles, clean, noisy = _zinger_data(O)
    nz, na, n = clean.shape
    P = O.Projector(nz, n, n, angles, 0.0, 4)
    L = O.power_method(P, np.random.default_rng(0).standard_normal((nz, n, n)).astype(np.float32))
    ref = O.fista(P, clean, 10, L, True)
    plain = O.fista(P, noisy, 10, L, True)
    hub = O.fista(P, noisy, 10, L, True, huber=0.5)
    e_plain = np.linalg.norm(plain - ref) / np.linalg.norm(ref)
    e_hub = np.linalg.norm(hub - ref) / np.linalg.norm(ref)
    assert e_hub < 0.5 * e_plain, (e_plain, e_hub)
    # an infinite Huber threshold is exactly the plain algorithm
    assert np.array_equal(O.fista(P, noisy, 2, L, True, huber=3e38), O.fista(P, noisy, 2, L, True))
    # Student's t scales the gradient by 2/delta^2 near zero residual: compare at the matching step size
    d = 4.0
    st = O.fista(P, noisy, 10, L * 2.0 / d ** 2, True, studentst=d)
    e_st = np.linalg.norm(st - ref) / np.linalg.norm(ref)
    assert e_st < 0.7 * e_plain, (e_plain, e_st)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["huber_ls_os_pdtv", "huber_pwls", "studentst_ls_os", "studentst_pwls_2d", "huber_swls_os",
                                  "huber_gh_pwls", "studentst_gh_os", "huber_ls_vertical"])
def test_robust_terms_hip_vs_oracle(oracle, case):
    """RecToolsIRCuPy.FISTA with the robust-term keys against the oracle's restatement, bit for bit"""
    import torch
    from tomobar_amd import ops
    from tomobar_amd.methodsIR_CuPy import RecToolsIRCuPy
    O = oracle
    angles, clean, noisy = _zinger_data(O, nz=5, n=72, na=45)
    nz, na, n = noisy.shape
    cor = 0.0
    if case.endswith("_vertical"):   # vertical CoR component: unfused residual + the element-wise re-weighting kernel
        cor = np.stack([np.linspace(-1.0, 1.5, na), 1.3 * np.cos(np.linspace(0.3, 2.9, na))], axis=1)
    two_d = case.endswith("_2d")
    if two_d:
        noisy, nz = noisy[2:3], 1
    os_n = 5 if "_os" in case else 1
    fid = "PWLS" if "pwls" in case else ("SWLS" if "swls" in case else "LS")
    data = np.abs(noisy) + np.float32(0.05) if fid != "LS" else noisy
    P = O.Projector(nz, n, n, angles, cor, os_n)
    L = O.power_method(P, np.random.default_rng(0).standard_normal((nz, n, n)).astype(np.float32))
    kw, d = {}, {}
    if case.startswith("huber"):
        kw["huber"], d["huber_threshold"] = 0.4, 0.4
    else:
        kw["studentst"], d["studentst_threshold"] = 1.5, 1.5
        L = L * 2.0 / 1.5 ** 2
    ring = None
    if "_gh" in case:
        ring = {"lambda": 1e-4, "accelerate": 4}
        d.update(ringGH_lambda=1e-4, ringGH_accelerate=4)
    reg = full_reg = None
    if "pdtv" in case:
        reg = {"method": "PD_TV", "regul_param": 0.002, "iterations": 6}
        full_reg = {"regul_param": 0.001, "iterations": 150, "time_marching_step": 0.005, "PD_LipschitzConstant": 12.0,
                    "methodTV": 0, **reg}
    d.update({"projection_data": torch.from_numpy(np.ascontiguousarray(data[0] if two_d else data)).cuda(),
              "data_axes_labels_order": ["angles", "detX"] if two_d else ["detY", "angles", "detX"], "data_fidelity": fid})
    ops.set_variant("pdtv", 22)   # the data terms are under test: PD_TV with the reference's roundings keeps it bit for bit
    rt = RecToolsIRCuPy(n, 0, None if two_d else nz, cor, angles, n, 0, os_n if os_n > 1 else None)
    got = rt.FISTA(d, {"iterations": 3, "lipschitz_const": L, "nonnegativity": True, "recon_mask_radius": None}, reg)
    torch.cuda.synchronize()
    want = O.fista(P, data, 3, L, True, full_reg, fid, ring=ring, beta_swls=0.1, **kw)
    g = got.cpu().numpy().reshape(want.shape)
    assert np.isfinite(want).all() and np.abs(want).max() > 0
    assert np.array_equal(g, want), float(np.abs(g - want).max())
    # and the term does something: the plain run differs
    plain = O.fista(P, data, 3, L, True, full_reg, fid, ring=ring, beta_swls=0.1)
    assert not np.array_equal(plain, want)


@pytest.mark.gpu
def test_robust_residual_both_layouts_and_elementwise_kernel(oracle):
    """tomo_fp3d_residual_robust in the planar and the quad-interleaved layout (nz not a multiple of 4) and tomo_sino_robust on
    an existing residual give the oracle's re-weighted residual, bit for bit"""
    import torch
    from tomobar_amd.projector import HipTools3D
    O = oracle
    nz, n, na = 6, 40, 21
    angles = np.linspace(0, np.pi, na, endpoint=False)
    P = O.Projector(nz, n, n, angles, 0.5, 3)
    H = HipTools3D(n, 0, nz, angles, 0.5, n, "gpu", 0, 3)
    rng = np.random.default_rng(2)
    x = rng.random((nz, n, n)).astype(np.float32)
    b = (rng.random((nz, na, n)) * 30).astype(np.float32)
    w = O.pwls_weights(b)
    xd, bd, wd = (torch.from_numpy(a).cuda() for a in (x, b, w))
    for s in range(3):
        idx = P.subsets[s]
        raw = (P.fp(x, s) - b[:, idx]) * w[:, idx]
        for mode, delta, kw in (("huber", 3.0, {"huber": 3.0}), ("studentst", 2.0, {"studentst": 2.0})):
            want = O.robust_weight(raw, **kw)
            for layout in ("planar", "zquad"):
                H.set_residual_layout(layout)
                try:
                    res = H.residual_buffer(s)
                    H.residual(xd, bd, wd, "PWLS", s, res, robust=(mode, delta))
                    assert np.array_equal(H.residual_as_planar(res, s).cpu().numpy(), want), (s, mode, layout)
                    plain = H.residual_buffer(s)
                    H.residual(xd, bd, wd, "PWLS", s, plain)
                    H.robust_apply(plain, mode, delta)
                    assert torch.equal(plain, res), (s, mode, layout)
                finally:
                    H.set_residual_layout("planar")
    with pytest.raises(ValueError):
        H.residual(xd, bd, None, "KL", 0, H.residual_buffer(0), robust=("huber", 1.0))
    with pytest.raises(ValueError):
        H.residual(xd, bd, None, "LS", 0, H.residual_buffer(0), robust=("huber", -1.0))


@pytest.mark.gpu
def test_robust_term_key_validation():
    import torch
    from tomobar_amd.methodsIR_CuPy import RecToolsIRCuPy
    n, na = 24, 12
    rt = RecToolsIRCuPy(n, 0, 2, 0.0, np.linspace(0, np.pi, na, endpoint=False), n, 0, None)
    b = torch.rand((2, na, n), device="cuda") + 0.1
    alg = {"iterations": 1, "lipschitz_const": 1e3}
    with pytest.raises(ValueError):
        rt.FISTA({"projection_data": b, "data_fidelity": "KL", "huber_threshold": 1.0}, dict(alg))
    with pytest.raises(ValueError):
        rt.FISTA({"projection_data": b, "huber_threshold": 1.0, "studentst_threshold": 1.0}, dict(alg))
    with pytest.raises(ValueError):
        rt.FISTA({"projection_data": b, "studentst_threshold": 0.0}, dict(alg))
    with pytest.raises(ValueError):
        rt.ADMM({"projection_data": b, "huber_threshold": 1.0}, dict(alg))
    d = {"projection_data": b}
    rt.FISTA(d, dict(alg))
    assert d["huber_threshold"] is None and d["studentst_threshold"] is None   # defaults populated
