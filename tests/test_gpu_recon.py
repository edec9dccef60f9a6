"""End-to-end parity of the drop-in classes: RecToolsIRCuPy.{FISTA, ADMM, powermethod, ...} on the MI355X against
(a) tests/golden/outer_golden.npz -- outputs of the REFERENCE's own Python loops (see make_outer_golden.py) -- and
(b) the CPU oracle on the same inputs.  The tests are written the way the reference's tests/test_RecToolsIRCuPy.py
are: build the class from the geometry, fill the three dictionaries, call the method, check shape/dtype/values."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

TOL = 1e-5


def rel(a, b):
    a = np.asarray(a, dtype=np.float64).ravel()
    b = np.asarray(b, dtype=np.float64).ravel()
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def host(t):
    torch.cuda.synchronize()
    return t.detach().cpu().numpy()


@pytest.fixture(scope="module")
def outer(golden_dir):
    return np.load(os.path.join(golden_dir, "outer_golden.npz"))


@pytest.fixture(scope="module")
def geom(outer):
    sino, angles = outer["sino"], outer["angles"]
    nz, na, n = sino.shape
    return dict(sino=sino, angles=angles, nz=nz, n=n, na=na)


def make(geom, pad=0, detV="3d", cor=0.0, os_number=None):
    from tomobar_amd.methodsIR_CuPy import RecToolsIRCuPy
    return RecToolsIRCuPy(DetectorsDimH=geom["n"], DetectorsDimH_pad=pad,
                          DetectorsDimV=geom["nz"] if detV == "3d" else None, CenterRotOffset=cor,
                          AnglesVec=geom["angles"], ObjSize=geom["n"], device_projector=0, OS_number=os_number)


def data_dict(geom, **extra):
    d = {"projection_data": torch.from_numpy(geom["sino"]).cuda(), "data_axes_labels_order": ["detY", "angles", "detX"]}
    d.update(extra)
    return d


CASES = {
    "fista_plain": ("FISTA", {}, {}, dict(iterations=6, lipschitz_const="L_full"), None),
    "fista_nonneg_mask": ("FISTA", {}, {}, dict(iterations=5, lipschitz_const="L_full", nonnegativity=True,
                                                recon_mask_radius=0.85), None),
    "fista_os4_pdtv": ("FISTA", dict(os_number=4), {}, dict(iterations=3, lipschitz_const="L_os4", nonnegativity=True),
                       dict(method="PD_TV", regul_param=0.002, iterations=8)),
    "fista_os7_roftv": ("FISTA", dict(os_number=7), {}, dict(iterations=2, lipschitz_const="L_os7"),
                        dict(method="ROF_TV", regul_param=0.002, iterations=8, time_marching_step=0.002)),
    "fista_os4_pdtv_half_aniso": ("FISTA", dict(os_number=4), {}, dict(iterations=2, lipschitz_const="L_os4"),
                                  dict(method="PD_TV", regul_param=0.002, iterations=6, methodTV=1, half_precision=True)),
    "fista_pwls_os4": ("FISTA", dict(os_number=4), dict(data_fidelity="PWLS"), dict(iterations=3, lipschitz_const="L_os4"), None),
    "fista_pad_os4": ("FISTA", dict(os_number=4, pad=4), {}, dict(iterations=3, lipschitz_const="L_pad_os4",
                                                                  recon_mask_radius=2.0), None),
    "admm_plain": ("ADMM", {}, {}, dict(iterations=5, lipschitz_const="L_full"), None),
    "admm_pdtv": ("ADMM", {}, {}, dict(iterations=4, lipschitz_const="L_full", ADMM_rho_const=2.0, ADMM_relax_par=1.5,
                                       nonnegativity=True), dict(method="PD_TV", regul_param=0.004, iterations=8)),
    "admm_os4_roftv": ("ADMM", dict(os_number=4), {}, dict(iterations=4, lipschitz_const="L_os4"),
                       dict(method="ROF_TV", regul_param=0.004, iterations=8, time_marching_step=0.002)),
    "admm_os4_pwls": ("ADMM", dict(os_number=4), dict(data_fidelity="PWLS"),
                      dict(iterations=3, lipschitz_const="L_os4", recon_mask_radius=0.9), None),
}


@pytest.mark.parametrize("name", sorted(k for k, c in CASES.items() if c[4] is not None and c[4]["method"] == "PD_TV"))
def test_exact_pdtv_against_reference_python_loops(outer, geom, name):
    """The PD_TV reconstructions once more with the reference's PD_TV roundings (variant 22), against the reference's own
    loops (test_against_reference_python_loops runs them as shipped)."""
    from tomobar_amd import ops
    ops.set_variant("pdtv", 22)
    test_against_reference_python_loops(outer, geom, name)


@pytest.mark.parametrize("name", sorted(CASES))
def test_against_reference_python_loops(outer, geom, name):
    method, mk, dk, ak, reg = CASES[name]
    rt = make(geom, **mk)
    alg = dict(ak)
    alg["lipschitz_const"] = float(outer[alg["lipschitz_const"]])
    reg_in = None if reg is None else dict(reg)
    rec = getattr(rt, method)(data_dict(geom, **dk), alg, reg_in)
    got = host(rec)
    assert got.dtype == np.float32
    assert got.shape == (geom["nz"], geom["n"], geom["n"])
    assert rel(got, outer[name]) < TOL, rel(got, outer[name])
    if reg is not None and method == "ADMM":
        assert reg_in["regul_param"] == reg["regul_param"], "caller's dictionary must not be rewritten"


def test_kl_warm_start_axis_swap_cor_2d(outer, geom):
    L = float(outer["L_full"])
    # KL fidelity with a warm start (pre-log data)
    rt = make(geom)
    d = {"projection_data": torch.from_numpy(outer["raw_kl"]).cuda(),
         "data_axes_labels_order": ["detY", "angles", "detX"], "data_fidelity": "KL"}
    x0 = torch.from_numpy(outer["x0_kl"]).cuda()
    rec = rt.FISTA(d, {"iterations": 3, "lipschitz_const": L, "initialise": x0, "nonnegativity": True})
    assert rel(host(rec), outer["fista_kl"]) < TOL
    assert np.array_equal(host(x0), outer["x0_kl"]), "the warm-start array must not be modified"
    # warm start from a previous reconstruction
    rec = make(geom).FISTA(data_dict(geom), {"iterations": 2, "lipschitz_const": L,
                                             "initialise": torch.from_numpy(outer["fista_plain"]).cuda()})
    assert rel(host(rec), outer["fista_warm"]) < TOL
    # data given as [angles, detY, detX] + centre-of-rotation offset (numpy input is accepted too)
    d = {"projection_data": np.ascontiguousarray(np.swapaxes(geom["sino"], 0, 1)),
         "data_axes_labels_order": ["angles", "detY", "detX"]}
    rec = make(geom, cor=1.5).FISTA(d, {"iterations": 4, "lipschitz_const": L})
    assert rel(host(rec), outer["fista_perm_cor"]) < TOL
    assert d["data_axes_labels_order"] is None and d["data_fidelity"] == "LS"  # dicts are populated in place
    # 2D input [angles, detX] -> output [1, N, N]
    d2 = {"projection_data": torch.from_numpy(np.ascontiguousarray(geom["sino"][1])).cuda(),
          "data_axes_labels_order": ["angles", "detX"]}
    rec = make(geom, detV=None, os_number=4).FISTA(d2, {"iterations": 3, "lipschitz_const": float(outer["L_os4"])},
                                                   {"method": "PD_TV", "regul_param": 0.002, "iterations": 8})
    assert tuple(rec.shape) == (1, geom["n"], geom["n"])
    assert rel(host(rec), outer["fista_2d_os4_pdtv"]) < TOL


def test_power_method_against_reference(outer, geom):
    for key, os_n in (("L_full", None), ("L_os4", 4), ("L_os7", 7)):
        rt = make(geom, os_number=os_n)
        rt.power_seed = 1
        lc = rt.powermethod({"projection_data": None})
        assert isinstance(lc, float)
        np.testing.assert_allclose(lc, float(outer[key]), rtol=2e-5)
    rt = make(geom, os_number=4, pad=4)
    np.testing.assert_allclose(rt.powermethod({"projection_data": None}), float(outer["L_pad_os4"]), rtol=2e-5)


def test_lipschitz_computed_when_absent_and_bad_initialise(geom, capsys):
    rt = make(geom, os_number=4)
    rec = rt.FISTA(data_dict(geom), {"iterations": 1, "initialise": torch.zeros((2, 3, 4), device="cuda")})
    assert "incorrect dimensions" in capsys.readouterr().out
    assert tuple(rec.shape) == (geom["nz"], geom["n"], geom["n"])


def test_errors_match_reference(geom):
    from tomobar_amd.methodsIR_CuPy import RecToolsIRCuPy
    rt = make(geom)
    with pytest.raises(NameError):
        rt.FISTA({"projection_data": None})
    with pytest.raises(ValueError):
        rt.FISTA(data_dict(geom, data_fidelity="L2"))
    with pytest.raises(ValueError):
        rt.FISTA(data_dict(geom), {"nonnegativity": "yes"})
    with pytest.raises(NameError):
        make(geom, os_number=3).SIRT(data_dict(geom))
    with pytest.raises(ValueError):
        RecToolsIRCuPy(0, 0, 4, 0.0, geom["angles"], 16)
    with pytest.raises(ValueError):
        RecToolsIRCuPy(16, 0, 4, 0.0, np.zeros((2, 2)), 16)
    with pytest.raises(ValueError):
        RecToolsIRCuPy(16, 0, 4, 0.0, geom["angles"], 16, OS_number=0)


def test_simple_iterative_methods_vs_oracle(oracle, geom):
    """Landweber / SIRT / CGLS loops (methodsIR_CuPy.py:128-309 of the reference) on the HIP operators vs the same
    loops written with the oracle's operators."""
    sino, angles, nz, n = geom["sino"], geom["angles"], geom["nz"], geom["n"]
    P = oracle.Projector(nz, n, n, angles)
    # Landweber
    x = np.zeros((nz, n, n), np.float32)
    for _ in range(5):
        x = x - np.float32(1e-3) * P.bp(P.fp(x) - sino)
    got = host(make(geom).Landweber(data_dict(geom), {"iterations": 5, "tau_step_lanweber": 1e-3, "recon_mask_radius": None}))
    assert rel(got, x) < TOL
    # SIRT
    with np.errstate(divide="ignore"):
        R = np.nan_to_num(np.float32(1) / P.fp(np.ones((nz, n, n), np.float32)), nan=1.0, posinf=1.0, neginf=1.0)
        Cm = np.nan_to_num(np.float32(1) / P.bp(np.ones_like(sino)), nan=1.0, posinf=1.0, neginf=1.0)
    x = np.ones((nz, n, n), np.float32)
    for _ in range(4):
        x = x + Cm * P.bp(R * (sino - P.fp(x)))
    got = host(make(geom).SIRT(data_dict(geom), {"iterations": 4, "recon_mask_radius": None}))
    assert rel(got, x) < TOL
    # CGLS
    x = np.zeros(nz * n * n, np.float32)
    d = P.bp(sino).ravel()
    normr2 = np.inner(d, d)
    r = sino.ravel().copy()
    for _ in range(4):
        Ad = P.fp(d.reshape(nz, n, n)).ravel()
        alpha = normr2 / np.inner(Ad, Ad)
        x = x + alpha * d
        r = r - alpha * Ad
        s = P.bp(r.reshape(sino.shape)).ravel()
        normr2_new = np.inner(s, s)
        d = s + (normr2_new / normr2) * d
        normr2 = normr2_new
    got = host(make(geom).CGLS(data_dict(geom), {"iterations": 4, "recon_mask_radius": None}))
    assert rel(got, x.reshape(nz, n, n)) < 1e-4  # inner products accumulate in a different order


def test_medium_size_fista_os_pdtv_30_inner_iterations(oracle, pd_arith):
    """PD_TV(30) inside a FISTA-OS run of several outer iterations: as shipped (relaxed float32 arithmetic) <= 1e-5 from
    the oracle (north-star tolerance), i.e. the per-call 1e-7 differences do not grow through the outer loop; with the
    reference's roundings (variant 22) bit for bit."""
    from tomobar_amd.methodsIR_CuPy import RecToolsIRCuPy
    nz, det, na, os_n = 12, 160, 72, 6
    angles = np.linspace(0, np.pi, na, endpoint=False)
    sino = oracle.shepp_logan_sino(det, nz, det, angles) / det
    sino += 0.01 * np.random.default_rng(0).standard_normal(sino.shape).astype(np.float32)
    P = oracle.Projector(nz, det, det, angles, 0.0, os_n)
    Lc = oracle.power_method(P, np.random.default_rng(1).standard_normal((nz, det, det)).astype(np.float32))
    reg = {"method": "PD_TV", "regul_param": 0.002, "iterations": 30, "methodTV": 0, "PD_LipschitzConstant": 12.0}
    want = oracle.fista(P, sino, 5, Lc, True, reg)
    rt = RecToolsIRCuPy(det, 0, nz, 0.0, angles, det, 0, os_n)
    rec = rt.FISTA({"projection_data": sino, "data_axes_labels_order": ["detY", "angles", "detX"]},
                   {"iterations": 5, "lipschitz_const": Lc, "nonnegativity": True, "recon_mask_radius": None},
                   {"method": "PD_TV", "regul_param": 0.002, "iterations": 30})
    r = pd_arith.check(host(rec), want, what="FISTA-OS(6) x 5 + PD_TV(30)")
    if not pd_arith.exact:
        print("FISTA-OS(6) x 5 + PD_TV(30), shipped arithmetic: rel-L2 vs oracle =", r)


def test_medium_size_fista_os_pdtv_pad_vs_oracle(oracle, pd_arith):
    """A geometry large enough to have several detector / voxel tiles, clipped windows and a padded detector:
    FISTA-OS + PD_TV on the MI355X must equal the oracle's run bit for bit (PD_TV variant 22; as shipped: 1e-5)."""
    from tomobar_amd.methodsIR_CuPy import RecToolsIRCuPy
    nz, det, pad, na, os_n = 6, 200, 20, 90, 5
    n = det + 2 * pad
    angles = np.linspace(0, np.pi, na, endpoint=False)
    sino = oracle.shepp_logan_sino(det, nz, det, angles) / det
    sino += 0.01 * np.random.default_rng(0).standard_normal(sino.shape).astype(np.float32)
    P = oracle.Projector(nz, n, n, angles, 0.0, os_n)
    Lc = oracle.power_method(P, np.random.default_rng(1).standard_normal((nz, n, n)).astype(np.float32))
    reg = {"method": "PD_TV", "regul_param": 0.002, "iterations": 7, "methodTV": 0, "PD_LipschitzConstant": 12.0}
    want = oracle.crop_recon(oracle.fista(P, oracle.pad_detector(sino, pad), 2, Lc, True, reg), det)
    rt = RecToolsIRCuPy(det, pad, nz, 0.0, angles, det, 0, os_n)
    rec = rt.FISTA({"projection_data": sino, "data_axes_labels_order": ["detY", "angles", "detX"]},
                   {"iterations": 2, "lipschitz_const": Lc, "nonnegativity": True},
                   {"method": "PD_TV", "regul_param": 0.002, "iterations": 7})
    pd_arith.check(host(rec), want, what="FISTA-OS(5) + PD_TV(7), padded detector")


def test_dir_forwproj_backproj(oracle, geom):
    from tomobar_amd.methodsDIR_CuPy import RecToolsDIRCuPy
    rt = RecToolsDIRCuPy(geom["n"], 0, geom["nz"], 0.0, geom["angles"], geom["n"], device_projector=0)
    P = oracle.Projector(geom["nz"], geom["n"], geom["n"], geom["angles"])
    vol = np.random.default_rng(0).random((geom["nz"], geom["n"], geom["n"])).astype(np.float32)
    assert rel(host(rt.FORWPROJ(torch.from_numpy(vol).cuda())), P.fp(vol)) < 1e-6
    # the OUTPUT axis order of FORWPROJ follows data_axes_labels_order like the reference's (methodsDIR_CuPy.py:84-88)
    for labels, perm in ((["angles", "detY", "detX"], (1, 0, 2)), (["detX", "angles", "detY"], (2, 1, 0)),
                         (["detY", "angles", "detX"], (0, 1, 2))):
        out = rt.FORWPROJ(torch.from_numpy(vol).cuda(), data_axes_labels_order=labels)
        assert out.is_contiguous()
        assert np.array_equal(host(out), np.transpose(P.fp(vol), perm)), labels
    swapped = torch.from_numpy(geom["sino"]).cuda().permute(1, 0, 2)  # a strided [angles, detY, detX] view
    got = rt.BACKPROJ(swapped, data_axes_labels_order=["angles", "detY", "detX"])
    assert rel(host(got), P.bp(geom["sino"])) < 1e-6


OSEM_CASES = {
    # name: (os_number, algorithm dict, regularisation)
    "mlem": (None, dict(iterations=3), None),
    "osem_os4": (4, dict(iterations=2), None),
    "osem_os7_mask": (7, dict(iterations=1, recon_mask_radius=0.9), None),
    "osem_os4_pdtv": (4, dict(iterations=2, nonnegativity=True), dict(method="PD_TV", regul_param=0.002, iterations=6)),
    "mlem_roftv": (None, dict(iterations=2), dict(method="ROF_TV", regul_param=0.002, iterations=5,
                                                  time_marching_step=0.002)),
}


@pytest.mark.parametrize("name", sorted(OSEM_CASES))
def test_osem_against_reference_python_loop(oracle, golden_dir, name, pd_arith):
    """RecToolsIRCuPy.OSEM vs the fixture made by the REFERENCE's own OSEM loop (make_osem_golden.py;
    methodsIR_CuPy.py:587-667), and bit for bit vs the oracle's restatement."""
    g = np.load(os.path.join(golden_dir, "osem_golden.npz"))
    os_n, alg, reg = OSEM_CASES[name]
    sino, angles = g["sino"], g["angles"]
    nz, _, n = sino.shape
    from tomobar_amd.methodsIR_CuPy import RecToolsIRCuPy
    rt = RecToolsIRCuPy(n, 0, nz, 0.0, angles, n, 0, os_n)
    d = {"projection_data": torch.from_numpy(sino).cuda(), "data_axes_labels_order": ["detY", "angles", "detX"]}
    rec = rt.OSEM(d, dict(alg), None if reg is None else dict(reg))
    got = host(rec)
    assert got.dtype == np.float32 and got.shape == (nz, n, n)
    assert rel(got, g[name]) < TOL, rel(got, g[name])
    P = oracle.Projector(nz, n, n, angles, 0.0, os_n or 1)
    full_reg = None if reg is None else {"regul_param": 0.001, "iterations": 150, "time_marching_step": 0.005,
                                         "PD_LipschitzConstant": 12.0, "methodTV": 0, **reg}
    want = oracle.circular_mask(oracle.osem(P, sino, alg["iterations"], alg.get("nonnegativity", False), full_reg),
                                alg.get("recon_mask_radius", 1.0))
    if reg is not None and reg["method"] == "PD_TV":
        pd_arith.check(got, want, what=name)
    else:
        assert np.array_equal(got, want), np.abs(got - want).max()


def test_exact_roundings_key_selects_the_reference_arithmetic_for_one_call(oracle, geom):
    """``_regularisation_["exact_roundings"] = True`` (an extension of the reference's dictionary): PD_TV with the reference's
    rounding sequence for this call only -- bit-identical to the oracle without touching tomo_set_variant; the library's
    switch is back at the default afterwards (the next call is the relaxed build again: within 1e-5, not equal)."""
    from tomobar_amd import ops
    os_n = 4
    rt = make(geom, os_number=os_n)
    P = oracle.Projector(geom["nz"], geom["n"], geom["n"], geom["angles"], 0.0, os_n)
    Lc = 2.0e4
    reg = {"method": "PD_TV", "regul_param": 0.002, "iterations": 9, "methodTV": 0, "PD_LipschitzConstant": 12.0}
    alg = {"iterations": 2, "lipschitz_const": Lc, "nonnegativity": True, "recon_mask_radius": None}
    want = oracle.fista(P, geom["sino"], 2, Lc, True, reg)
    assert ops.get_variant("pdtv") == 0
    exact = host(rt.FISTA(data_dict(geom), dict(alg), {"method": "PD_TV", "regul_param": 0.002, "iterations": 9,
                                                      "exact_roundings": True}))
    assert np.array_equal(exact, want), float(np.abs(exact - want).max())
    assert ops.get_variant("pdtv") == 0
    default = host(rt.FISTA(data_dict(geom), dict(alg), {"method": "PD_TV", "regul_param": 0.002, "iterations": 9}))
    assert rel(default, want) < TOL and not np.array_equal(default, want)


@pytest.mark.parametrize("method", ["PD_TV", None])
def test_fista_repeated_calls_on_one_object_are_bit_identical(oracle, geom, method, pd_arith):
    """the transposed X_t that the momentum kernel leaves in the projector context is a one-shot
    token; it must never survive a FISTA call.  Three calls on ONE object with the Lipschitz constant supplied (no power
    method in between, so the allocator hands the next call's X_t the address of the last one's): original data, other
    data, original data again -- calls 1 and 3 must agree bit for bit with each other and with the oracle."""
    os_n = 4
    rt = make(geom, os_number=os_n)
    P = oracle.Projector(geom["nz"], geom["n"], geom["n"], geom["angles"], 0.0, os_n)
    Lc = 2.0e4
    reg = None if method is None else {"method": "PD_TV", "regul_param": 0.002, "iterations": 6, "methodTV": 0,
                                       "PD_LipschitzConstant": 12.0}
    alg = {"iterations": 2, "lipschitz_const": Lc, "nonnegativity": True, "recon_mask_radius": None}
    want = oracle.fista(P, geom["sino"], 2, Lc, True, reg)
    outs = []
    for scale in (1.0, -3.0, 1.0):
        d = data_dict(geom)
        d["projection_data"] = d["projection_data"] * scale
        r = None if reg is None else {"method": "PD_TV", "regul_param": 0.002, "iterations": 6}
        outs.append(host(rt.FISTA(d, dict(alg), r)))
    assert np.array_equal(outs[0], outs[2])
    if method is None:
        assert np.array_equal(outs[0], want), float(np.abs(outs[0] - want).max())
    else:
        pd_arith.check(outs[0], want, what="FISTA-OS(4) + PD_TV(6), repeated calls")
    # and the explicit entry point: an armed token is dropped by tomo_ctx_invalidate / spent by an unrelated projection
    A = rt.Atools
    x = torch.rand(A.vol_shape(), device="cuda")
    xo = torch.rand(A.vol_shape(), device="cuda")
    xt = torch.empty_like(x)
    A.momentum(x, xo, xt, 0.5)
    ref = host(A.forward(xt, 1))                 # uses the token
    xt.mul_(2.0)                                 # same address, new contents
    assert np.array_equal(host(A.forward(xt, 1)), 2.0 * ref)   # token spent: fresh transpose (power of two is exact)
    A.momentum(x, xo, xt, 0.5)
    A.invalidate()
    xt.mul_(2.0)
    assert np.array_equal(host(A.forward(xt, 1)), 2.0 * ref)
    A.momentum(x, xo, xt, 0.5)
    A.forward(xo, 0)                             # another volume's projection spends the token too
    xt.mul_(2.0)
    assert np.array_equal(host(A.forward(xt, 1)), 2.0 * ref)


def test_reserve_scratch_is_an_optional_set_up_step(oracle):
    """RecToolsIRCuPy.reserve_scratch (no reference counterpart): allocates / places the TV arena ahead of the first call.  It
    changes no result, tolerates every regulariser dictionary the drivers accept (None, no method, 2D geometry, binary16
    duals, ROF_TV) and a second call is a no-op; the drivers make the same reservation themselves."""
    import ctypes as C
    from tomobar_amd import _lib
    from tomobar_amd.methodsIR_CuPy import RecToolsIRCuPy
    nz, n, na = 6, 40, 24
    angles = np.linspace(0, np.pi, na, endpoint=False)
    sino = torch.rand((nz, na, n), device="cuda")
    data = {"projection_data": sino, "data_axes_labels_order": ["detY", "angles", "detX"]}
    algo = {"iterations": 2, "lipschitz_const": 2000.0}
    reg = {"method": "PD_TV", "regul_param": 1e-3, "iterations": 6}
    a = RecToolsIRCuPy(n, 0, nz, 0.0, angles, n, 0, None)
    want = a.FISTA(dict(data), dict(algo), dict(reg))
    _lib.check(_lib.lib().tomo_release_scratch(0))
    b = RecToolsIRCuPy(n, 0, nz, 0.0, angles, n, 0, None)
    for r in (None, {}, {"method": None}, dict(reg), dict(reg), dict(reg, half_precision=True), {"method": "ROF_TV"}):
        b.reserve_scratch(r)
    assert torch.equal(b.FISTA(dict(data), dict(algo), dict(reg)), want)
    flat = RecToolsIRCuPy(n, 0, None, 0.0, angles, n, 0, None)       # 2D geometry: the arena of the 2D kernels
    flat.reserve_scratch(dict(reg))
    assert flat.FISTA({"projection_data": sino[0], "data_axes_labels_order": ["angles", "detX"]}, dict(algo), dict(reg)).shape == (1, n, n)
