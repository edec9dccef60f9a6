"""GPU parity tests: the HIP path (through the C-ABI of libtomo_mi355x.so) against the CPU oracle on the same
seeded inputs, and against the committed golden fixtures.  Tolerance: 1e-5 relative L2 (BASELINE.json north_star);
where the kernels reproduce the oracle's rounding sequence the tests also report / require bit-exactness."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

TOL = 1e-5
# kernel variants other than the defaults live in libtomo_mi355x_dev.so: a parameter / test marked `dev_variants` runs with the
# package pointed at that library (tests/conftest.py); everything else runs the shipped libtomo_mi355x.so
DEV = pytest.mark.dev_variants


def _v(*variants):
    return [v if v == 0 else pytest.param(v, marks=DEV) for v in variants]



def rel(a, b):
    a = np.asarray(a, dtype=np.float64).ravel()
    b = np.asarray(b, dtype=np.float64).ravel()
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


def host(t):
    torch.cuda.synchronize()
    return t.detach().cpu().numpy()


@pytest.fixture(scope="module")
def ops():
    from tomobar_amd import ops as _ops
    return _ops


# ------------------------------------------------------------------------------------------ projector pair
GEOMS = [
    # nz, n, nu, na, cor, os
    (5, 40, 40, 24, 0.0, 1),
    (3, 37, 45, 19, 1.25, 1),       # odd sizes, detector wider than the grid, CoR offset
    (18, 70, 64, 31, -2.5, 4),      # several z-batches, detector narrower than the grid, OS with a trimmed subset
    (1, 130, 130, 50, 0.0, 7),      # single slice (2D case), >2 x-tiles
    (33, 24, 24, 9, "vec", 2),      # per-angle CoR vector, nz not a multiple of 16
    (6, 800, 800, 25, 0.0, 1),      # wide detector: the 1024-thread whole-row FP kernel, 25 BP bricks per row
    (5, 780, 900, 18, 0.7, 3),      # same with detector != grid, CoR offset, subsets of 6 angles (tail batches)
]


def make_pair(oracle, g, flags=0):
    from tomobar_amd.projector import HipTools3D
    nz, n, nu, na, cor, os_n = g
    angles = np.linspace(0.1, np.pi + 0.1, na, endpoint=False)
    if isinstance(cor, str):
        cor = np.linspace(-1.5, 2.0, na)
    P = oracle.Projector(nz, n, nu, angles, cor, os_n, flags=flags)
    H = HipTools3D(nu, 0, nz, angles, cor, n, "gpu", 0, os_n if os_n > 1 else None, lerp8=bool(flags))
    return P, H


@pytest.mark.parametrize("g", GEOMS)
@pytest.mark.parametrize("variant", _v(0, 1, 2, 3))   # 0 re-lays the planar sinogram quad-interleaved (LDS-DMA staging); 3 = planar staging
def test_backprojection_vs_oracle(oracle, ops, g, variant):
    P, H = make_pair(oracle, g)
    ops.set_variant("bp", variant)
    rng = np.random.default_rng(1)
    subsets = [None] if P.os_number == 1 else list(range(P.os_number))
    for s in subsets:
        nsel = len(P.subsets[s]) if s is not None else P.na
        sino = rng.standard_normal((P.nz, nsel, P.nu)).astype(np.float32)
        want = P.bp(sino, s)
        got = host(H.backward(dev(sino), s))
        assert got.shape == want.shape
        assert rel(got, want) < 1e-6, (g, variant, s, rel(got, want))
        assert np.array_equal(got, want), f"BP not bit-identical: max abs {np.abs(got - want).max()}"


@pytest.mark.parametrize("g", GEOMS + [(4, 300, 520, 40, 3.0, 1), (5, 64, 700, 13, 0.0, 1)])
@pytest.mark.parametrize("variant", _v(0, 1, 2))
def test_forward_projection_vs_oracle(oracle, ops, g, variant):
    P, H = make_pair(oracle, g)
    ops.set_variant("fp", variant)
    rng = np.random.default_rng(2)
    vol = rng.standard_normal((P.nz, P.n, P.n)).astype(np.float32)
    subsets = [None] if P.os_number == 1 else list(range(P.os_number))
    for s in subsets:
        want = P.fp(vol, s)
        got = host(H.forward(dev(vol), s))
        assert got.shape == want.shape
        assert rel(got, want) < 1e-6, (g, s, rel(got, want))
        assert np.array_equal(got, want), f"FP not bit-identical: max abs {np.abs(got - want).max()}"


@pytest.mark.parametrize("n,na,os_n", [(280, 180, 5), (256, 180, None), (300, 60, None), (280, 90, 3)])
@pytest.mark.parametrize("variant", _v(0, 2))
def test_forward_projection_oblique_windows(oracle, ops, n, na, os_n, variant):
    """Regression: detector tiles whose staged row window is clipped by the volume at both ends of the march but not
    in the middle (oblique rays at the detector edge) -- the LDS pitch must come from the unclipped window."""
    from tomobar_amd.projector import HipTools3D
    ops.set_variant("fp", variant)
    angles = np.linspace(0, np.pi, na, endpoint=False)
    H = HipTools3D(n, 0, 4, angles, 0.0, n, "gpu", 0, os_n)
    P = oracle.Projector(4, n, n, angles, 0.0, os_n or 1)
    vol = np.random.default_rng(4).standard_normal((4, n, n)).astype(np.float32)
    for s in (range(os_n) if os_n else [None]):
        got = host(H.forward(dev(vol), s))
        assert np.array_equal(got, P.fp(vol, s)), (n, na, s)


@pytest.mark.dev_variants
@pytest.mark.parametrize("g", [(6, 300, 300, 400, 0.0, 1), (5, 520, 700, 300, 1.5, 1), (9, 200, 333, 512, "vec", 2),
                               (4, 1100, 1100, 1000, -2.0, 1)])
def test_forward_projection_dense_angle_form(oracle, ops, g):
    """The dense-angle form of the forward projector (256 pixels x 16 angles per workgroup, round 4), forced wherever it is
    applicable (fp variant 3, dev flavour): bit for bit against the oracle, plain and with the residual epilogue."""
    P, H = make_pair(oracle, g)
    ops.set_variant("fp", 3)
    rng = np.random.default_rng(12)
    vol = rng.standard_normal((P.nz, P.n, P.n)).astype(np.float32)
    b = rng.random((P.nz, P.na, P.nu)).astype(np.float32)
    for s in ([None] if P.os_number == 1 else list(range(P.os_number))):
        want = P.fp(vol, s)
        got = host(H.forward(dev(vol), s))
        assert "dense(256 pixels x 16 angles" in H.kernel_path("fp"), H.kernel_path("fp")
        assert np.array_equal(got, want), (g, s, np.abs(got - want).max())
        res = torch.empty(H.sino_shape(s), dtype=torch.float32, device="cuda")
        H.residual(dev(vol), dev(b), None, "LS", s, res)
        idx = P.subsets[s] if s is not None else slice(None)
        assert np.array_equal(host(res), (want - b[:, idx]).astype(np.float32)), (g, s)


@pytest.mark.dev_variants
@pytest.mark.parametrize("seed", range(12))
def test_forward_projection_dense_angle_form_random_geometries(oracle, ops, seed):
    """Seeded random geometries with MANY angles (130-700: classes of >= 32 angles, windows of a 16-angle group <= 512 columns
    or not), detector wider / narrower than the grid, rotation-axis offsets, arbitrary angle ranges, subsets: the dense-angle
    form wherever it applies (fp variant 3; the other forms where it does not), bit for bit against the oracle."""
    from tomobar_amd.projector import HipTools3D
    rng = np.random.default_rng(3000 + seed)
    nz = int(rng.integers(1, 10))
    n = int(rng.integers(64, 400))
    nu = int(max(32, n + rng.integers(-n // 3, n // 2 + 1)))
    na = int(rng.integers(130, 700))
    start = float(rng.uniform(-np.pi, np.pi))
    span = float(rng.choice([np.pi, 2 * np.pi, 1.3])) * float(rng.choice([1.0, -1.0]))
    angles = start + np.linspace(0, span, na, endpoint=False)
    cor = float(rng.uniform(-0.1, 0.1) * nu) if seed % 3 else np.asarray(rng.uniform(-2, 2, na))
    os_n = int(rng.choice([1, 1, 2, 3]))
    P = oracle.Projector(nz, n, nu, angles, cor, os_n)
    H = HipTools3D(nu, 0, nz, angles, cor, n, "gpu", 0, os_n if os_n > 1 else None)
    ops.set_variant("fp", 3)
    vol = rng.standard_normal((nz, n, n)).astype(np.float32)
    took = []
    for s in ([None] if os_n == 1 else list(range(os_n))):
        got = host(H.forward(dev(vol), s))
        took.append("dense(" in H.kernel_path("fp"))
        assert np.array_equal(got, P.fp(vol, s)), (seed, s, (nz, n, nu, na, os_n), H.kernel_path("fp"))
    print("dense form taken:", took, (nz, n, nu, na, os_n))


@pytest.mark.parametrize("bp_variants", [(0,), pytest.param((1, 2, 3), marks=DEV)])
def test_lerp8_mode_and_reference_literals(oracle, ops, bp_variants):
    """tests/test_RecToolsDIRCuPy.py:671-694 of the reference: ones(128,160,160) -> min 67.27458 max 225.27428."""
    from tomobar_amd.projector import HipTools3D
    angles = np.deg2rad(np.arange(180.0))
    H = HipTools3D(160, 0, 8, angles, 0.0, 160, "gpu", 0, None, lerp8=True)
    s = host(H.forward(torch.ones((8, 160, 160), dtype=torch.float32, device="cuda")))
    np.testing.assert_allclose(s.min(), 67.27458, rtol=2e-6)
    np.testing.assert_allclose(s.max(), 225.27428, rtol=2e-6)
    P = oracle.Projector(8, 160, 160, angles, flags=oracle.FLAG_LERP8)
    sino = np.random.default_rng(0).random((8, 180, 160)).astype(np.float32)
    for v in bp_variants:
        ops.set_variant("bp", v)
        assert rel(host(H.backward(dev(sino))), P.bp(sino)) < 1e-6


def test_power_method_literals(oracle):
    """tests/test_RecToolsIRCuPy.py:316,573,639 of the reference."""
    from tomobar_amd.methodsIR_CuPy import RecToolsIRCuPy
    angles = np.deg2rad(np.arange(180.0))
    dummy = {"projection_data": None}
    for pad, os_n, literal in ((0, None, 27550.463), (0, 5, 5510.867), (60, 5, 9644.283)):
        rt = RecToolsIRCuPy(160, pad, 4, 0.0, angles, 160, 0, os_n)
        rt.power_seed = 0
        np.testing.assert_allclose(rt.powermethod(dict(dummy)), literal, rtol=1e-5)


@pytest.mark.parametrize("g", [(6, 36, 40, 22, 0.5, 3),
                               # 3 x 6 x 2 whole 32x16x16 bricks (epilogue through LDS in dwordx4 row segments) next to
                               # ragged ones in x, y and z (direct stores) in the same launch
                               (37, 104, 96, 21, -0.75, 3)])
@pytest.mark.parametrize("bp_variants", [(0,), pytest.param((1, 2, 3), marks=DEV)])
def test_fused_residual_and_gradient_steps(oracle, ops, g, bp_variants):
    P, H = make_pair(oracle, g)
    rng = np.random.default_rng(3)
    b = rng.random((P.nz, P.na, P.nu)).astype(np.float32)
    w = oracle.pwls_weights(b)
    x = rng.random((P.nz, P.n, P.n)).astype(np.float32) * 0.05
    xo = rng.random((P.nz, P.n, P.n)).astype(np.float32) * 0.05
    u = rng.standard_normal((P.nz, P.n, P.n)).astype(np.float32) * 0.01
    wd = ops.pwls_weights(dev(b))
    assert np.array_equal(host(wd), w)
    for s in range(3):
        idx = P.subsets[s]
        ax = P.fp(x, s)
        res = torch.empty(H.sino_shape(s), dtype=torch.float32, device="cuda")
        for fid, want in (("LS", ax - b[:, idx]), ("PWLS", (ax - b[:, idx]) * w[:, idx]),
                          ("KL", np.float32(1) - b[:, idx] / np.clip(ax, np.float32(1e-8), None))):
            H.residual(dev(x), dev(b), wd if fid == "PWLS" else None, fid, s, res)
            assert np.array_equal(host(res), want.astype(np.float32)), fid
        # gradient step epilogues
        r = (ax - b[:, idx]).astype(np.float32)
        grad = P.bp(r, s)
        linv, beta = np.float32(1 / 300.0), np.float32(0.37)
        for variant in bp_variants:
            ops.set_variant("bp", variant)
            for nonneg in (False, True):
                X = x - linv * grad
                if nonneg:
                    X = np.maximum(X, 0)
                out = torch.empty_like(dev(x))
                H.grad_step(dev(r), dev(x), out, linv, nonneg, s)
                assert np.array_equal(host(out), X)
                xt_d, xo_d = dev(x), dev(xo)
                H.grad_step_momentum(dev(r), xt_d, xo_d, linv, beta, nonneg, s)
                assert np.array_equal(host(xo_d), X)
                assert np.array_equal(host(xt_d), X + beta * (X - xo))
            # ADMM z-update (z = x here), with and without relaxation
            tau, rho, al = np.float32(0.002), np.float32(1.7), 1.6
            for relax_on in (False, True):
                z = x.copy()
                ga = rho * (z - xo + u)
                zn = z - tau * (grad + ga)
                zn = np.maximum(zn, 0)
                if relax_on:
                    zn = np.float32(1.0 - al) * z + np.float32(al) * zn
                z_d, zu_d = dev(x), torch.empty_like(dev(x))
                H.admm_z_update(dev(r), z_d, dev(xo), dev(u), zu_d, tau, rho, relax_on, np.float32(1.0 - al),
                                np.float32(al), True, s)
                assert np.array_equal(host(z_d), zn)
                assert np.array_equal(host(zu_d), zn + u)


@pytest.mark.parametrize("g", [(6, 36, 40, 22, 0.5, 3),          # nz not a multiple of 4: zero-filled tail of the last quad
                               (37, 104, 96, 21, -0.75, 3),      # whole and ragged bricks; 37 slices = 2 bricks + 5 slices
                               (1, 130, 130, 50, 0.0, 7),        # a single slice
                               (16, 300, 520, 40, 3.0, 1),       # wide detector, whole-row FP form, no subsets
                               (9, 200, 333, 512, "vec", 2)])    # dense angle set (256 x 16 FP form when forced), per-angle CoR
@pytest.mark.parametrize("fp_variant", _v(0, 3, 2, 1))
def test_quad_interleaved_residual_layout(oracle, ops, g, fp_variant):
    """The private residual layout between the fused forward and back projection (TOMO_RESIDUAL_ZQUAD, round 5): what the
    forward projector leaves is the planar residual with the four slices of a quad interleaved (zeros past nz), and the
    three fused back-projection epilogues fed with it return bit for bit what they return for the planar layout -- which
    test_fused_residual_and_gradient_steps pins to the oracle."""
    P, H = make_pair(oracle, g)
    ops.set_variant("fp", fp_variant)
    rng = np.random.default_rng(31)
    b = rng.random((P.nz, P.na, P.nu)).astype(np.float32)
    wd = ops.pwls_weights(dev(b))
    x, xo = (rng.random((P.nz, P.n, P.n)).astype(np.float32) * 0.05 for _ in range(2))
    u = rng.standard_normal((P.nz, P.n, P.n)).astype(np.float32) * 0.01
    linv, beta = np.float32(1 / 300.0), np.float32(0.37)
    tau, rho, al = np.float32(0.002), np.float32(1.7), 1.6
    for s in ([None] if P.os_number == 1 else list(range(P.os_number))):
        idx = P.subsets[s] if s is not None else slice(None)
        for fid in ("LS", "PWLS", "KL"):
            assert H.residual_layout() == "planar"
            planar = H.residual_buffer(s)
            assert tuple(planar.shape) == H.sino_shape(s)
            H.residual(dev(x), dev(b), wd if fid == "PWLS" else None, fid, s, planar)
            if fid == "LS":
                assert np.array_equal(host(planar), (P.fp(x, s) - b[:, idx]).astype(np.float32))
            H.set_residual_layout("zquad")
            try:
                quad = H.residual_buffer(s)
                nq = -(-P.nz // 4)
                assert tuple(quad.shape) == (nq, H.subset_size(s), P.nu, 4)
                quad.fill_(float("nan"))
                H.residual(dev(x), dev(b), wd if fid == "PWLS" else None, fid, s, quad)
                q = host(quad)
                assert np.array_equal(host(H.residual_as_planar(quad, s)), host(planar)), (g, s, fid)
                tail = q.transpose(0, 3, 1, 2).reshape(4 * nq, -1)[P.nz:]
                assert not np.any(tail), "slices past the end of the volume must be written as zeros"
                if fid != "LS":
                    continue
                for nonneg in (False, True):
                    want, got = torch.empty_like(dev(x)), torch.empty_like(dev(x))
                    H.set_residual_layout("planar"); H.grad_step(planar, dev(x), want, linv, nonneg, s)
                    H.set_residual_layout("zquad"); H.grad_step(quad, dev(x), got, linv, nonneg, s)
                    assert "quad-interleaved" in H.kernel_path("bp") or H.kernel_path("bp").startswith("direct")
                    assert np.array_equal(host(got), host(want)), (g, s, nonneg)
                    xt_w, xo_w, xt_g, xo_g = dev(x), dev(xo), dev(x), dev(xo)
                    H.set_residual_layout("planar"); H.grad_step_momentum(planar, xt_w, xo_w, linv, beta, nonneg, s)
                    H.set_residual_layout("zquad"); H.grad_step_momentum(quad, xt_g, xo_g, linv, beta, nonneg, s)
                    assert np.array_equal(host(xt_g), host(xt_w)) and np.array_equal(host(xo_g), host(xo_w))
                for relax_on in (False, True):
                    z_w, zu_w, z_g, zu_g = dev(x), torch.empty_like(dev(x)), dev(x), torch.empty_like(dev(x))
                    H.set_residual_layout("planar")
                    H.admm_z_update(planar, z_w, dev(xo), dev(u), zu_w, tau, rho, relax_on, np.float32(1.0 - al), np.float32(al), True, s)
                    H.set_residual_layout("zquad")
                    H.admm_z_update(quad, z_g, dev(xo), dev(u), zu_g, tau, rho, relax_on, np.float32(1.0 - al), np.float32(al), True, s)
                    assert np.array_equal(host(z_g), host(z_w)) and np.array_equal(host(zu_g), host(zu_w))
                # the plain operators never follow the context's setting
                assert np.array_equal(host(H.backward(planar, s)), P.bp(host(planar), s))
            finally:
                H.set_residual_layout("planar")


def test_quad_interleaved_residual_layout_rules(oracle, ops):
    """Who may run while the private layout is set: the ring-term residual (read by tomo_ring_gh_reduce as [detY, angles,
    detX]) refuses, an unknown layout and a vertical CoR component are rejected, and the drivers leave the context planar."""
    from tomobar_amd.methodsIR_CuPy import RecToolsIRCuPy
    from tomobar_amd.projector import HipTools3D
    angles = np.linspace(0, np.pi, 12, endpoint=False)
    H = HipTools3D(24, 0, 5, angles, 0.0, 24, "gpu", 0, None)
    H.set_residual_layout("zquad")
    vol, b = torch.zeros((5, 24, 24), device="cuda"), torch.zeros((5, 12, 24), device="cuda")
    with pytest.raises(ValueError, match="planar"):
        H.residual_ring(vol, b, torch.zeros((5, 24), device="cuda"), 1.0, None, H.residual_buffer(None))
    with pytest.raises(KeyError):
        H.set_residual_layout("columns")
    H.set_residual_layout("planar")
    cor = np.stack([np.zeros(12), np.full(12, 0.25)], axis=1)
    Hv = HipTools3D(24, 0, 5, angles, cor, 24, "gpu", 0, None)
    with pytest.raises(ValueError, match="vertical"):
        Hv.set_residual_layout("zquad")
    rt = RecToolsIRCuPy(24, 0, 5, 0.0, angles, 24, 0, None)
    sino = torch.rand((5, 12, 24), device="cuda")
    for call in (rt.FISTA, rt.ADMM):
        call({"projection_data": sino, "data_axes_labels_order": ["detY", "angles", "detX"]}, {"iterations": 2, "lipschitz_const": 500.0})
        assert rt.Atools.residual_layout() == "planar"
    with pytest.raises(ValueError):   # an exception inside the loop must not leave the context in the private layout
        rt.FISTA({"projection_data": sino, "data_axes_labels_order": ["detY", "angles", "detX"]},
                 {"iterations": 1, "lipschitz_const": 500.0}, {"method": "no such prox", "regul_param": 1e-3, "iterations": 2})
    assert rt.Atools.residual_layout() == "planar"


# ------------------------------------------------------------------------------------------ TV operators
TV_SHAPES = [(6, 9, 13), (1, 20, 17), (12, 1, 70), (10, 11, 1), (8, 8, 8), (3, 5, 131), (24, 19), (20, 70, 150)]
# bit-identical to the oracle: 22 = shipped, opt-in (FMA-corrected roundings for float32 duals as well); dev flavour: 2 / 21 = the
# compiler's IEEE sequences on the two- / three-iteration kernel, 1 = per-voxel kernel.  0 = shipped default (float32 duals:
# relaxed arithmetic, tolerance; binary16 duals: exact); 3 (dev) = relaxed arithmetic for binary16 duals as well
PD_EXACT_VARIANTS = [22] + _v(2, 1, 21)


@pytest.mark.parametrize("shape", TV_SHAPES)
@pytest.mark.parametrize("variant", PD_EXACT_VARIANTS)
def test_pdtv_vs_oracle(oracle, ops, shape, variant):
    from tomobar_amd.regularisersCuPy import PD_TV_cupy
    ops.set_variant("pdtv", variant)
    rng = np.random.default_rng(5)
    x = (rng.random(shape) * 0.3 + (np.indices(shape)[-1] > shape[-1] // 2)).astype(np.float32)
    for half in (False, True):
        for mtv in (0, 1):
            for nn in (0, 1):
                xi = (x - 0.6).astype(np.float32) if nn else x
                want = oracle.pd_tv(xi, 0.04, 11, mtv, nn, 8.0, half)
                got = host(PD_TV_cupy(dev(xi), 0.04, 11, mtv, nn, 8.0, 0, half))
                assert got.shape == want.shape and got.dtype == np.float32
                assert rel(got, want) < 1e-6, (shape, variant, half, mtv, nn, rel(got, want))
                assert np.array_equal(got, want), (shape, variant, half, mtv, nn, np.abs(got - want).max())


@pytest.mark.parametrize("shape", TV_SHAPES)
def test_pdtv_default_arithmetic_vs_oracle(oracle, ops, shape):
    """The shipped default on the shapes of test_pdtv_vs_oracle: float32 duals (relaxed arithmetic) within the north-star
    tolerance, with the bit-level statistics the review asked for -- drift of the relaxed build shows up here as a larger
    fraction of differing values / a larger ulp distance long before it reaches 1e-5 --, binary16 duals bit for bit."""
    from tomobar_amd.regularisersCuPy import PD_TV_cupy
    from conftest import ulp_distance
    rng = np.random.default_rng(5)
    x = (rng.random(shape) * 0.3 + (np.indices(shape)[-1] > shape[-1] // 2)).astype(np.float32)
    for half in (False, True):
        for mtv in (0, 1):
            # (nonneg, negative values in the data): without the clip negative iterates must pass through untouched -- the
            # relaxed float32 kernel is ONE instantiation whose clip threshold is an argument (0 or -inf)
            for nn, shifted in ((0, False), (0, True), (1, True)):
                xi = (x - 0.6).astype(np.float32) if shifted else x
                want = oracle.pd_tv(xi, 0.04, 11, mtv, nn, 8.0, half)
                got = host(PD_TV_cupy(dev(xi), 0.04, 11, mtv, nn, 8.0, 0, half))
                if shifted and not nn:
                    assert (got < 0).any() and (want < 0).any()
                if half:
                    assert np.array_equal(got, want), (shape, mtv, nn)
                    continue
                d = ulp_distance(got, want)
                assert rel(got, want) < 2e-6, (shape, mtv, nn, rel(got, want))      # 11 iterations: far inside 1e-5
                # values of magnitude ~1: one ulp is 6e-8 relative; the relaxed 1/sqrt and reciprocal are 1-2 ulp each
                assert np.percentile(d, 99.9) <= 64, (shape, mtv, nn, int(d.max()), float((d > 0).mean()))


@pytest.mark.parametrize("shape", [(9, 40, 70), (20, 70, 150), (5, 33, 131)])
@pytest.mark.parametrize("variant", _v(0, 3))
def test_pdtv_relaxed_arithmetic_vs_oracle(oracle, ops, shape, variant):
    """Relaxed arithmetic (v_rsq_f32, hoisted 1/(1+lt)) stays within the north-star tolerance of the oracle after 60
    iterations: the shipped default (0: float32 duals relaxed, binary16 duals exact) and the dev build that relaxes
    binary16 duals as well (3: one flipped binary16 rounding is 5e-4 of a dual value, so 2e-4 there)."""
    from tomobar_amd.regularisersCuPy import PD_TV_cupy
    ops.set_variant("pdtv", variant)
    rng = np.random.default_rng(6)
    x = (rng.random(shape) * 0.3 + (np.indices(shape)[-1] > shape[-1] // 2)).astype(np.float32)
    for half in (False, True):
        for mtv in (0, 1):
            want = oracle.pd_tv(x, 0.04, 60, mtv, 1, 12.0, half)
            got = host(PD_TV_cupy(dev(x), 0.04, 60, mtv, 1, 12.0, 0, half))
            if half and variant == 0:   # shipped build: binary16 duals run the exact arithmetic
                assert np.array_equal(got, want)
            assert rel(got, want) < (1e-5 if not half else 2e-4), (shape, variant, half, mtv, rel(got, want))


@pytest.mark.parametrize("shape", TV_SHAPES + [(9, 40, 70)])
def test_roftv_shipped_arithmetic_is_bit_identical_on_noise(oracle, ops, shape):
    """The shipped ROF_TV (round 3: the reference's sqrt / divide roundings reproduced with FMA correction steps instead
    of the compiler's IEEE expansions) against the oracle after 60 iterations on a noise-dominated input -- the input on
    which the former relaxed build (float32 sum + v_rsq_f32, now variant 3) drifted to 2.5e-5: D = a / sqrt(a^2 + m + 1e-8)
    has a gain of ~1e4 where all differences are ~1e-4, so only identical roundings hold the 1e-5 bar there."""
    from tomobar_amd.regularisersCuPy import ROF_TV_cupy
    ops.set_variant("roftv", 0)
    for seed in (6, 7):
        rng = np.random.default_rng(seed)
        x = (rng.random(shape) * 0.3 + (np.indices(shape)[-1] > shape[-1] // 2)).astype(np.float32)
        for half in (False, True):
            want = oracle.rof_tv(x, 0.05, 60, 0.005, half)
            got = host(ROF_TV_cupy(dev(x), 0.05, 60, 0.005, 0, half))
            assert np.array_equal(got, want), (shape, seed, half, rel(got, want))


@pytest.mark.parametrize("shape", TV_SHAPES)
@pytest.mark.parametrize("variant", _v(0, 2, 1))
def test_roftv_vs_oracle(oracle, ops, shape, variant):
    from tomobar_amd.regularisersCuPy import ROF_TV_cupy
    ops.set_variant("roftv", variant)
    rng = np.random.default_rng(6)
    x = (rng.random(shape) * 0.3 + (np.indices(shape)[-1] > shape[-1] // 2)).astype(np.float32)
    for half in (False, True):
        want = oracle.rof_tv(x, 0.05, 11, 0.005, half)
        got = host(ROF_TV_cupy(dev(x), 0.05, 11, 0.005, 0, half))
        assert got.shape == want.shape
        assert rel(got, want) < 1e-6, (shape, half, rel(got, want))
        assert np.array_equal(got, want), (shape, half, np.abs(got - want).max())


def test_tv_against_reference_fixtures(golden_dir, ops, pd_arith):
    """tests/golden/tv_golden.npz: outputs of the reference's own kernel sources (see make_tv_golden.py); both PD_TV
    arithmetics of the shipped library."""
    from tomobar_amd.regularisersCuPy import PD_TV_cupy, ROF_TV_cupy
    tv = np.load(os.path.join(golden_dir, "tv_golden.npz"))
    n = 0
    for key in tv.files:
        if not key.endswith("_meta"):
            continue
        kind, cid = key.split("_")[0], key.split("_")[1]
        m = tv[key]
        x = tv[f"in_{int(m[0])}"]
        if kind == "pd":
            _, half, mtv, nn, iters, lam, lip = m
            xi = (x - 0.6).astype(np.float32) if nn else x
            got = host(PD_TV_cupy(dev(xi), float(lam), int(iters), int(mtv), int(nn), float(lip), 0, bool(half)))
        else:
            _, half, iters, lam, tms = m
            got = host(ROF_TV_cupy(dev(x), float(lam), int(iters), float(tms), 0, bool(half)))
        for build in ("off", "fma"):
            assert rel(got, tv[f"{kind}_{cid}_{build}"]) < TOL, (key, build)
        n += 1
    assert n > 60


def test_tv_errors_and_2d_squeeze():
    """regularisersCuPy.py:67,73,208,315 and tests/test_regularisers.py:7-36 of the reference."""
    from tomobar_amd.regularisersCuPy import PD_TV_cupy, ROF_TV_cupy, _check_if_input_2d_or_3d
    for shape, want in (((100, 100), ((100, 100), True, 0)), ((10, 100, 100), ((10, 100, 100), False, 0)),
                        ((1, 100, 100), ((100, 100), True, 0)), ((16, 1, 100), ((16, 100), True, 1))):
        d, flag, ax = _check_if_input_2d_or_3d(torch.zeros(shape, dtype=torch.float32, device="cuda"))
        assert (tuple(d.shape), flag, ax) == want
    with pytest.raises(ValueError):
        PD_TV_cupy(torch.zeros((4, 4), dtype=torch.float64, device="cuda"))
    with pytest.raises(ValueError):
        ROF_TV_cupy(torch.zeros((4, 4), dtype=torch.float32, device="cuda"), gpu_id=-1)
    with pytest.raises(ValueError):
        PD_TV_cupy(torch.zeros((2, 2, 2, 2), dtype=torch.float32, device="cuda"))
    out = PD_TV_cupy(torch.rand((16, 1, 40), dtype=torch.float32, device="cuda"), 0.05, 5)
    assert tuple(out.shape) == (16, 1, 40) and out.dtype == torch.float32


# ------------------------------------------------------------------------------------------ glue kernels
def test_glue_kernels(ops):
    rng = np.random.default_rng(7)
    for n in (1, 5, 1024, 4099):
        x = rng.standard_normal(n).astype(np.float32)
        y = rng.standard_normal(n).astype(np.float32)
        z = rng.standard_normal(n).astype(np.float32)
        beta = np.float32(0.731)
        out = torch.empty(n, dtype=torch.float32, device="cuda")
        ops.momentum(dev(x), dev(y), out, beta)
        assert np.array_equal(host(out), x + beta * (x - y))
        u = dev(x)
        ops.admm_dual(u, dev(y), dev(z))
        assert np.array_equal(host(u), x + (y - z))
        yy = dev(y)
        ops.axpby(np.float32(0.3), dev(x), np.float32(-1.1), yy)
        assert np.array_equal(host(yy), np.float32(0.3) * x + np.float32(-1.1) * y)
        np.testing.assert_allclose(ops.norm2(dev(x)), np.linalg.norm(x.astype(np.float64)), rtol=1e-12)
        np.testing.assert_allclose(ops.dot(dev(x), dev(y)), np.dot(x.astype(np.float64), y.astype(np.float64)),
                                   rtol=1e-9, atol=1e-12)
        # unaligned views take the scalar path
        if n > 8:
            big = dev(np.concatenate([x, x]))
            v = big[1:n]
            ops.clamp_min(v, 0.0)
            assert np.array_equal(host(v), np.maximum(x[1:], 0))
    r = dev(np.array([0.0, 2.0, -4.0, np.inf, np.nan], np.float32))
    o = torch.empty_like(r)
    ops.recip_safe(r, o)
    assert host(o).tolist() == [1.0, 0.5, -0.25, 0.0, 1.0]


def test_pad_crop_mask_permute(oracle, ops):
    rng = np.random.default_rng(8)
    b = rng.random((3, 7, 10)).astype(np.float32)
    assert np.array_equal(host(ops.pad_edge(dev(b), 4)), oracle.pad_detector(b, 4))
    v = rng.standard_normal((3, 21, 21)).astype(np.float32)
    assert np.array_equal(host(ops.crop_center(dev(v), 12)), oracle.crop_recon(v, 12))
    for n in (20, 21):
        v = rng.standard_normal((2, n, n)).astype(np.float32)
        for radius in (0.85, 1.0, 0.5, 2.0):
            assert np.array_equal(host(ops.circ_mask_(dev(v), radius)), oracle.circular_mask(v, radius).astype(np.float32))
    t = dev(rng.random((4, 5, 6)).astype(np.float32))
    for perm in ((1, 0, 2), (2, 1, 0), (0, 2, 1), (2, 0, 1)):
        assert np.array_equal(host(ops.contiguous(t.permute(*perm))), np.ascontiguousarray(host(t).transpose(perm)))


@pytest.mark.parametrize("variants", [(0,), pytest.param((2, 1, 3), marks=DEV)])
@pytest.mark.parametrize("seed", range(24))
def test_projector_pair_random_geometries(oracle, ops, seed, variants):
    """seeded random geometries (sizes that are not multiples of any tile, detector wider / narrower than the grid,
    large rotation-axis offsets, arbitrary angle ranges incl. > 360 degrees and descending order, subsets with ragged
    tails): FP and BP of every kernel variant against the oracle, bit for bit"""
    from tomobar_amd.projector import HipTools3D
    rng = np.random.default_rng(1000 + seed)
    nz = int(rng.integers(1, 40))
    n = int(rng.integers(3, 150))
    nu = int(max(2, n + rng.integers(-n // 2, n // 2 + 1)))
    na = int(rng.integers(1, 70))
    start = float(rng.uniform(-np.pi, np.pi))
    span = float(rng.choice([np.pi, 2 * np.pi, 0.3, 7.5])) * float(rng.choice([1.0, -1.0]))
    angles = start + np.linspace(0, span, na, endpoint=False)
    cor = float(rng.uniform(-0.2, 0.2) * nu) if seed % 3 else np.asarray(rng.uniform(-3, 3, na))
    os_n = int(rng.integers(1, min(na, 9) + 1))
    P = oracle.Projector(nz, n, nu, angles, cor, os_n)
    H = HipTools3D(nu, 0, nz, angles, cor, n, "gpu", 0, os_n if os_n > 1 else None)
    vol = rng.standard_normal((nz, n, n)).astype(np.float32)
    subsets = [None] if os_n == 1 else list(range(os_n))
    for s in subsets:
        nsel = len(P.subsets[s]) if s is not None else na
        if nsel == 0:  # the reference's one-element trim can empty a subset (OS_number == angles): nothing to project
            continue
        sino = rng.standard_normal((nz, nsel, nu)).astype(np.float32)
        want_fp, want_bp = P.fp(vol, s), P.bp(sino, s)
        for v in variants:
            ops.set_variant("fp", v)
            got = host(H.forward(dev(vol), s))
            assert np.array_equal(got, want_fp), ("fp", v, seed, s, (nz, n, nu, na, os_n))
        ops.set_variant("fp", 0)
        for v in variants:
            ops.set_variant("bp", v)
            got = host(H.backward(dev(sino), s))
            assert np.array_equal(got, want_bp), ("bp", v, seed, s, (nz, n, nu, na, os_n))
        ops.set_variant("bp", 0)


@pytest.mark.parametrize("pd_variants", [(22, 0), pytest.param((21, 3), marks=DEV)])
@pytest.mark.parametrize("seed", range(3))
def test_tv_large_odd_shapes(oracle, ops, seed, pd_variants):
    """Large odd-shaped volumes (several z-chunks, hundreds of interior waves running the short form of the z-march
    kernels next to edge waves running the general form, ragged edges in every direction): the shipped three-iteration
    PD_TV with the reference's roundings (variant 22; dev flavour: 21 = compiler IEEE sequences on the same tiling) and
    the shipped ROF_TV against the oracle, bit for bit; the shipped default (0: float32 duals relaxed -> 1e-5, binary16
    duals exact -> bit for bit; dev 3: relaxed for both)."""
    from tomobar_amd.regularisersCuPy import PD_TV_cupy, ROF_TV_cupy
    rng = np.random.default_rng(9100 + seed)
    shape = (int(rng.integers(75, 230)), int(rng.integers(150, 420)), int(rng.integers(250, 700)))
    x = (rng.random(shape) * 0.4 + (np.indices(shape)[-1] > shape[-1] // 3) - 0.3).astype(np.float32)
    iters = int(rng.choice([3, 6, 7, 9]))
    half, mtv, nn = bool(seed & 1), int(rng.integers(0, 2)), int(rng.integers(0, 2))
    lam = float(rng.choice([0.01, 0.05]))
    want = oracle.pd_tv(x, lam, iters, mtv, nn, 8.0, half)
    xd = dev(x)
    for v in pd_variants:
        ops.set_variant("pdtv", v)
        got = host(PD_TV_cupy(xd, lam, iters, mtv, nn, 8.0, 0, half))
        if v == 3 or (v == 0 and not half):
            assert rel(got, want) < (2e-4 if half else 1e-5), ("pd relaxed", shape, iters, half, mtv, nn, rel(got, want))
        else:
            assert np.array_equal(got, want), ("pd", v, shape, iters, half, mtv, nn, np.abs(got - want).max())
    ops.set_variant("pdtv", 0)
    want_rof = oracle.rof_tv(x, lam, iters, 0.004, half)
    got = host(ROF_TV_cupy(xd, lam, iters, 0.004, 0, half))
    assert np.array_equal(got, want_rof), ("rof", shape, iters, half, np.abs(got - want_rof).max())


def test_scratch_arena_placement_search(oracle, ops):
    """The library picks its large scratch arenas among several candidate allocations scored by a z-march probe (on MI355X
    the same PD_TV launch runs 10 % faster or slower depending on where its arena lies, docs/kernels/placement.md).  The search
    must be invisible in the results: same bits with it on (3 candidates) and off, and the report must describe it."""
    from tomobar_amd import _lib
    from tomobar_amd.regularisersCuPy import PD_TV_cupy
    ops.set_variant("pdtv", 22)
    L = _lib.lib()
    shape = (40, 2048, 2048)                     # 8 work arrays of 0.67 GB: an arena of 5.4 GB
    rng = np.random.default_rng(11)
    x = (rng.random(shape) * 0.3 + (np.indices(shape)[-1] > shape[-1] // 2)).astype(np.float32)
    xd = dev(x)
    got = {}
    before = ops.placement_tries()
    try:
        for tries in (3, 1):
            _lib.check(L.tomo_release_scratch(0))
            ops.set_placement_tries(tries)
            if tries == 3:   # set-up time reservation: the search runs here, the first prox finds its arena in place
                ops.reserve_tv_scratch(shape, xd.device, "PD_TV", False)
                reserved = ops.placement_last()
                assert reserved is not None
            got[tries] = host(PD_TV_cupy(xd, 0.04, 6, 0, 1, 8.0, 0, False))
            if tries == 3:
                rep = ops.placement_last()
                assert rep == reserved
                assert rep is not None and rep["bytes"] == L.tomo_pdtv_scratch_bytes(shape[2], shape[1], shape[0], 3, 0)
                assert 1 <= len(rep["scores_GBps"]) <= 3 and 0 <= rep["chosen"] < len(rep["scores_GBps"])
                assert all(500.0 < s < 8000.0 for s in rep["scores_GBps"]), rep      # a z-march over HBM
                assert rep["scores_GBps"][rep["chosen"]] == max(rep["scores_GBps"])
    finally:
        ops.set_placement_tries(before)
        _lib.check(L.tomo_release_scratch(0))
    assert np.array_equal(got[3], got[1])
    want = oracle.pd_tv(x[:14], 0.04, 6, 0, 1, 8.0, False)   # the oracle on a slab: 6 iterations reach 6 planes up, 8 of 14 are exact
    assert np.array_equal(got[3][:8], want[:8])


def test_back_projection_relay_scratch_is_per_stream(oracle, ops):
    """tomo_bp3d* re-lays a planar sinogram quad-interleaved into a scratch arena keyed by (device, stream): two streams driving
    ONE context at the same time -- different subsets, different sinograms, launches interleaved -- never share that scratch.
    (Each stream's result equals the oracle; a shared block would mix the sinograms.)  Releasing the arenas and calling again
    re-allocates."""
    from tomobar_amd import _lib
    P, H = make_pair(oracle, (37, 104, 96, 21, -0.75, 3))
    rng = np.random.default_rng(41)
    sinos = [rng.standard_normal((P.nz, len(P.subsets[s]), P.nu)).astype(np.float32) for s in range(3)]
    want = [P.bp(sinos[s], s) for s in range(3)]
    streams = [torch.cuda.Stream() for _ in range(3)]
    dsin = [dev(x) for x in sinos]
    outs = [torch.empty(H.vol_shape(), dtype=torch.float32, device="cuda") for _ in range(3)]
    torch.cuda.synchronize()
    for rep in range(4):                       # interleaved submission, several rounds in flight
        for s, st in enumerate(streams):
            with torch.cuda.stream(st):
                H.backward(dsin[s], s, out=outs[s])
    torch.cuda.synchronize()
    for s in range(3):
        assert np.array_equal(host(outs[s]), want[s]), s
    assert "re-laid quad-interleaved" in H.kernel_path("bp")
    _lib.check(_lib.lib().tomo_release_scratch(0))
    assert np.array_equal(host(H.backward(dsin[1], 1)), want[1])


@pytest.mark.parametrize("shape,iters", [((1023, 2049), 7), ((2500, 777), 30), ((64, 5000), 4), ((3000, 61), 5)])
def test_pdtv_2d_large_images(oracle, ops, shape, iters, pd_arith):
    """The fused 2D kernel (pd_rows2d.inl: three iterations per launch, remainders of two / one) on images with hundreds of
    tiles, ragged edges in both directions, very wide / very tall aspect ratios; float32 and binary16 duals, both TV norms."""
    from tomobar_amd.regularisersCuPy import PD_TV_cupy
    rng = np.random.default_rng(31)
    x = (rng.random(shape) * 0.4 + (np.indices(shape)[-1] > shape[-1] // 3) + 0.5 * (np.indices(shape)[0] % 37 > 18) - 0.3).astype(np.float32)
    for half, mtv, nn in ((False, 0, 1), (True, 1, 0)):
        want = oracle.pd_tv(x, 0.03, iters, mtv, nn, 12.0, half)
        got = host(PD_TV_cupy(dev(x), 0.03, iters, mtv, nn, 12.0, 0, half))
        pd_arith.check(got, want, half=half, what=f"2D {shape} x{iters}")


@pytest.mark.parametrize("flavour", ["shipped", pytest.param("dev", marks=DEV)])
@pytest.mark.parametrize("seed", range(16))
def test_tv_random_shapes(oracle, ops, seed, flavour):
    """seeded random 2D/3D shapes (straddling the 60/62-lane segments, the 4/8-row blocks and the z-chunk boundaries of
    the z-march kernels), random iteration counts (odd counts end with the single-iteration kernel), every option.
    shipped: PD_TV with the reference's roundings (22) and ROF_TV against the oracle, bit for bit, the default PD_TV within
    tolerance (binary16 duals: bit for bit); dev: the builds with the compiler's IEEE sequences (pdtv 2, 21; roftv 2), bit
    for bit, and relaxed arithmetic for both dual types (3) within tolerance."""
    from tomobar_amd.regularisersCuPy import PD_TV_cupy, ROF_TV_cupy
    rng = np.random.default_rng(2000 + seed)
    if seed % 4 == 0:
        shape = (int(rng.integers(2, 200)), int(rng.integers(2, 260)))
    else:
        shape = (int(rng.integers(2, 120)), int(rng.integers(2, 70)), int(rng.integers(2, 200)))
    x = (rng.random(shape) * 0.4 + (np.indices(shape)[-1] > shape[-1] // 3) - 0.3).astype(np.float32)
    iters = int(rng.integers(1, 9))
    half, mtv, nn = bool(rng.integers(0, 2)), int(rng.integers(0, 2)), int(rng.integers(0, 2))
    lam = float(rng.choice([0.01, 0.05, 0.3]))
    want_pd = oracle.pd_tv(x, lam, iters, mtv, nn, 8.0, half)
    want_rof = oracle.rof_tv(x, lam, iters, 0.004, half)
    for v in ((22,) if flavour == "shipped" else (2, 21)):
        ops.set_variant("pdtv", v)
        got = host(PD_TV_cupy(dev(x), lam, iters, mtv, nn, 8.0, 0, half))
        assert np.array_equal(got, want_pd), ("pd", v, shape, iters, half, mtv, nn, np.abs(got - want_pd).max())
    ops.set_variant("roftv", 0 if flavour == "shipped" else 2)
    got = host(ROF_TV_cupy(dev(x), lam, iters, 0.004, 0, half))
    assert np.array_equal(got, want_rof), ("rof", shape, iters, half, np.abs(got - want_rof).max())
    ops.set_variant("pdtv", 0 if flavour == "shipped" else 3)   # relaxed: the shipped default (float32 duals) / dev 3 (both)
    got = host(PD_TV_cupy(dev(x), lam, iters, mtv, nn, 8.0, 0, half))
    if flavour == "shipped" and half:
        assert np.array_equal(got, want_pd), ("pd default, binary16 duals", shape, iters, mtv, nn)
    assert rel(got, want_pd) < (2e-4 if half else 1e-5), ("pd relaxed", shape, iters, half, mtv, nn, rel(got, want_pd))


def test_residual_buffer_must_match_the_contexts_layout(oracle):
    """The residual layout is state on the projector context; a buffer of the other layout would be written past its end (nz not
    a multiple of 4) or misread.  The host class refuses it instead of handing it to the C-ABI, which cannot see buffer sizes."""
    P, H = make_pair(oracle, (6, 36, 40, 22, 0.5, 3))
    rng = np.random.default_rng(5)
    x = dev(rng.random((P.nz, P.n, P.n)).astype(np.float32))
    b = dev(rng.random((P.nz, P.na, P.nu)).astype(np.float32))
    planar = H.residual_buffer(0)
    H.set_residual_layout("zquad")
    try:
        quad = H.residual_buffer(0)
        assert quad.numel() > planar.numel()          # 6 slices -> 2 quads = 8 slice rows
        with pytest.raises(ValueError):
            H.residual(x, b, None, "LS", 0, planar)   # a second driver / thread still holding a planar buffer
        H.residual(x, b, None, "LS", 0, quad)
        with pytest.raises(ValueError):
            H.grad_step(planar, x, torch.empty_like(x), 1e-3, True, 0)
    finally:
        H.set_residual_layout("planar")
    with pytest.raises(ValueError):
        H.grad_step(quad, x, torch.empty_like(x), 1e-3, True, 0)
    H.residual(x, b, None, "LS", 0, planar)
    assert np.array_equal(host(H.residual_as_planar(quad, 0)), host(planar))


@pytest.mark.parametrize("g", [(5, 840, 1000, 40, 3.3, 1),      # one 1024-thread tile, 24 dead pixels, rays leaving the volume sideways
                               (3, 1100, 901, 45, -7.25, 3),     # detector narrower than the volume, tile of 960 threads, subsets
                               (6, 700, 1800, 24, "vec", 1),     # two tiles of 960 of a detector much wider than the volume
                               (2, 1000, 768, 33, 0.0, 1)])      # the smallest tile that takes the lane multipliers
@pytest.mark.parametrize("variant", _v(0, 4))
def test_forward_projection_lane_multipliers_odd_wide_detectors(oracle, ops, g, variant):
    """The per-angle lane -> pixel multipliers of the whole-row form (round 6; from 768-pixel tiles up) on detector widths that are
    not a multiple of the tile, with dead pixels, clipped windows and per-angle offsets: forward projection and the residual epilogue
    in both layouts, bit for bit (variant 4, dev flavour: the same form with pixel = lane)."""
    P, H = make_pair(oracle, g)
    ops.set_variant("fp", variant)
    rng = np.random.default_rng(12)
    vol = rng.standard_normal((P.nz, P.n, P.n)).astype(np.float32)
    b = rng.standard_normal((P.nz, P.na, P.nu)).astype(np.float32)
    for s in ([None] if P.os_number == 1 else [0, P.os_number - 1]):
        want = P.fp(vol, s)
        got = host(H.forward(dev(vol), s))
        assert "whole-row" in H.kernel_path("fp"), H.kernel_path("fp")
        assert np.array_equal(got, want), (g, s, float(np.abs(got - want).max()))
        idx = slice(None) if s is None else P.subsets[s]
        for layout in ("planar", "zquad"):
            H.set_residual_layout(layout)
            try:
                res = H.residual_buffer(s)
                H.residual(dev(vol), dev(b), None, "LS", s, res)
                assert np.array_equal(host(H.residual_as_planar(res, s)), (want - b[:, idx]).astype(np.float32)), (g, s, layout)
            finally:
                H.set_residual_layout("planar")
