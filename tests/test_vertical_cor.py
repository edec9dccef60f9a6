"""CenterRotOffset given as [angles, 2] = (horizontal, vertical) offsets (reference: supp/funcs.py:52-55, the detector
centre of angle a moves to z = CenterRotOffset[a, 1]).  Parallel rays stay inside their slice, so the operator is the
per-slice one composed with a per-angle linear resampling of the detector rows.  ASTRA (which would pin this) is not in
the image, so the oracle side is a formula-level restatement; what IS checked here:

  * CPU: the resampling is consistent with geometry (an integer shift of every angle == projecting the z-shifted volume)
    and the back projector's resampling is the exact adjoint of the forward one;
  * GPU: HipTools3D and RecToolsIRCuPy.FISTA / ADMM with a vertical component equal the oracle's composition bit for bit.
"""
import numpy as np
import pytest


def _geometry(oracle, nz=10, n=48, na=36, os_n=1, seed=0, integer=None):
    rng = np.random.default_rng(seed)
    angles = np.linspace(0, np.pi, na, endpoint=False)
    cor = np.zeros((na, 2))
    cor[:, 0] = rng.uniform(-2, 2, na)
    cor[:, 1] = integer if integer is not None else rng.uniform(-2.5, 2.5, na)
    return angles, cor, oracle.Projector(nz, n, n, angles, cor, os_n)


def test_integer_vertical_shift_is_a_slice_shift(oracle):
    nz, n = 10, 48
    angles, cor, P = _geometry(oracle, nz, n, integer=2.0)
    P0 = oracle.Projector(nz, n, n, angles, cor[:, 0], 1)
    vol = np.random.default_rng(1).random((nz, n, n), dtype=np.float32)
    shifted = np.zeros_like(vol)
    shifted[:-2] = vol[2:]  # detector row r looks at slice r + 2
    assert np.array_equal(P.fp(vol), P0.fp(shifted))
    sino = np.random.default_rng(2).standard_normal((nz, len(angles), n)).astype(np.float32)
    want = np.zeros_like(vol)
    want[2:] = P0.bp(sino)[:-2]  # slice k is seen by detector row k - 2
    assert np.array_equal(P.bp(sino), want)


def test_row_resampling_adjoint(oracle):
    nz, n = 9, 32
    angles, cor, P = _geometry(oracle, nz, n, na=20, seed=3)
    rng = np.random.default_rng(4)
    x = rng.standard_normal((nz, len(angles), n)).astype(np.float32)
    y = rng.standard_normal((nz, len(angles), n)).astype(np.float32)
    lhs = np.vdot(P.shift_rows(x, None, 1.0).astype(np.float64), y.astype(np.float64))
    rhs = np.vdot(x.astype(np.float64), P.shift_rows(y, None, -1.0).astype(np.float64))
    assert abs(lhs - rhs) <= 1e-5 * max(abs(lhs), 1.0)


# ---------------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("os_n", [1, 5])
def test_gpu_projector_pair_with_vertical_component(oracle, os_n):
    import torch
    from tomobar_amd.projector import HipTools3D
    nz, n, na = 11, 96, 50
    angles, cor, P = _geometry(oracle, nz, n, na, os_n, seed=5)
    H = HipTools3D(n, 0, nz, angles, cor, n, "gpu", 0, os_n)
    assert H.has_vertical_shift
    rng = np.random.default_rng(6)
    vol = rng.random((nz, n, n), dtype=np.float32)
    for sub in ([None] if os_n == 1 else range(os_n)):
        got = H.forward(torch.from_numpy(vol).cuda(), sub)
        assert np.array_equal(got.cpu().numpy(), P.fp(vol, sub)), sub
        sino = rng.standard_normal(P.fp(vol, sub).shape).astype(np.float32)
        assert np.array_equal(H.backward(torch.from_numpy(sino).cuda(), sub).cpu().numpy(), P.bp(sino, sub)), sub
    # vec-geometry table carries the vertical component (supp/funcs.py:55,59-60)
    geom = H.proj_geom if os_n == 1 else H.proj_geom_OS[0]
    idx = np.arange(na) if os_n == 1 else H.subset_indices(0)
    assert np.allclose(geom["Vectors"][:, 5], cor[idx, 1])


@pytest.mark.gpu
@pytest.mark.parametrize("method,fid", [("FISTA", "LS"), ("FISTA", "PWLS"), ("FISTA", "KL"), ("ADMM", "LS"), ("OSEM", "KL")])
def test_gpu_reconstruction_with_vertical_component(oracle, method, fid):
    import torch
    from tomobar_amd.methodsIR_CuPy import RecToolsIRCuPy
    nz, n, na, os_n = 10, 80, 60, 4
    angles, cor, P = _geometry(oracle, nz, n, na, os_n, seed=7)
    # the data are full sinograms: project with the no-subset operator of the same geometry
    Pf = oracle.Projector(nz, n, n, angles, cor, 1)
    sino = Pf.fp(np.random.default_rng(8).random((nz, n, n), dtype=np.float32)) / n + np.float32(0.01)
    Lc = oracle.power_method(P, np.random.default_rng(9).standard_normal((nz, n, n)).astype(np.float32))
    reg = {"method": "PD_TV", "regul_param": 0.002, "iterations": 6, "methodTV": 0, "PD_LipschitzConstant": 12.0}
    from tomobar_amd import ops
    ops.set_variant("pdtv", 22)   # the projector composition is under test: PD_TV with the reference's roundings keeps it bit for bit
    rt = RecToolsIRCuPy(n, 0, nz, cor, angles, n, 0, os_n)
    data = {"projection_data": sino, "data_axes_labels_order": ["detY", "angles", "detX"], "data_fidelity": fid}
    alg = {"iterations": 3, "lipschitz_const": Lc, "nonnegativity": True, "recon_mask_radius": None}
    if method == "FISTA":
        want = oracle.fista(P, sino, 3, Lc, True, reg, fid)
        got = rt.FISTA(data, alg, dict(reg))
    elif method == "ADMM":
        want = oracle.admm(P, sino, 3, Lc, 1.0, 1.6, True, reg, fid)
        got = rt.ADMM(data, alg, dict(reg))
    else:
        want = oracle.osem(P, sino, 3, True, reg)
        got = rt.OSEM(data, alg, dict(reg))
    torch.cuda.synchronize()
    assert np.array_equal(got.cpu().numpy(), want), float(np.abs(got.cpu().numpy() - want).max())
