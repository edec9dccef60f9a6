"""Pins the CPU oracle (oracle/tomo_oracle.c) against every data-free literal the reference's own tests hold for
the projector pair.  The upstream tests take their angles from tests/test_data/normalised_data.npz (absent from the
checkout, .MISSING_LARGE_BLOBS); the literals below are reproduced with the 1-degree grid 0..179 deg on a 160-pixel
detector, which is therefore what they pin."""
import numpy as np
import pytest
from numpy.testing import assert_allclose

ANGLES = np.deg2rad(np.arange(180.0))


def test_forwproj3D_ones_cube_literals(oracle):
    # reference: tests/test_RecToolsDIRCuPy.py:671-694 -> min 67.27458, max 225.27428 (rtol 2e-6).
    # Those literals embed the 8-bit interpolation weights of NVIDIA texture units (LERP8 mode).
    P = oracle.Projector(2, 160, 160, ANGLES, flags=oracle.FLAG_LERP8)
    s = P.fp(np.ones((2, 160, 160), np.float32))
    assert_allclose(s.min(), 67.27458, rtol=2e-6)
    assert_allclose(s.max(), 225.27428, rtol=2e-6)
    assert s.shape == (2, 180, 160) and s.dtype == np.float32


def test_forwproj3D_ones_cube_exact_lerp(oracle):
    # exact float32 interpolation: analytic values 2(80 sqrt2 - 79.5) and (160 - 1/sqrt2) sqrt2
    P = oracle.Projector(1, 160, 160, ANGLES)
    s = P.fp(np.ones((1, 160, 160), np.float32))
    assert_allclose(s.min(), 2 * (80 * np.sqrt(2) - 79.5), rtol=2e-6)
    assert_allclose(s.max(), (160 - 1 / np.sqrt(2)) * np.sqrt(2), rtol=2e-6)


@pytest.mark.parametrize("n,os_number,literal", [
    (160, 1, 27550.463),   # tests/test_RecToolsIRCuPy.py:316 (lipschitz_const literal)
    (160, 5, 5510.867),    # tests/test_RecToolsIRCuPy.py:573,682 (rtol 1e-5)
    (280, 5, 9644.283),    # tests/test_RecToolsIRCuPy.py:639 (DetectorsDimH_pad=60 -> N = Nu = 280)
])
def test_power_method_constants(oracle, n, os_number, literal):
    P = oracle.Projector(1, n, n, ANGLES, os_number=os_number)
    x = np.random.default_rng(0).standard_normal((1, n, n)).astype(np.float32)
    assert_allclose(oracle.power_method(P, x), literal, rtol=1e-5)


def test_power_pad50_range(oracle):
    # tests/test_RecToolsIRCuPy.py:273-295: pad 50, OS 5 -> 8000 <= lc <= 9000
    P = oracle.Projector(1, 260, 260, ANGLES, os_number=5)
    x = np.random.default_rng(1).standard_normal((1, 260, 260)).astype(np.float32)
    assert 8000 <= oracle.power_method(P, x) <= 9000


def test_os_index_table(oracle):
    # astra_base.py:195-209 with 12 angles / 3 subsets, and the one-element trim of methodsIR_CuPy.py:454-456
    table, bins, subsets = oracle.os_indices(12, 3)
    assert bins == 4 and table.tolist() == [[0, 3, 6, 9], [1, 4, 7, 10], [2, 5, 8, 11]]
    table, bins, subsets = oracle.os_indices(10, 4)
    assert table.tolist() == [[0, 4, 8], [1, 5, 9], [2, 6, 0], [3, 7, 0]]
    assert [s.tolist() for s in subsets] == [[0, 4, 8], [1, 5, 9], [2, 6], [3, 7]]


def test_fp_matches_analytic_ellipsoids(oracle):
    # model check independent of the reference: line integrals of the voxelised phantom vs analytic chords
    n, nz = 96, 4
    ang = np.linspace(0, np.pi, 30, endpoint=False)
    P = oracle.Projector(nz, n, n, ang)
    s_num = P.fp(oracle.shepp_logan_3d(n, nz))
    s_ana = oracle.shepp_logan_sino(n, nz, n, ang)
    rel = np.linalg.norm(s_num - s_ana) / np.linalg.norm(s_ana)
    assert rel < 0.05, rel  # voxelisation error only


def test_half_round_trip(oracle):
    import struct
    L = oracle.lib()
    vals = np.concatenate([np.random.default_rng(0).standard_normal(2000).astype(np.float32) * s
                           for s in (1e-8, 1e-5, 1e-3, 1, 100, 7e4)] + [np.array([0, -0.0, 65504, 65520, 1e-30], np.float32)])
    ref = vals.astype(np.float16).astype(np.float32)
    got = np.array([L.orc_round_half(float(v)) for v in vals], np.float32)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
