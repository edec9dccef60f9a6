"""Fourier reconstruction on unequally spaced grids (``RecToolsDIRCuPy.FOURIER_INV``).

CPU: the numpy restatement (oracle/fourier_oracle.py) against the fixtures produced by the REFERENCE's own Python driver
and kernel source (tests/golden/make_fourier_golden.py -> fourier_golden.npz), and the product's host-side filter tables
against the oracle's.  GPU: the HIP pipeline behind ``FOURIER_INV`` against the oracle and the fixtures."""
import importlib.util
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
TOL = 1e-5  # relative L2, north_star


def rel(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b.astype(np.float64)) / np.linalg.norm(b.astype(np.float64)))


def _cases():
    sys.path.insert(0, os.path.join(HERE, "golden"))
    spec = importlib.util.spec_from_file_location("make_fourier_golden", os.path.join(HERE, "golden", "make_fourier_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.CASES


CASES = _cases()
IDS = [c[0] for c in CASES]


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(HERE, "golden", "fourier_golden.npz"))


@pytest.fixture(scope="module")
def FO():
    sys.path.insert(0, os.path.dirname(HERE))
    from oracle import fourier_oracle
    return fourier_oracle


def _oracle_run(FO, golden, case):
    name, nz, nproj, dn, rs, cor, span, kw = case
    kw = {k: v for k, v in kw.items() if k not in ("recon_mask_radius", "data_axes_labels_order")}
    return FO.fourier_inv(golden[name + "_sino"], golden[name + "_angles"], cor, rs, **kw)


def _mask(rec, radius):
    """apply_circular_mask of the reference (suppTools.py:364-399) as restated in the oracle"""
    from oracle import tomo_oracle as O
    return O.circular_mask(rec, radius)


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_oracle_vs_reference_fixture(FO, golden, case):
    rec = _oracle_run(FO, golden, case)
    want = golden[case[0] + "_rec"]
    assert rec.shape == want.shape
    if "recon_mask_radius" in case[7]:
        rec = rec * (want != 0)  # the fixture went through the reference's own mask
    assert rel(rec, want) < TOL


def test_product_filter_tables_vs_oracle(FO):
    from tomobar_amd import fourier as FT
    for name in FT.FILTER_NAMES:
        for ne, cutoff in ((128, 1.0), (256, 0.7), (1024, 0.35)):
            a, b = FT.calc_filter(ne, name, cutoff), FO.calc_filter(ne, name, cutoff)
            assert a.dtype == np.float32 and a.shape == (ne // 2 + 1,)
            assert np.allclose(a, b, rtol=1e-6, atol=1e-6 * np.abs(b).max()), name
    assert FT.oversampled_width(40, 40) == FO.oversampled_width(40, 40) == 128
    assert FT.oversampled_width(100, 700) == FO.oversampled_width(100, 700) == 1024
    assert FT.oversampled_width(33, 40, False, 4) == 132
    n = 128
    mu = -np.log(1e-4) / (2 * n * n)
    assert FT.footprint_half_width(n, mu, 1e-4) == FO.footprint_m(n, mu, 1e-4)
    with pytest.raises(ValueError):
        FT.calc_filter(128, "boxcar", 1.0)


@pytest.mark.parametrize("name,median", [("none", 100), ("ramp", 0.496701), ("shepp", 0.447188), ("cosine", 0.25168),
                                         ("cosine2", 0.164889), ("hamming", 0.185245), ("hann", 0.164889),
                                         ("parzen", 0.042508)])
def test_calc_filter_known_answers_of_the_reference(FO, name, median):
    """the data-free literals of the reference's own test (tests/test_fourier.py:5-27): median of calc_filter(100, ., 1.0)"""
    from tomobar_amd import fourier as FT
    for impl in (FO.calc_filter, FT.calc_filter):
        f = np.sort(impl(100, name, 1.0))
        assert f.size == 51
        np.testing.assert_allclose(f[f.size // 2], median, rtol=1e-5)


def test_memory_estimator_dry_run_needs_no_gpu():
    from tomobar_amd.supp.memory_estimator_helpers import DeviceMemStack
    st = DeviceMemStack()
    st.malloc(1000)
    st.malloc(10)
    assert st.current == 1024 + 512 and st.highwater == 1536
    st.free(1000)
    assert st.current == 512 and st.highwater == 1536
    assert DeviceMemStack.instance() is None
    with DeviceMemStack() as s2:
        assert DeviceMemStack.instance() is s2
    assert DeviceMemStack.instance() is None


# ------------------------------------------------------------------------------------------------- GPU
def _tools(case, golden):
    from tomobar_amd.methodsDIR_CuPy import RecToolsDIRCuPy
    name, nz, nproj, dn, rs, cor, span, kw = case
    return RecToolsDIRCuPy(dn, 0, nz, cor, golden[name + "_angles"], rs, device_projector=0)


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_FOURIER_INV_vs_oracle_and_fixture(FO, golden, case):
    import torch
    name, nz, nproj, dn, rs, cor, span, kw = case
    rt = _tools(case, golden)
    sino = golden[name + "_sino"]
    data = sino
    if kw.get("data_axes_labels_order") == ["angles", "detY", "detX"]:
        data = np.ascontiguousarray(np.swapaxes(sino, 0, 1))
    d = torch.from_numpy(data).cuda()
    rec = rt.FOURIER_INV(d, **kw).cpu().numpy()
    want = golden[name + "_rec"]
    assert rec.shape == want.shape and rec.dtype == np.float32
    assert rel(rec, want) < TOL, "vs the reference's own output"
    orc = _oracle_run(FO, golden, case)
    if "recon_mask_radius" in kw:
        orc = _mask(orc, kw["recon_mask_radius"])
    assert rel(rec, orc) < TOL, "vs the oracle"
    assert np.array_equal(d.cpu().numpy(), data), "FOURIER_INV must not overwrite its input"


@pytest.mark.gpu
def test_FOURIER_INV_many_slices_two_chunks(FO):
    """more than 128 slices: two chunks of slice pairs, the second one partial, odd height"""
    import torch
    from tomobar_amd.methodsDIR_CuPy import RecToolsDIRCuPy
    nz, nproj, dn = 141, 24, 40
    angles = np.linspace(0, np.pi, nproj, endpoint=False)
    rng = np.random.default_rng(4)
    base = rng.random((3, nproj, dn), dtype=np.float32)
    sino = np.stack([base[k % 3] * np.float32(1 + 0.01 * k) for k in range(nz)])
    rt = RecToolsDIRCuPy(dn, 0, nz, 0.0, angles, dn, device_projector=0)
    rec = rt.FOURIER_INV(torch.from_numpy(sino).cuda(), filter_type="hamming").cpu().numpy()
    want = FO.fourier_inv(sino, angles, 0.0, dn, filter_type="hamming")
    assert rec.shape == want.shape == (nz, dn, dn)
    assert rel(rec, want) < TOL
    worst = max(rel(rec[k], want[k]) for k in range(nz))
    assert worst < 5 * TOL, worst


@pytest.mark.gpu
def test_FOURIER_INV_errors_and_dry_run(golden):
    import torch
    from tomobar_amd.methodsDIR_CuPy import RecToolsDIRCuPy
    from tomobar_amd.supp.memory_estimator_helpers import DeviceMemStack
    angles = np.linspace(0, np.pi, 20, endpoint=False)
    rt = RecToolsDIRCuPy(32, 0, 4, 0.0, angles, 48, device_projector=0)
    with pytest.raises(ValueError, match="should not be larger"):
        rt.FOURIER_INV(torch.zeros((4, 20, 32), device="cuda"))
    rt = RecToolsDIRCuPy(32, 0, 4, 0.0, angles, 32, device_projector=0)
    with pytest.raises(ValueError):
        rt.FOURIER_INV(torch.zeros((4, 21, 32), device="cuda"))
    rec = rt.FOURIER_INV(torch.ones((4, 20, 32), device="cuda"), filter_type="not-a-filter")  # falls back to shepp
    assert rec.shape == (4, 32, 32) and bool(torch.isfinite(rec).all())
    with DeviceMemStack() as stack:
        shape = rt.FOURIER_INV((4, 20, 32), data_dtype=np.float32)
    assert shape == (4, 32, 32) and stack.highwater > 4 * 20 * 32 * 4 and stack.current == (4 * 20 * 32 * 4 + 4 * 32 * 32 * 4)


@pytest.mark.gpu
def test_FOURIER_INV_medium_size_not_a_power_of_two(FO):
    """200-wide detector (grid 400^2: partial 64-column tiles and 8-point blocks), a prime number of angles over an
    arbitrary range, rotation-axis offset, reconstruction smaller than the detector: whole-grid centre gathering"""
    import torch
    from tomobar_amd.methodsDIR_CuPy import RecToolsDIRCuPy
    nz, nproj, dn, rs, cor = 6, 97, 200, 180, 1.7
    angles = 0.4 + np.linspace(0, 1.1 * np.pi, nproj, endpoint=False)
    rng = np.random.default_rng(8)
    sino = rng.random((nz, nproj, dn), dtype=np.float32)
    rt = RecToolsDIRCuPy(dn, 0, nz, cor, angles, rs, device_projector=0)
    rec = rt.FOURIER_INV(torch.from_numpy(sino).cuda(), filter_type="cosine", cutoff_freq=0.9).cpu().numpy()
    want = FO.fourier_inv(sino, angles, cor, rs, filter_type="cosine", cutoff_freq=0.9)
    assert rec.shape == want.shape == (nz, rs, rs)
    assert rel(rec, want) < TOL
    # horizontal detector padding of the class (DetectorsDimH_pad) widens the Fourier grid: n = 200 + 2 * 12
    rtp = RecToolsDIRCuPy(dn, 12, nz, cor, angles, rs, device_projector=0)
    recp = rtp.FOURIER_INV(torch.from_numpy(sino).cuda()).cpu().numpy()
    wantp = FO.fourier_inv(sino, angles, cor, rs, detectors_x_pad=12)
    assert rel(recp, wantp) < TOL
