"""BASELINE configs[0] ("plumbing, no GPU"): 2D phantom 256^2, 180 angles, FBP through
``RecToolsDIR(..., device_projector="cpu")`` -- the numpy sinc-ramp filter + a host back projection -- against the fixture
made by the REFERENCE's own class (tests/golden/make_cfg1_golden.py), and the host projector pair of libtomo_mi355x.so
against the oracle.  Runs without a GPU."""
import os
import time

import numpy as np
import pytest


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a, np.float64) - b) / max(np.linalg.norm(b), 1e-30))


@pytest.fixture(scope="module")
def cfg1(golden_dir):
    return np.load(os.path.join(golden_dir, "cfg1_golden.npz"))


def test_cfg1_fbp_cpu_against_reference_class(cfg1):
    from tomobar_amd.methodsDIR import RecToolsDIR
    n, na = 256, 180
    rt = RecToolsDIR(DetectorsDimH=n, DetectorsDimH_pad=0, DetectorsDimV=None, CenterRotOffset=0.0,
                     AnglesVec=cfg1["angles"], ObjSize=n, device_projector="cpu")
    t0 = time.perf_counter()
    rec = rt.FBP(cfg1["sino"].copy(), recon_mask_radius=0.95)
    dt = time.perf_counter() - t0
    print(f"configs[0] (256^2, 180 angles) FBP on the host: {dt * 1e3:.1f} ms")
    assert rec.shape == (n, n) and rec.dtype == np.float32
    assert rel(rec, cfg1["fbp"]) < 1e-5, rel(rec, cfg1["fbp"])
    # padded detector: the reconstruction grid stays ObjSize (methodsDIR.py:44-69), no mask requested
    rt = RecToolsDIR(n, 16, None, 0.0, cfg1["angles"], n, device_projector="cpu")
    assert rel(rt.FBP(cfg1["sino"].copy()), cfg1["fbp_pad"]) < 1e-5
    # data given as [detX, angles]
    rt = RecToolsDIR(n, 0, None, 0.0, cfg1["angles"], n, device_projector="cpu")
    rec2 = rt.FBP(np.ascontiguousarray(cfg1["sino"].T), data_axes_labels_order=["detX", "angles"], recon_mask_radius=0.95)
    assert np.array_equal(rec2, rec)


def test_filter_against_reference_filtersinc2d(golden_dir):
    from tomobar_amd.methodsDIR import _filtersinc2D
    g = np.load(os.path.join(golden_dir, "fbp_golden.npz"))
    for i in range(3):
        assert rel(_filtersinc2D(g[f"sino_{i}"]), g[f"filt_{i}"]) < 1e-6


def test_host_projector_pair_equals_oracle(oracle):
    from tomobar_amd.methodsDIR import RecToolsDIR
    for n, nu_pad, na in ((37, 0, 23), (64, 5, 40)):
        angles = np.linspace(0.05, np.pi + 0.05, na, endpoint=False)
        rt = RecToolsDIR(n, nu_pad, 0, None, angles, n, device_projector="cpu")
        nu = n + 2 * nu_pad
        P = oracle.Projector(1, n, nu, angles, 0.0, 1)
        rng = np.random.default_rng(n)
        img = rng.random((n, n), dtype=np.float32)
        assert np.array_equal(rt.Atools._forwproj(img), P.fp(img[None])[0])
        sino = rng.standard_normal((na, n)).astype(np.float32)        # un-padded data: BACKPROJ pads the detector
        want = P.bp(np.pad(sino, ((0, 0), (nu_pad, nu_pad)), mode="edge")[None])[0]
        assert np.array_equal(rt.BACKPROJ(sino), want)
        out = rt.FORWPROJ(img, data_axes_labels_order=["detX", "angles"])
        assert out.shape == (nu, na)


def test_cpu_device_restrictions():
    from tomobar_amd.methodsDIR import RecToolsDIR
    angles = np.linspace(0, np.pi, 10, endpoint=False)
    with pytest.raises(ValueError):   # no 3D projector on the CPU (astra_tools3d.py:56-59)
        RecToolsDIR(16, 0, 4, 0.0, angles, 16, device_projector="cpu")
    with pytest.raises(ValueError):   # the CPU path rejects a centre-of-rotation offset (astra_base.py:150-153)
        RecToolsDIR(16, 0, None, 1.5, angles, 16, device_projector="cpu")
    with pytest.raises(ValueError):
        RecToolsDIR(16, 0, None, 0.0, angles, 16, device_projector="tpu")


@pytest.mark.gpu
def test_rectoolsdir_gpu_device_wraps_the_device_class(cfg1):
    """device_projector='gpu' / an index: numpy in, numpy out around RecToolsDIRCuPy (same kernels as the iterative path)."""
    import torch
    from tomobar_amd.methodsDIR import RecToolsDIR
    from tomobar_amd.methodsDIR_CuPy import RecToolsDIRCuPy
    n = 256
    host = RecToolsDIR(n, 0, None, 0.0, cfg1["angles"], n, device_projector=0)
    dev = RecToolsDIRCuPy(n, 0, None, 0.0, cfg1["angles"], n, device_projector=0)
    a = host.FBP(cfg1["sino"].copy(), recon_mask_radius=0.95)
    b = dev.FBP(torch.from_numpy(cfg1["sino"].copy()).cuda(), recon_mask_radius=0.95).cpu().numpy()
    assert isinstance(a, np.ndarray) and np.array_equal(a, b)
    # the GPU FBP uses the sinc filter with its default cut-off 0.35 (methodsDIR_CuPy.py:114-150), the CPU path a = 1.1:
    # both are the reference's choices; the back projections behind them are the same operator
    cpu = RecToolsDIR(n, 0, None, 0.0, cfg1["angles"], n, device_projector="cpu")
    s = np.random.default_rng(0).standard_normal((180, n)).astype(np.float32)
    assert np.array_equal(cpu.BACKPROJ(s), host.BACKPROJ(s))
