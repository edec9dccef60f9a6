"""FBP (the first "next" row): sinc-ramp filter + back projection.
CPU: the oracle's filter vs the reference's own numpy filter (tests/golden/fbp_golden.npz, make_fbp_golden.py).
GPU: RecToolsDIRCuPy.FBP / tomo_fbp_filter vs the oracle."""
import os

import numpy as np
import pytest


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a, np.float64).ravel() - np.asarray(b, np.float64).ravel()) /
                 max(np.linalg.norm(np.asarray(b, np.float64).ravel()), 1e-30))


def test_oracle_filter_vs_reference_numpy_filter(oracle, golden_dir):
    """methodsDIR._filtersinc2D of the reference (a = 1.1, x 1/angles) on even detector widths.  (For odd widths the
    reference's numpy path multiplies the FULL spectrum by a non-symmetric filter and drops the imaginary part, while its
    CuPy path -- the one mirrored here -- filters the half spectrum: the two reference paths disagree there by ~3 %.)"""
    g = np.load(os.path.join(golden_dir, "fbp_golden.npz"))
    for i in (0, 2):
        s = g[f"sino_{i}"]
        mine = oracle.fbp_filter(s[:, None, :], 1.1)[:, 0, :]
        assert rel(mine, g[f"filt_{i}"]) < 1e-5


@pytest.mark.gpu
def test_fbp_filter_kernel_vs_oracle(oracle):
    import ctypes as C
    import torch
    from tomobar_amd import _lib as L
    rng = np.random.default_rng(3)
    for na, nz, nu, cutoff in ((12, 5, 64, 0.35), (7, 3, 90, 0.6), (9, 4, 45, 1.1), (3, 2, 1024, 0.35)):
        x = rng.random((na, nz, nu)).astype(np.float32)
        d = torch.from_numpy(x).cuda()
        L.check(L.lib().tomo_fbp_filter(0, C.c_void_p(d.data_ptr()), na * nz, nu, cutoff, 1.0 / na / nu,
                                        C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        want = oracle.fbp_filter(x, cutoff)
        assert rel(d.cpu().numpy(), want) < 1e-5, (na, nz, nu, rel(d.cpu().numpy(), want))


@pytest.mark.gpu
def test_FBP_end_to_end_vs_oracle(oracle):
    import torch
    from tomobar_amd.methodsDIR_CuPy import RecToolsDIRCuPy
    nz, n, na = 6, 48, 40
    angles = np.linspace(0, np.pi, na, endpoint=False)
    sino = oracle.shepp_logan_sino(n, nz, n, angles)                    # [detY, angles, detX]
    data = np.ascontiguousarray(np.swapaxes(sino, 0, 1))               # [angles, detY, detX]
    P = oracle.Projector(nz, n, n, angles)
    want = oracle.fbp(P, data, 0.35)
    rt = RecToolsDIRCuPy(n, 0, nz, 0.0, angles, n, device_projector=0)
    d = torch.from_numpy(data).cuda()
    rec = rt.FBP(d)
    assert rec.shape == (nz, n, n) and rec.dtype == torch.float32
    assert rel(rec.cpu().numpy(), want) < 1e-5
    assert np.array_equal(d.cpu().numpy(), data), "FBP must not overwrite its input"
    # reconstruction quality sanity: FBP of analytic projections resembles the phantom
    ph = oracle.shepp_logan_3d(n, nz)
    assert np.corrcoef(rec.cpu().numpy().ravel(), ph.ravel())[0, 1] > 0.8  # (the reference's filter is not amplitude-calibrated)
    # other axis order + circular mask + cut-off keyword
    rec2 = rt.FBP(torch.from_numpy(sino).cuda(), data_axes_labels_order=["detY", "angles", "detX"], recon_mask_radius=0.9,
                  cutoff_freq=0.5)
    want2 = oracle.circular_mask(oracle.fbp(P, data, 0.5), 0.9)
    assert rel(rec2.cpu().numpy(), want2) < 1e-5
    # padded detector: the reconstruction grid stays ObjSize (methodsDIR.py:44-69)
    rtp = RecToolsDIRCuPy(n, 5, nz, 0.0, angles, n, device_projector=0)
    Pp = oracle.Projector(nz, n, n + 10, angles)
    padded = np.pad(data, ((0, 0), (0, 0), (5, 5)), mode="edge")
    assert rel(rtp.FBP(d).cpu().numpy(), oracle.fbp(Pp, padded, 0.35)) < 1e-5


@pytest.mark.gpu
def test_2D_geometry_direct_methods_vs_oracle(oracle):
    """BASELINE configs[0] shape of call: 2D geometry (DetectorsDimV=None), data [angles, detX], images [Y, X]
    (reference: RecToolsDIR, methodsDIR.py:71-96,322-371) -- one slice through the same kernels."""
    import torch
    from tomobar_amd.methodsDIR_CuPy import RecToolsDIRCuPy
    n, na = 64, 45
    angles = np.linspace(0, np.pi, na, endpoint=False)
    P = oracle.Projector(1, n, n, angles)
    sino3 = oracle.shepp_logan_sino(n, 1, n, angles)                   # [1, angles, detX]
    sino2 = np.ascontiguousarray(sino3[0])                             # [angles, detX]
    rt = RecToolsDIRCuPy(n, 0, None, 0.0, angles, n, device_projector=0)
    assert rt.geom == "2D"
    rec = rt.FBP(torch.from_numpy(sino2).cuda(), recon_mask_radius=0.95)
    want = oracle.circular_mask(oracle.fbp(P, np.ascontiguousarray(np.swapaxes(sino3, 0, 1)), 0.35), 0.95)[0]
    assert rec.shape == (n, n) and rel(rec.cpu().numpy(), want) < 1e-5
    # transposed labels
    rec_t = rt.FBP(torch.from_numpy(np.ascontiguousarray(sino2.T)).cuda(), data_axes_labels_order=["detX", "angles"],
                   recon_mask_radius=0.95)
    assert np.array_equal(rec_t.cpu().numpy(), rec.cpu().numpy())
    img = oracle.shepp_logan_3d(n, 1)[0]
    fp = rt.FORWPROJ(torch.from_numpy(img).cuda())
    assert fp.shape == (na, n) and np.array_equal(fp.cpu().numpy(), P.fp(img[None])[0])
    bp = rt.BACKPROJ(torch.from_numpy(sino2).cuda())
    assert bp.shape == (n, n) and rel(bp.cpu().numpy(), P.bp(sino3)[0]) < 1e-6
