"""The multi-GPU configurations of BASELINE.json at the size ONE rank really holds (round 6):

  * configs[3]: a quarter of 2048^2 x 1024 = 256 slices of 2048^2, 1500 angles, no subsets, ADMM + ROF_TV
  * configs[4]: an eighth of 2560^2 x 2160 = 270 slices of 2560^2, 1800 angles in 12 subsets (150 per subset), FISTA-OS +
    PD_TV + Group-Huber ring term

Until round 5 the suite ran these geometries on 70 / 50 slices (tests/test_gpu_fullsize.py); the whole shares ran only in bench
lines.  Here every kernel of a rank's sub-iteration runs on the whole share against the oracle through the same size-independent
properties: A, A^T and their fused epilogues act slice by slice and commute exactly with a power-of-two scale per slice, so a
volume whose slice k is 2^(k%5-2) x base must give 2^(k%5-2) x (the ORACLE's one-slice result) in every slice, bit for bit; the
TV operators move information by a bounded number of cells per iteration, so the oracle on a block pins the cells whose
dependency cone stays inside it (the cone argument of test_gpu_fullsize.py, here in all three directions).  The RCCL legs (4 and 8 ranks) are the part that needs more
than one GPU: tests/test_gpu_slab*.py, skipped on a one-GPU box.
"""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from test_gpu_fullsize import _scales, _z_varying_volume  # noqa: E402


def test_config3_share_256_slices_residual_and_admm_update(oracle):
    """2048^2 x 256 x 1500: forward projector (dense-angle form) with the LS residual epilogue in the quad-interleaved layout the
    ADMM driver uses, then the brick back projector with the fused ADMM z-update (relaxed and not), on the whole share."""
    from tomobar_amd.projector import HipTools3D
    n, nz, na = 2048, 256, 1500
    angles = np.linspace(0, np.pi, na, endpoint=False)
    H = HipTools3D(n, 0, nz, angles, -3.25, n, "gpu", 0, None)
    P1 = oracle.Projector(1, n, n, angles, -3.25, 1)
    rng = np.random.default_rng(31)
    base_v = rng.random((1, n, n), dtype=np.float32)
    base_b = (rng.random((1, na, n), dtype=np.float32) * np.float32(900.0)).astype(np.float32)
    ax1 = P1.fp(base_v, None)
    r1 = (ax1 - base_b).astype(np.float32)
    grad1 = P1.bp(r1, None)
    sc = _scales(nz, "cuda")
    vol = torch.from_numpy(base_v).cuda() * sc
    b = torch.from_numpy(base_b).cuda() * sc
    H.set_residual_layout("zquad")
    try:
        res = H.residual_buffer(None)
        assert res.dim() == 4 and res.shape[0] == nz // 4
        H.residual(vol, b, None, "LS", None, res)
        path = H.kernel_path("fp")
        print("configs[3] share forward-projection path:", path)
        assert "dense(256 pixels x 16 angles" in path, path
        got = H.residual_as_planar(res, None)
        want = torch.from_numpy(r1).cuda() * sc
        assert torch.equal(got, want), float((got - want).abs().max())
        del got, want, b
        # fused ADMM z-update on that residual: z <- P+(z - tau (A^T r + rho (z - x + u))), relaxed, and z + u
        x1 = rng.random((1, n, n), dtype=np.float32) * np.float32(0.05)
        u1 = rng.standard_normal((1, n, n)).astype(np.float32) * np.float32(0.01)
        tau, rho, al = np.float32(1.0 / 4096.0), np.float32(1.7), 1.6
        x_d = torch.from_numpy(x1).cuda() * sc
        u_d = torch.from_numpy(u1).cuda() * sc
        for relax_on in (True, False):
            zn = base_v - tau * (grad1 + rho * (base_v - x1 + u1))
            zn = np.maximum(zn, 0)
            if relax_on:
                zn = np.float32(1.0 - al) * base_v + np.float32(al) * zn
            z_d = vol.clone()
            zu_d = torch.empty_like(z_d)
            H.admm_z_update(res, z_d, x_d, u_d, zu_d, tau, rho, relax_on, np.float32(1.0 - al), np.float32(al), True, None)
            assert "brick" in H.kernel_path("bp"), H.kernel_path("bp")
            wz = torch.from_numpy(zn.astype(np.float32)).cuda() * sc
            assert torch.equal(z_d, wz), (relax_on, float((z_d - wz).abs().max()))
            wzu = torch.from_numpy((zn + u1).astype(np.float32)).cuda() * sc
            assert torch.equal(zu_d, wzu), (relax_on, float((zu_d - wzu).abs().max()))
            del z_d, zu_d, wz, wzu
    finally:
        H.set_residual_layout("planar")


def _valid(lo, hi, size, reach):
    """the part of [lo, hi) whose dependency cone of `reach` cells stays inside the block or ends at a true face"""
    return (lo if lo == 0 else lo + reach), (hi if hi == size else hi - reach)


def _check_blocks(got, vol, run_oracle, reach, blocks, Z, B, check, what):
    """Cone argument in all three directions: one TV iteration moves information by a bounded number of cells in z, y and x, so
    the oracle on a Z x B x B BLOCK of the input pins the cells whose cone of `reach` cells stays inside the block or ends at a
    true face of the volume.  Blocks at the places the kernels treat differently (corners, faces, z-chunk seams, ragged last
    x-segment / y-tile) replace 40 whole planes of 17-26 MB each, which the CPU oracle needs minutes for."""
    nz, dy, dx = vol.shape
    checked = 0
    for z0, y0, x0 in blocks:
        blk = vol[z0:z0 + Z, y0:y0 + B, x0:x0 + B].contiguous().cpu().numpy()
        want = run_oracle(blk)
        (za, zb), (ya, yb), (xa, xb) = _valid(z0, z0 + Z, nz, reach), _valid(y0, y0 + B, dy, reach), _valid(x0, x0 + B, dx, reach)
        w = torch.from_numpy(np.ascontiguousarray(want[za - z0:zb - z0, ya - y0:yb - y0, xa - x0:xb - x0])).cuda()
        g = got[za:zb, ya:yb, xa:xb]
        assert g.shape == w.shape and w.shape[0] >= 16, (g.shape, w.shape)
        check(g, w, f"{what}, block at {(z0, y0, x0)}")
        checked += w.numel()
        # the block's boundary IS artificial: one cell inside it the block result differs from the whole-volume one (the
        # perturbation decays per cell it crosses, so nothing can be said about the first cell outside the cone)
        if x0 > 0:
            assert not np.array_equal(want[za - z0:zb - z0, ya - y0:yb - y0, 1], got[za:zb, ya:yb, x0 + 1].cpu().numpy())
    print(f"{what}: {checked} voxels in {len(blocks)} blocks compared with the oracle")


def test_config3_share_256_slices_roftv_cone(oracle):
    """ROF_TV (the prox of configs[3]) on a z-varying 256 x 2048^2 share, 6 iterations (reach 2 cells per iteration), bit for
    bit on the cone-valid cells of ten blocks."""
    from tomobar_amd.regularisersCuPy import ROF_TV_cupy
    nz, dy, dx, iters = 256, 2048, 2048, 6
    vol = _z_varying_volume(nz, dy, dx)
    got = ROF_TV_cupy(vol, 0.04, iters, 0.005, 0, False)
    B, Z = 384, 40
    blocks = [(0, 0, 0), (0, dy - B, dx - B), (nz - Z, 0, dx - B), (nz - Z, dy - B, 0),
              (108, 832, 832), (108, dy - B, 1000), (64 - Z // 2, 700, dx - B), (128 - Z // 2, 0, 1500),
              (200, 1600, 40), (nz - Z, 1000, 1000)]

    def check(g, w, what):
        assert torch.equal(g, w), (what, float((g - w).abs().max()))
    _check_blocks(got, vol, lambda blk: oracle.rof_tv(blk, 0.04, iters, 0.005, False), 2 * iters, blocks, Z, B, check,
                  "configs[3] share ROF_TV cone")


def test_config4_share_270_slices_ring_residual_and_fista_step(oracle):
    """2560^2 x 270, subsets of 150 of 1800 angles: the 3-pass whole-row forward projector with the Group-Huber ring residual
    (full-sinogram b addressed through the subset's angle indices), the offsets' reduction, and the brick back projector with
    the FISTA epilogue -- ragged last z-brick (270 = 16 x 16 + 14) and ragged last slice quad (270 = 67 x 4 + 2)."""
    from tomobar_amd.projector import HipTools3D
    n, nz, na, os_n, sub = 2560, 270, 1800, 12, 5
    angles = np.linspace(0, np.pi, na, endpoint=False)
    H = HipTools3D(n, 0, nz, angles, 1.75, n, "gpu", 0, os_n)
    P1 = oracle.Projector(1, n, n, angles, 1.75, os_n)
    idx = P1.subsets[sub]
    assert H.subset_size(sub) == 150 == len(idx)
    rng = np.random.default_rng(41)
    base_v = rng.random((1, n, n), dtype=np.float32)
    base_b = (rng.random((1, len(idx), n), dtype=np.float32) * np.float32(1100.0)).astype(np.float32)
    base_rx = rng.standard_normal((1, n)).astype(np.float32)
    acc, l_inv = np.float32(50.0), np.float32(1.0 / 4096.0)
    ax1 = P1.fp(base_v, sub)
    res1 = (ax1 - base_b) + (acc * base_rx)[:, None, :]
    vec = np.zeros((1, n), np.float32)
    for a in range(res1.shape[1]):
        vec = vec + res1[:, a, :]
    r1 = base_rx - l_inv * vec
    grad1 = P1.bp(np.ascontiguousarray(res1, dtype=np.float32), sub)
    sc = _scales(nz, "cuda")
    vol = torch.from_numpy(base_v).cuda() * sc
    b_full = torch.zeros((nz, na, n), dtype=torch.float32, device="cuda")
    b_full[:, torch.from_numpy(idx).cuda(), :] = torch.from_numpy(base_b).cuda() * sc
    r_x = torch.from_numpy(base_rx).cuda() * sc.view(nz, 1)
    res = H.residual_buffer(sub)
    assert res.dim() == 3   # the ring terms read the residual as [detY, angles, detX]
    H.residual_ring(vol, b_full, r_x, acc, sub, res)
    path = H.kernel_path("fp")
    print("configs[4] share forward-projection path:", path)
    assert "whole-row" in path and "march" not in path and "sync" not in path, path
    want = torch.from_numpy(res1.astype(np.float32)).cuda() * sc
    assert torch.equal(res, want), float((res - want).abs().max())
    del want, b_full
    r_out = torch.empty_like(r_x)
    H.ring_reduce(res, None, r_x, l_inv, sub, r_out)
    wr = torch.from_numpy(r1.astype(np.float32)).cuda() * sc.view(nz, 1)
    assert torch.equal(r_out, wr), float((r_out - wr).abs().max())
    # FISTA epilogue on the same residual: X = P+(x_t - A^T res / L) with x_t = the volume itself
    out = torch.empty_like(vol)
    H.grad_step(res, vol, out, l_inv, True, sub)
    assert "brick" in H.kernel_path("bp"), H.kernel_path("bp")
    X1 = np.maximum(base_v - l_inv * grad1, 0).astype(np.float32)
    wX = torch.from_numpy(X1).cuda() * sc
    assert torch.equal(out, wX), float((out - wX).abs().max())


def test_config4_share_270_slices_pdtv_cone(oracle, pd_arith):
    """PD_TV (the prox of configs[4]) on a z-varying 270 x 2560^2 share, 9 iterations = launch plan [3, 3, 3]: the first, the
    STEADY-STATE (reads and writes the duals) and the last instantiation; one cell of reach per iteration; ten blocks."""
    from tomobar_amd.regularisersCuPy import PD_TV_cupy
    nz, dy, dx, iters = 270, 2560, 2560, 9
    vol = _z_varying_volume(nz, dy, dx)
    got = PD_TV_cupy(vol, 0.04, iters, 0, 1, 12.0, 0, False)
    B, Z = 384, 40
    blocks = [(0, 0, 0), (0, dy - B, dx - B), (nz - Z, 0, dx - B), (nz - Z, dy - B, 0),      # corners
              (115, 1088, 1088), (115, dy - B, 1000), (64 - Z // 2, 700, dx - B), (128 - Z // 2, 0, 1500),   # interior, faces, z seams
              (230, 2000, 40), (nz - Z, 1300, 1300)]
    _check_blocks(got, vol, lambda blk: oracle.pd_tv(blk, 0.04, iters, 0, 1, 12.0, False), iters, blocks, Z, B,
                  lambda g, w, what: pd_arith.check(g, w, half=False, what=what), "configs[4] share PD_TV cone")
