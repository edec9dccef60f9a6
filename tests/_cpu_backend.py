"""TEST INFRASTRUCTURE: lets the product's Python drivers (RecToolsIRCuPy.FISTA / ADMM / powermethod, dicts_check, the
slab TV drivers) run in a GPU-less process by standing the ORACLE in for the C-ABI library at the two seams the drivers
use -- the projector object (``HipTools3D``) and the ``tomobar_amd.ops`` module.  Only the multi-process gloo tests use
it (tests/test_slab_gloo.py): what they check is the drivers' control flow, slab bookkeeping and collectives, which are
the same Python lines on the GPU.  Nothing here is reachable from the product package."""
import types

import numpy as np
import torch

from oracle import tomo_oracle as O


def _np(t):
    return t.numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


def _put(dst: torch.Tensor, arr):
    dst.copy_(torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float32)).view(dst.shape))
    return dst


class OracleTools3D:
    """The HipTools3D interface the iterative drivers use, computed by the oracle projector on CPU tensors."""

    def __init__(self, detectors_x, detectors_x_pad, detectors_y, angles_vec, centre_of_rotation, recon_size,
                 processing_arch="gpu", device_index=0, ordsub_number=None, lerp8=False):
        self.detectors_x_pad = int(detectors_x_pad)
        self.device_index = int(device_index)
        self.nz, self.n = int(detectors_y), int(recon_size)
        self.nu = int(detectors_x) + 2 * self.detectors_x_pad
        self.ordsub_number = 1 if ordsub_number is None else int(ordsub_number)
        self.P = O.Projector(self.nz, self.n, self.nu, np.asarray(angles_vec, np.float64), centre_of_rotation,
                             self.ordsub_number)
        self.na = self.P.na
        self._device = torch.device("cpu")
        self.vol_geom = {"GridRowCount": self.n, "GridColCount": self.n, "GridSliceCount": self.nz}
        self.slab = None   # set through RecToolsIRCuPy.slab, as on HipTools3D

    # -- the oracle's projector pair; with a vertical CoR component in z-slab mode the row resampling runs on the slab's rows
    #    plus the neighbours' ghost rows, fetched by the PRODUCT's exchange (tomobar_amd.slab.extend_detector_rows), exactly
    #    as HipTools3D._shift_rows does around the HIP kernels
    def _shift(self, sino, sub, sign):
        from tomobar_amd.slab import check_ghost_rows, extend_detector_rows
        P = self.P
        g = int(np.ceil(float(np.abs(P.vshift).max()))) + 1
        if getattr(self, "_vshift_checked", None) is not self.slab:
            check_ghost_rows(self.slab, g, self.nz, "a vertical CoR component")
            self._vshift_checked = self.slab
        ext, lo = extend_detector_rows(self.slab, torch.from_numpy(np.ascontiguousarray(sino)), g)
        whole = types.SimpleNamespace(nz=int(ext.shape[0]), na=P.na, vshift=P.vshift, subsets=P.subsets)
        return np.ascontiguousarray(O.Projector.shift_rows(whole, ext.numpy(), sub, sign)[lo:lo + self.nz])

    def _sharded_shift(self):
        return self.P.vshift is not None and self.slab is not None and self.slab.world > 1

    def _fp(self, vol, sub):
        if not self._sharded_shift():
            return self.P.fp(vol, sub)
        keep, self.P.vshift = self.P.vshift, None
        try:
            sino = self.P.fp(vol, sub)
        finally:
            self.P.vshift = keep
        return self._shift(sino, sub, 1.0)

    def _bp(self, sino, sub):
        if not self._sharded_shift():
            return self.P.bp(sino, sub)
        sino = self._shift(sino, sub, -1.0)
        keep, self.P.vshift = self.P.vshift, None
        try:
            return self.P.bp(sino, sub)
        finally:
            self.P.vshift = keep

    def _idx(self, os_index):
        return slice(None) if (os_index is None or self.ordsub_number == 1) else self.P.subsets[os_index]

    def _sub(self, os_index):
        return None if (os_index is None or self.ordsub_number == 1) else int(os_index)

    def subset_size(self, os_index):
        return self.na if self._sub(os_index) is None else len(self.P.subsets[os_index])

    def vol_shape(self):
        return (self.nz, self.n, self.n)

    def sino_shape(self, os_index=None):
        return (self.nz, self.subset_size(os_index), self.nu)

    def set_residual_layout(self, layout):   # the oracle keeps the planar layout
        pass

    def residual_buffer(self, os_index=None):
        return torch.empty(self.sino_shape(os_index), dtype=torch.float32)

    def forward(self, vol, os_index=None, out=None):
        r = self._fp(np.ascontiguousarray(_np(vol)), self._sub(os_index))
        return torch.from_numpy(r) if out is None else _put(out, r)

    def backward(self, sino, os_index=None, out=None):
        r = self._bp(np.ascontiguousarray(_np(sino)), self._sub(os_index))
        return torch.from_numpy(r) if out is None else _put(out, r)

    def residual(self, vol, b, w, fidelity, os_index, out, gathered=0, robust=None):
        if robust is not None:
            self.residual(vol, b, w, fidelity, os_index, out, gathered)
            return self.robust_apply(out, *robust)
        ax = self._fp(np.ascontiguousarray(_np(vol)), self._sub(os_index))
        idx = self._idx(os_index)
        bs = _np(b)[:, idx, :]
        if fidelity in ("LS", "PWLS"):
            r = ax - bs
            if w is not None:
                r = r * _np(w)[:, idx, :]
        elif fidelity == "KL":
            r = np.float32(1) - bs / np.clip(ax, np.float32(1e-8), None)
        else:
            r = bs / np.clip(ax, np.float32(1e-8), None)
        return _put(out, r)

    # -- ring-artefact data terms (Group-Huber offsets / SWLS): the formulas of oracle.fista, one seam call each
    def residual_ring(self, vol, b, r_x, accelerate, os_index, out):
        ax = self._fp(np.ascontiguousarray(_np(vol)), self._sub(os_index))
        res = ax - _np(b)[:, self._idx(os_index), :]
        return _put(out, res + (np.float32(accelerate) * _np(r_x))[:, None, :])

    def ring_reduce(self, res, w, r_x, l_inv, os_index, r_out):
        r = _np(res)
        vec = np.zeros((self.nz, self.nu), np.float32)
        for a in range(r.shape[1]):
            vec = vec + r[:, a, :]
        _put(r_out, _np(r_x) - np.float32(l_inv) * vec)
        if w is not None:
            _put(res, r * _np(w)[:, self._idx(os_index), :])

    def swls_apply(self, res, w, beta, os_index):
        r, ws_ = _np(res), _np(w)[:, self._idx(os_index), :]
        wr = np.zeros((self.nz, self.nu), np.float32)
        ws = np.zeros((self.nz, self.nu), np.float32)
        for a in range(r.shape[1]):
            wr = wr + ws_[:, a, :] * r[:, a, :]
            ws = ws + ws_[:, a, :]
        q = wr / (ws + np.float32(beta))
        _put(res, ws_ * r - ws_ * q[:, None, :])

    def robust_apply(self, res, mode, delta):
        from oracle import tomo_oracle as O
        return _put(res, O.robust_weight(_np(res), **{mode: delta}))

    def ring_update(self, r, r_old, r_x, lam, beta):
        v = _np(r).copy()
        t = (np.sign(v) * np.maximum(np.abs(v) - np.float32(lam), np.float32(0.0))).astype(np.float32)
        _put(r_x, t + np.float32(beta) * (t - _np(r_old)))
        _put(r, t)
        _put(r_old, t)

    def momentum(self, x, x_old, x_t, beta):
        _put(x_t, _np(x) + np.float32(beta) * (_np(x) - _np(x_old)))

    def invalidate(self):
        pass

    def grad_step(self, res, x_t, x_out, l_inv, nonneg, os_index):
        x = _np(x_t) - np.float32(l_inv) * self._bp(np.ascontiguousarray(_np(res)), self._sub(os_index))
        if nonneg:
            np.maximum(x, 0, out=x)
        _put(x_out, x)

    def grad_step_momentum(self, res, x_t, x_old_then_x, l_inv, beta, nonneg, os_index):
        x = _np(x_t) - np.float32(l_inv) * self._bp(np.ascontiguousarray(_np(res)), self._sub(os_index))
        if nonneg:
            np.maximum(x, 0, out=x)
        xt = x + np.float32(beta) * (x - _np(x_old_then_x))
        _put(x_old_then_x, x)
        _put(x_t, xt)

    def admm_z_update(self, res, z, x, u, zu_out, tau, rho, relax_on, one_minus_alpha, alpha, nonneg, os_index):
        g = self._bp(np.ascontiguousarray(_np(res)), self._sub(os_index))
        z0, xn, un = _np(z).copy(), _np(x), _np(u)
        ga = np.float32(rho) * (z0 - xn + un)
        zn = z0 - np.float32(tau) * (g + ga)
        if nonneg:
            np.maximum(zn, 0, out=zn)
        if relax_on:
            zn = np.float32(one_minus_alpha) * z0 + np.float32(alpha) * zn
        _put(z, zn)
        _put(zu_out, zn + un)


def make_ops():
    """A stand-in for the ``tomobar_amd.ops`` module on CPU tensors (numpy float32 arithmetic, the oracle's TV)."""
    m = types.ModuleType("cpu_ops")
    m.to_device = lambda x, device_index=0: x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x))
    m.contiguous = lambda t: t.contiguous()
    from tomobar_amd import ops as real_ops   # array-library plumbing is pure Python: the product's own functions
    m.like, m.is_cupy, m.base_ptr = real_ops.like, real_ops.is_cupy, real_ops.base_ptr
    m.fill = lambda x, v: x.fill_(float(v))
    m.momentum = lambda x, xo, xt, beta: _put(xt, _np(x) + np.float32(beta) * (_np(x) - _np(xo)))
    m.admm_dual = lambda u, z, x: _put(u, _np(u) + (_np(z) - _np(x)))
    m.scale = lambda a, x, y: _put(y, np.float32(a) * _np(x))
    m.clamp_min = lambda x, lo=0.0: x.clamp_(min=float(lo))
    m.norm2 = lambda x: float(np.sqrt(np.sum(_np(x).astype(np.float64) ** 2)))
    m.dot = lambda x, y: float(np.sum(_np(x).astype(np.float64) * _np(y)))

    def pwls_weights(b, slab=None):
        w = np.maximum(_np(b), np.float32(1e-6))
        wmax = w.max() if slab is None else np.float32(slab.allreduce_max(float(w.max())))
        return torch.from_numpy(w / wmax)

    m.pwls_weights = pwls_weights
    m.pad_edge = lambda b, pad: torch.from_numpy(O.pad_detector(_np(b), pad))
    m.crop_center = lambda v, size: torch.from_numpy(np.ascontiguousarray(O.crop_recon(_np(v), size)))
    m.circ_mask_ = lambda v, radius: _put(v, O.circular_mask(_np(v), radius))

    def pdtv(data, out, sigma, tau, lt, theta, iterations, methodTV, nonneg, half):
        raise AssertionError("whole-volume TV must not be called on a slab rank")

    m.pdtv = m.roftv = pdtv
    m.reserve_tv_scratch = lambda *args, **kw: None   # (slab ranks never reserve whole-volume TV scratch anyway)
    return m


def install(monkeypatch=None):
    """Point the drivers at the stand-ins (module attributes; undone by pytest's monkeypatch when given)."""
    import tomobar_amd.methodsIR_CuPy as IR
    import tomobar_amd.regularisersCuPy as REG
    import tomobar_amd.slab as SL
    import tomobar_amd.supp.dicts as DI
    import tomobar_amd.supp.suppTools as ST
    ops = make_ops()
    # host tensors never reach tomo_halo_pack / tomo_halo_unpack / tomo_pdtv_iters_per_launch: slab.py packs them with
    # plain copies and plans three iterations per launch without a GPU (round 4) -- no stand-in needed for those
    sets = [(IR, "ops", ops), (IR, "HipTools3D", OracleTools3D), (REG, "ops", ops), (DI, "ops", ops), (ST, "ops", ops),
            (SL, "_hip_pd_pair", O.pd_pair_slab), (SL, "_hip_pd_step", O.pd_step_slab), (SL, "_hip_rof_step", O.rof_step_slab)]
    for mod, name, val in sets:
        if monkeypatch is not None:
            monkeypatch.setattr(mod, name, val)
        else:
            setattr(mod, name, val)
