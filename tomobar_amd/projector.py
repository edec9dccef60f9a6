"""Parallel-beam 3D projector object: the drop-in for the reference's ``Atools`` attribute
(``tomobar/astra_wrappers/astra_tools3d.py:19-110`` on top of ``astra_base.py:34-308,518-606``).

All geometry lives in one native context created once (``tomo_ctx_create``); nothing is built per call.
The attribute surface the reconstruction classes read is kept: ``vol_geom``, ``proj_geom``, ``proj_geom_OS``,
``newInd_Vec``, ``NumbProjBins``, ``detectors_x_pad``, ``device_index`` and the four
``_forwproj[OS]CuPy`` / ``_backproj[OS]CuPy`` methods.
"""

from __future__ import annotations

import ctypes as C
from typing import Optional, Union

import numpy as np
import torch

from . import _lib as L
from . import ops


def geom_size(geom: dict) -> tuple:
    """Shape of the array a geometry describes (``astra.geom_size``): volume (Z, Y, X); projections (detY, angles, detX)."""
    if "GridRowCount" in geom:
        return (geom["GridSliceCount"], geom["GridRowCount"], geom["GridColCount"])
    return (geom["DetectorRowCount"], geom["Vectors"].shape[0], geom["DetectorColCount"])


def vec_geom_init3D(angles_rad, DetectorSpacingX, DetectorSpacingY, CenterRotOffset) -> np.ndarray:
    """The 12-vector-per-angle table of ``supp/funcs.py:45-65`` (ray, detector centre, u, v), vectorised."""
    th = np.asarray(angles_rad, dtype=np.float64).ravel()
    c, s = np.cos(th), np.sin(th)
    cor = np.asarray(CenterRotOffset, dtype=np.float64)
    if cor.ndim == 0:
        c0, c1 = np.full_like(th, float(cor)), np.zeros_like(th)
    elif cor.ndim == 1:
        c0, c1 = cor, np.zeros_like(th)
    else:
        c0, c1 = cor[:, 0], cor[:, 1]
    out = np.zeros((th.size, 12))
    out[:, 0], out[:, 1] = s, -c                       # Rz(theta) . (0,-1,0)
    out[:, 3], out[:, 4], out[:, 5] = c * c0, s * c0, c1  # Rz(theta) . (c0,0,c1)
    out[:, 6], out[:, 7] = c * DetectorSpacingX, s * DetectorSpacingX
    out[:, 11] = DetectorSpacingY
    return out


class HipTools3D:
    """MI355X projector with the AstraTools3D interface."""

    def __init__(self, detectors_x: int, detectors_x_pad: int, detectors_y: int, angles_vec: np.ndarray,
                 centre_of_rotation: Union[float, np.ndarray], recon_size: int, processing_arch: str = "gpu",
                 device_index: int = 0, ordsub_number: Optional[int] = None, lerp8: bool = False):
        # ---- validation with the reference's messages (astra_base.py:74-193)
        if detectors_x <= 0:
            raise ValueError("The size of the horizontal detector cannot be negative or zero")
        if detectors_x_pad < 0:
            raise ValueError("The padding size of the horizontal detector cannot be negative")
        if len(angles_vec) == 0:
            raise ValueError("The length of angles array cannot be zero")
        if np.ndim(angles_vec) >= 2:
            raise ValueError("The array of angles must be 1D")
        if np.ndim(centre_of_rotation) == 1 and len(centre_of_rotation) != len(angles_vec):
            raise ValueError("The CoR must be a scalar or a 1D array of the SAME size as angles")
        if centre_of_rotation is None:
            centre_of_rotation = 0.0
        if isinstance(recon_size, tuple):
            raise ValueError("Reconstruction is currently available for squared or cubic objects only, please provide a scalar")
        if recon_size <= 0:
            raise ValueError("The size of the reconstruction object cannot be zero")
        if processing_arch not in ("cpu", "gpu"):
            raise ValueError("Please choose the processing architecture to be either 'cpu' or 'gpu'")
        if processing_arch == "cpu":
            raise ValueError("3D CPU reconstruction is not supported, please use GPU")
        if device_index <= -2:
            raise ValueError("The GPU device index can be only -1, 0 and larger than 0")
        if ordsub_number is None:
            ordsub_number = 1
        if ordsub_number <= 0:
            raise ValueError("The number of ordered subsets cannot be negative or zero")
        if detectors_y is not None and detectors_y <= 0:
            raise ValueError("The size of the vertical detector cannot be negative or zero")

        self.detectors_x = int(detectors_x)
        self.detectors_x_pad = int(detectors_x_pad)
        self.detectors_y = int(detectors_y)
        self.angles_vec = np.ascontiguousarray(angles_vec, dtype=np.float64)
        self.centre_of_rotation = centre_of_rotation
        self.recon_size = int(recon_size)
        self.processing_arch = processing_arch
        self.device_index = int(device_index)
        self.ordsub_number = int(ordsub_number)

        self.nz, self.n = self.detectors_y, self.recon_size
        self._vshift = None
        self.slab = None   # tomobar_amd.slab.SlabComm when this object projects ONE z-slab of a larger volume (RecToolsIRCuPy.slab sets it)
        self.nu = self.detectors_x + 2 * self.detectors_x_pad
        self.na = self.angles_vec.size

        cor = np.asarray(centre_of_rotation, dtype=np.float64)
        if cor.ndim == 0:
            cor_arr, stride = cor.reshape(1).copy(), 0
        elif cor.ndim == 1:
            cor_arr, stride = np.ascontiguousarray(cor), 1
        elif cor.ndim == 2 and cor.shape == (self.na, 2):
            # (horizontal, vertical) per angle, supp/funcs.py:52-55.  The context takes the horizontal components; the
            # vertical ones become a per-angle resampling of the detector rows around the per-slice operators (below).
            cor_arr, stride = np.ascontiguousarray(cor[:, 0]), 1
            if np.any(cor[:, 1] != 0):
                self._vshift = np.ascontiguousarray(cor[:, 1], dtype=np.float32)
        else:
            raise ValueError("The CoR must be a scalar, a vector [angles] or an array [angles, 2]")

        self._device = ops._require_gpu(self.device_index)
        self._lib = L.lib()   # the context lives in THIS library: every later call goes to the same handle (see _lib.use_flavour)
        handle = C.c_void_p()
        self._chk(self._lib.tomo_ctx_create(
            self.device_index, self.nz, self.n, self.nu, self.na,
            self.angles_vec.ctypes.data_as(C.POINTER(C.c_double)), cor_arr.ctypes.data_as(C.POINTER(C.c_double)),
            stride, self.ordsub_number, L.FLAG_LERP8 if lerp8 else 0, C.byref(handle)))
        self._ctx = handle

        # ---- geometry dictionaries in ASTRA's vocabulary (astra_base.py:215-222,244-255,287-308)
        self.vol_geom = {"GridRowCount": self.n, "GridColCount": self.n, "GridSliceCount": self.nz,
                         "option": {"WindowMinX": -self.n / 2, "WindowMaxX": self.n / 2,
                                    "WindowMinY": -self.n / 2, "WindowMaxY": self.n / 2,
                                    "WindowMinZ": -self.nz / 2, "WindowMaxZ": self.nz / 2}}
        self.NumbProjBins = self._lib.tomo_ctx_num_bins(self._ctx)
        table = np.zeros((self.ordsub_number, self.NumbProjBins), dtype=np.int64)
        self._chk(self._lib.tomo_ctx_newind_table(self._ctx, table.ctypes.data_as(C.POINTER(C.c_int64))))
        self.newInd_Vec = table
        if self.ordsub_number == 1:
            self.proj_geom = self._proj_geom(np.arange(self.na))
        else:
            self.proj_geom_OS = {s: self._proj_geom(self.subset_indices(s)) for s in range(self.ordsub_number)}

    # ------------------------------------------------------------------ helpers
    def _proj_geom(self, idx):
        cor = self.centre_of_rotation
        cor_sel = cor if np.ndim(cor) == 0 else np.asarray(cor)[idx]
        return {"type": "parallel3d_vec", "DetectorRowCount": self.nz, "DetectorColCount": self.nu,
                "Vectors": vec_geom_init3D(self.angles_vec[idx], 1.0, 1.0, cor_sel)}

    def subset_indices(self, sub_ind: int) -> np.ndarray:
        """OS-specific angle indices after the reference's one-element trim (methodsIR_CuPy.py:454-456)."""
        ind = self.newInd_Vec[sub_ind, :]
        if ind[self.NumbProjBins - 1] == 0:
            ind = ind[:-1]
        return ind

    def subset_size(self, os_index) -> int:
        return self._lib.tomo_ctx_subset_size(self._ctx, -1 if os_index is None else int(os_index))

    def vol_shape(self):
        return (self.nz, self.n, self.n)

    def sino_shape(self, os_index=None):
        return (self.nz, self.subset_size(os_index), self.nu)

    def _sub(self, os_index):
        return -1 if (os_index is None or self.ordsub_number == 1) else int(os_index)

    def _vol_in(self, x):
        x = ops.contiguous(ops.to_device(x, self.device_index))
        ops._chk_f32(x, "volume")
        if tuple(x.shape) != self.vol_shape():
            raise ValueError(f"volume has shape {tuple(x.shape)}, expected {self.vol_shape()}")
        return x

    def _sino_in(self, b, os_index):
        b = ops.contiguous(ops.to_device(b, self.device_index))
        ops._chk_f32(b, "projection data")
        if tuple(b.shape) != self.sino_shape(os_index):
            raise ValueError(f"projection data has shape {tuple(b.shape)}, expected {self.sino_shape(os_index)}")
        return b

    # ------------------------------------------------------------------ AstraTools3D interface
    def _forwprojCuPy(self, object3D):
        return self.forward(object3D, None)

    def _forwprojOSCuPy(self, object3D, os_index: int):
        return self.forward(object3D, os_index)

    def _backprojCuPy(self, proj_data):
        return self.backward(proj_data, None)

    def _backprojOSCuPy(self, proj_data, os_index: int):
        return self.backward(proj_data, os_index)

    # ------------------------------------------------------------------ vertical CoR component
    @property
    def has_vertical_shift(self) -> bool:
        return self._vshift is not None

    def _shift_rows(self, sino, os_index, sign, out=None):
        """Resample the detector rows of a subset's projections by sign * CenterRotOffset[:, 1] per angle (2-tap linear,
        zero outside): detector row r of angle a looks at slice r + shift[a]."""
        key = self._sub(os_index)
        tabs = self.__dict__.setdefault("_vshift_tables", {})
        if key not in tabs:
            idx = np.arange(self.na) if key < 0 else self.subset_indices(key)
            tabs[key] = torch.from_numpy(np.ascontiguousarray(self._vshift[idx])).to(self._device)
        if out is None:
            out = torch.empty_like(sino)
        slab = self.slab
        if slab is not None and slab.world > 1:
            # z-slab mode (round 5): detector row r of angle a looks at slice r + shift[a], which may lie in the NEIGHBOUR's
            # slab.  g = ceil(max |shift|) + 1 ghost rows per interior side travel with one packed exchange; the resampling
            # then runs on the extended rows (zero beyond the global first / last row, like the whole-volume operator) and the
            # local rows are kept.  Same arithmetic per sample as unsharded: bit-identical.
            from .slab import check_ghost_rows, extend_detector_rows
            g = int(np.ceil(float(np.abs(self._vshift).max()))) + 1
            if getattr(self, "_vshift_checked", None) is not slab:           # once per communicator
                check_ghost_rows(slab, g, self.nz, f"a vertical CoR component of up to {float(np.abs(self._vshift).max()):.2f} rows")
                self._vshift_checked = slab
            ext, lo = extend_detector_rows(slab, sino, g)
            ext_out = torch.empty_like(ext)
            with torch.cuda.device(self._device):
                self._chk(self._lib.tomo_shift_rows(ops.ptr(ext), ops.ptr(ext_out), int(ext.shape[0]), int(sino.shape[1]), self.nu,
                                                ops.ptr(tabs[key]), float(sign), ops.stream_ptr(sino)))
            out.copy_(ext_out[lo:lo + self.nz])
            return out
        with torch.cuda.device(self._device):
            self._chk(self._lib.tomo_shift_rows(ops.ptr(sino), ops.ptr(out), self.nz, int(sino.shape[1]), self.nu,
                                            ops.ptr(tabs[key]), float(sign), ops.stream_ptr(sino)))
        return out

    def _adjoint_in(self, res, os_index):
        """What the back projector takes for ``res``: the rows resampled by -shift when there is a vertical component."""
        return res if self._vshift is None else self._shift_rows(res, os_index, -1.0)

    # ------------------------------------------------------------------ operators
    def forward(self, vol, os_index=None, out=None):
        vol = self._vol_in(vol)
        if out is None:
            out = torch.empty(self.sino_shape(os_index), dtype=torch.float32, device=self._device)
        if self._vshift is not None:
            tmp = torch.empty_like(out)
            with torch.cuda.device(self._device):
                self._chk(self._lib.tomo_fp3d(self._ctx, self._sub(os_index), ops.ptr(vol), ops.ptr(tmp), ops.stream_ptr(vol)))
            return self._shift_rows(tmp, os_index, 1.0, out)
        with torch.cuda.device(self._device):
            self._chk(self._lib.tomo_fp3d(self._ctx, self._sub(os_index), ops.ptr(vol), ops.ptr(out), ops.stream_ptr(vol)))
        return out

    def backward(self, sino, os_index=None, out=None):
        sino = self._adjoint_in(self._sino_in(sino, os_index), os_index)
        if out is None:
            out = torch.empty(self.vol_shape(), dtype=torch.float32, device=self._device)
        with torch.cuda.device(self._device):
            self._chk(self._lib.tomo_bp3d(self._ctx, self._sub(os_index), ops.ptr(sino), ops.ptr(out), ops.stream_ptr(sino)))
        return out

    # ---- private residual layout between `residual` and `grad_step` / `grad_step_momentum` / `admm_z_update`
    def set_residual_layout(self, layout: str):
        """"planar" (default: the residual is the [detY, angles, detX] array every entry point documents) or "zquad": the
        forward projector leaves the four slices of a quad interleaved, which the back projector stages with one 16-byte
        load instead of four gathers (include/tomo_mi355x.h, TOMO_RESIDUAL_ZQUAD).  Same values, bit for bit; only the
        fused calls above follow it, and only buffers from ``residual_buffer`` may be handed between them.  The drivers
        switch it on around their loops and back off before they return."""
        if layout == "zquad" and self._vshift is not None:
            raise ValueError("the quad-interleaved residual is not available with a vertical CoR component")
        self._chk(self._lib.tomo_ctx_set_residual_layout(self._ctx, L.RESIDUAL_LAYOUT[layout]))

    def residual_layout(self) -> str:
        code = self._lib.tomo_ctx_residual_layout(self._ctx)
        return {v: k for k, v in L.RESIDUAL_LAYOUT.items()}[code]

    def residual_buffer(self, os_index=None) -> torch.Tensor:
        """Uninitialised buffer for the residual of a subset in the context's CURRENT residual layout."""
        if self.residual_layout() == "planar":
            return torch.empty(self.sino_shape(os_index), dtype=torch.float32, device=self._device)
        n = int(self._lib.tomo_ctx_residual_elems(self._ctx, self._sub(os_index)))
        return torch.empty((n // (4 * self.nu * self.subset_size(os_index)), self.subset_size(os_index), self.nu, 4),
                           dtype=torch.float32, device=self._device)

    def _check_residual(self, res, os_index):
        """The C-ABI cannot see buffer sizes; the context's residual layout is state another thread or stream sharing this
        object may have flipped (a driver holds "zquad" around its loop): a buffer of the other layout would be written
        past its end when nz % 4 != 0, or misread.  Refuse anything but a buffer of the current layout's size and rank."""
        want = int(self._lib.tomo_ctx_residual_elems(self._ctx, self._sub(os_index)))
        quad = self.residual_layout() == "zquad"
        if res.numel() != want or (res.dim() == 4) != quad:
            raise ValueError(f"residual buffer of {res.numel()} floats / rank {res.dim()} does not match the context's current "
                             f"'{self.residual_layout()}' layout ({want} floats): take it from residual_buffer() after "
                             "set_residual_layout(), and do not share one projector object between drivers running concurrently")

    def residual_as_planar(self, res, os_index=None) -> torch.Tensor:
        """A [detY, angles, detX] copy of a residual buffer whatever layout it was written in (diagnostics / tests)."""
        if res.dim() == 3:
            return res
        return res.permute(0, 3, 1, 2).reshape(-1, res.shape[1], res.shape[2])[: self.nz].contiguous()

    # fused forms used by the FISTA / ADMM drivers (buffers validated by the drivers)
    def residual(self, vol, b, w, fidelity: str, os_index, out, gathered: int = 0, robust=None):
        """out = w_s*(A_s vol - b_s) (LS/PWLS) or 1 - b_s/max(A_s vol, 1e-8) (KL); ``gathered`` bit0/bit1: b / w is
        already the subset's array instead of the full sinogram.  ``robust`` = ("huber" | "studentst", threshold):
        the Huber / Student's-t re-weighting of the residual in the same epilogue (include/tomo_mi355x.h)."""
        if self._vshift is None:
            self._check_residual(out, os_index)
        if robust is not None:
            mode, delta = robust
            if self._vshift is not None:
                self.residual(vol, b, w, fidelity, os_index, out, gathered)
                return self.robust_apply(out, mode, delta)
            with torch.cuda.device(self._device):
                self._chk(self._lib.tomo_fp3d_residual_robust(self._ctx, self._sub(os_index), ops.ptr(vol), ops.ptr(b),
                                                          ops.ptr(w), int(gathered), L.FID[fidelity], L.ROBUST[mode],
                                                          float(delta), ops.ptr(out), ops.stream_ptr(vol)))
            return out
        if self._vshift is not None:
            ax = self.forward(vol, os_index)
            src = self._src_table(os_index)
            with torch.cuda.device(self._device):
                self._chk(self._lib.tomo_sino_residual(ops.ptr(ax), ops.ptr(b), ops.ptr(w), ops.ptr(src), self.nz,
                                                   int(src.numel()), self.na, self.nu, int(gathered), L.FID[fidelity],
                                                   ops.ptr(out),
                                                   ops.stream_ptr(vol)))
            return out
        with torch.cuda.device(self._device):
            self._chk(self._lib.tomo_fp3d_residual(self._ctx, self._sub(os_index), ops.ptr(vol), ops.ptr(b),
                                               ops.ptr(w), int(gathered), L.FID[fidelity], ops.ptr(out),
                                               ops.stream_ptr(vol)))
        return out

    # ---- ring-artefact data terms (Group-Huber offsets / stripe-weighted least squares); see include/tomo_mi355x.h
    def _src_table(self, os_index):
        """int32 device table of the subset's indices into the full angle axis (cached)."""
        key = self._sub(os_index)
        tabs = self.__dict__.setdefault("_src_tables", {})
        if key not in tabs:
            idx = np.arange(self.na) if key < 0 else self.subset_indices(key)
            tabs[key] = torch.from_numpy(np.ascontiguousarray(idx, dtype=np.int32)).to(self._device)
        return tabs[key]

    def residual_ring(self, vol, b, r_x, accelerate, os_index, out):
        """out = (A_s vol - b_s) + accelerate * r_x[z, u]  (LS residual with the Group-Huber offsets added)."""
        if self._vshift is not None:   # the row resampling sits between projector and residual: unfused, same roundings
            self.residual(vol, b, None, "LS", os_index, out)
            with torch.cuda.device(self._device):
                self._chk(self._lib.tomo_sino_add_ring(ops.ptr(out), ops.ptr(r_x), float(accelerate), self.nz,
                                                   self.subset_size(os_index), self.nu, ops.stream_ptr(vol)))
            return out
        with torch.cuda.device(self._device):
            self._chk(self._lib.tomo_fp3d_residual_ring(self._ctx, self._sub(os_index), ops.ptr(vol), ops.ptr(b), ops.ptr(r_x),
                                                    float(accelerate), ops.ptr(out), ops.stream_ptr(vol)))
        return out

    def ring_reduce(self, res, w, r_x, l_inv, os_index, r_out):
        """r_out = r_x - l_inv * sum_angles(res); afterwards res *= w_s in place when PWLS weights are given."""
        src = self._src_table(os_index)
        with torch.cuda.device(self._device):
            self._chk(self._lib.tomo_ring_gh_reduce(ops.ptr(res), ops.ptr(w), ops.ptr(src), self.nz, int(src.numel()), self.na,
                                                self.nu, ops.ptr(r_x), float(l_inv), ops.ptr(r_out), ops.stream_ptr(res)))

    def swls_apply(self, res, w, beta, os_index):
        src = self._src_table(os_index)
        with torch.cuda.device(self._device):
            self._chk(self._lib.tomo_swls_apply(ops.ptr(res), ops.ptr(w), ops.ptr(src), self.nz, int(src.numel()), self.na,
                                            self.nu, float(beta), ops.stream_ptr(res)))

    def robust_apply(self, res, mode: str, delta):
        """res <- Huber / Student's-t re-weighting of res, in place (any residual layout)."""
        with torch.cuda.device(self._device):
            self._chk(self._lib.tomo_sino_robust(ops.ptr(res), res.numel(), L.ROBUST[mode], float(delta), ops.stream_ptr(res)))
        return res

    def ring_update(self, r, r_old, r_x, lam, beta):
        with torch.cuda.device(self._device):
            self._chk(self._lib.tomo_ring_gh_update(ops.ptr(r), ops.ptr(r_old), ops.ptr(r_x), float(lam), float(beta),
                                                r.numel(), ops.stream_ptr(r)))

    def momentum(self, x, x_old, x_t, beta):
        """x_t = x + beta (x - x_old), leaving the in-plane transposed x_t in the context for the next forward
        projection of x_t (which then skips its transpose pass)."""
        with torch.cuda.device(self._device):
            self._chk(self._lib.tomo_momentum_transposed(self._ctx, ops.ptr(x), ops.ptr(x_old), ops.ptr(x_t), float(beta),
                                                     ops.stream_ptr(x)))

    def invalidate(self):
        """Drop the one-shot transposed copy ``momentum`` may have left in the context (see tomo_ctx_invalidate)."""
        self._chk(self._lib.tomo_ctx_invalidate(self._ctx))

    def grad_step(self, res, x_t, x_out, l_inv, nonneg, os_index):
        if self._vshift is None:
            self._check_residual(res, os_index)
        res = self._adjoint_in(res, os_index)
        with torch.cuda.device(self._device):
            self._chk(self._lib.tomo_bp3d_fista(self._ctx, self._sub(os_index), ops.ptr(res), ops.ptr(x_t), ops.ptr(x_out),
                                            float(l_inv), int(bool(nonneg)), ops.stream_ptr(x_t)))

    def grad_step_momentum(self, res, x_t, x_old_then_x, l_inv, beta, nonneg, os_index):
        if self._vshift is None:
            self._check_residual(res, os_index)
        res = self._adjoint_in(res, os_index)
        with torch.cuda.device(self._device):
            self._chk(self._lib.tomo_bp3d_fista_momentum(self._ctx, self._sub(os_index), ops.ptr(res), ops.ptr(x_t),
                                                     ops.ptr(x_old_then_x), float(l_inv), float(beta),
                                                     int(bool(nonneg)), ops.stream_ptr(x_t)))

    def admm_z_update(self, res, z, x, u, zu_out, tau, rho, relax_on, one_minus_alpha, alpha, nonneg, os_index):
        if self._vshift is None:
            self._check_residual(res, os_index)
        res = self._adjoint_in(res, os_index)
        with torch.cuda.device(self._device):
            self._chk(self._lib.tomo_bp3d_admm(self._ctx, self._sub(os_index), ops.ptr(res), ops.ptr(z), ops.ptr(x),
                                           ops.ptr(u), ops.ptr(zu_out), float(tau), float(rho), int(bool(relax_on)),
                                           float(one_minus_alpha), float(alpha), int(bool(nonneg)), ops.stream_ptr(z)))

    def angle_table(self, os_index=None):
        n = self.subset_size(os_index)
        tab = (L.AngleRecord * max(n, 1))()
        self._chk(self._lib.tomo_ctx_angle_table(self._ctx, self._sub(os_index), tab, max(n, 1)))
        return tab, n

    def kernel_path(self, op: str = "fp") -> str:
        """Which kernel form the last forward ("fp") / back ("bp") projection of this object took (diagnostics)."""
        return self._lib.tomo_ctx_kernel_path(self._ctx, op.encode()).decode()

    def _chk(self, rc):
        L.check(rc, self._lib)

    def release_scratch(self):
        self._chk(self._lib.tomo_ctx_release_scratch(self._ctx))

    def __del__(self):
        ctx = getattr(self, "_ctx", None)
        if ctx:
            try:
                self._lib.tomo_ctx_destroy(ctx)
            except Exception:
                pass
            self._ctx = None


# name the reference's code imports
AstraTools3D = HipTools3D
