"""Iterative reconstruction on MI355X behind the reference's ``RecToolsIRCuPy`` surface
(``tomobar/methodsIR_CuPy.py:36-667``): same constructor, methods, dictionaries and return shapes; arrays are
float32 ``torch.Tensor`` on ``cuda:<device_projector>`` where the reference uses ``cupy.ndarray``.

What differs is how a sub-iteration is executed.  The reference issues ~10 CuPy launches with temporaries plus two
ASTRA calls that rebuild a projector object each time (``methodsIR_CuPy.py:447-475``, ``astra_base.py:538-604``).
Here a FISTA-OS sub-iteration is

    residual  = forward-projection kernel with the (w *)(Ax - b) / KL epilogue, subset gathered by angle index
    gradient  = back-projection kernel whose epilogue applies  X = P+(X_t - g/L)  (and, when no proximal
                operator sits in between, the momentum update as well)
    prox      = ``tomo_pdtv`` / ``tomo_roftv``: one fused kernel per inner iteration
    momentum  = one streaming kernel

on persistent buffers, all asynchronous on the current stream.  Scalar bookkeeping (t-sequence, step sizes) is done
in float32 exactly where the reference's NumPy-2 semantics make it float32 (``methodsIR_CuPy.py:438-475,523-554``).
"""

from __future__ import annotations

from typing import Optional, Union

import numpy as np
import torch
from numpy import float32

from . import ops
from .projector import HipTools3D, geom_size
from .regularisersCuPy import prox_regul, reserve_prox_scratch
from .supp.dicts import dicts_check
from .supp.suppTools import _apply_horiz_detector_padding, check_kwargs, perform_recon_crop


class RecToolsIRCuPy:
    """Iterative reconstruction algorithms (FISTA, ADMM, OSEM, SIRT, CGLS, Landweber, power method).

    Args:
        DetectorsDimH (int): Horizontal detector dimension size.
        DetectorsDimH_pad (int): The amount of padding for the horizontal detector.
        DetectorsDimV (int, None): Vertical detector dimension size, 'None' for 2D or an integer for 3D.
        CenterRotOffset (float, np.ndarray): The Centre of Rotation (CoR) scalar or a vector for each angle.
        AnglesVec (np.ndarray): Vector of projection angles in radians.
        ObjSize (int): The size of the reconstructed object (a slice) defined as [recon_size, recon_size].
        device_projector (int, optional): GPU index. Defaults to 0.
        OS_number (int, optional): The number of ordered subsets, None for non-OS reconstruction.
    """

    def __init__(self, DetectorsDimH: int, DetectorsDimH_pad: int, DetectorsDimV: Union[int, None],
                 CenterRotOffset: Union[float, np.ndarray], AnglesVec: np.ndarray, ObjSize: int,
                 device_projector: int = 0, OS_number: Optional[int] = None):
        self.OS_number = OS_number
        # padding the detector enlarges the reconstruction grid; the result is cropped back (methodsIR_CuPy.py:72-79)
        self.objsize_user_given = ObjSize if DetectorsDimH_pad != 0 else None
        if DetectorsDimH_pad > 0:
            ObjSize = DetectorsDimH + 2 * DetectorsDimH_pad
        if DetectorsDimV == 0 or DetectorsDimV is None:
            DetectorsDimV = 1  # 2D is one slice of the 3D geometry (:81-82)
        self.geom = "3D"
        self.Atools = HipTools3D(DetectorsDimH, DetectorsDimH_pad, DetectorsDimV, AnglesVec, CenterRotOffset,
                                 ObjSize, "gpu", device_projector, OS_number)
        self.power_seed = None  # set to an int for a reproducible power-method start vector
        self.slab = None        # tomobar_amd.slab.SlabComm when this object reconstructs one z-slab of a larger volume

    @property
    def slab(self):
        return self._slab

    @slab.setter
    def slab(self, comm):
        self._slab = comm
        self.Atools.slab = comm   # the projector exchanges ghost detector rows itself when a vertical CoR component is set

    @property
    def OS_number(self) -> int:
        return self._OS_number

    @OS_number.setter
    def OS_number(self, value):
        self._OS_number = 1 if value is None else value

    @property
    def objsize_user_given(self):
        return self._objsize_user_given

    @objsize_user_given.setter
    def objsize_user_given(self, value):
        self._objsize_user_given = value

    def reserve_scratch(self, _regularisation_: Union[dict, None]) -> None:
        """Optional set-up call (no reference counterpart): allocate and PLACE the TV operators' scratch arena for this
        geometry now.  ``FISTA`` / ``ADMM`` / ``OSEM`` do it themselves at the start of their first call; calling it right
        after the constructor -- before the projection data are moved to the GPU -- lets the placement search see an
        empty device, where it ends on a block of the fast class in 5 of 5 processes instead of 2 of 5
        (docs/kernels/placement.md, profiles/r5f_bench_reserve_order_ab.txt).  A later call is a no-op."""
        if _regularisation_ is not None and _regularisation_.get("method") is not None:
            reserve_prox_scratch(self, self.Atools.vol_shape(), _regularisation_)

    # ------------------------------------------------------------------ operators
    def _Ax(self, x, sub_ind: int = 1, os: bool = False):
        return self.Atools._forwprojOSCuPy(x, os_index=sub_ind) if os else self.Atools._forwprojCuPy(x)

    def _Atb(self, b, sub_ind: int = 1, os: bool = False):
        return self.Atools._backprojOSCuPy(b, os_index=sub_ind) if os else self.Atools._backprojCuPy(b)

    # ------------------------------------------------------------------ shared set-up / tear-down
    def _gdot(self, x, y) -> float:
        """Inner product over the whole volume / sinogram (sum-all-reduced over the z-slabs when sharded)."""
        v = ops.dot(x, y)
        return v if self.slab is None else self.slab.allreduce_sum(v)

    def _new_vol(self, fill=None):
        v = torch.empty(self.Atools.vol_shape(), dtype=torch.float32, device=self.Atools._device)
        if fill is not None:
            ops.fill(v, fill)
        return v

    def _prepare_data(self, _data_, _algorithm_, _regularisation_, method_run):
        self._given = _data_.get("projection_data") if isinstance(_data_, dict) else None  # decides the array library of the result
        d, a, r = dicts_check(self, _data_, _algorithm_, _regularisation_, method_run=method_run)
        d["projection_data"] = _apply_horiz_detector_padding(d["projection_data"], self.Atools.detectors_x_pad, True)
        expect = self.Atools.sino_shape(None)
        if tuple(d["projection_data"].shape) != expect:
            raise ValueError(f"projection data has shape {tuple(d['projection_data'].shape)} after axis swap and "
                             f"padding, the geometry expects {expect}")
        if d["projection_data"].dtype != torch.float32:
            raise ValueError("projection data must be float32")
        return d, a, r

    def _finalise(self, x, _algorithm_):
        given = getattr(self, "_given", None)
        self._given = None
        if self.objsize_user_given is not None:
            return ops.like(perform_recon_crop(x, self.objsize_user_given), given)  # cropped result is not masked (:477-478)
        return ops.like(check_kwargs(x, cupyrun=True, recon_mask_radius=_algorithm_["recon_mask_radius"]), given)

    def __common_initialisation(self, _data_, _algorithm_, _regularisation_, method_run):
        d, a, r = self._prepare_data(_data_, _algorithm_, _regularisation_, method_run)
        if a.get("lipschitz_const") is None:
            a["lipschitz_const"] = self.powermethod(d)
        rec_dim = geom_size(self.Atools.vol_geom)
        x0 = None
        if a["initialise"] is not None:
            if tuple(a["initialise"].shape) == rec_dim:
                x0 = ops.contiguous(ops.to_device(a["initialise"], self.Atools.device_index)).clone()
            else:
                print(f"Provided initialisation (array) has incorrect dimensions, the correct dims are {rec_dim}. "
                      "Zero initialisation is used.")
        if x0 is None:
            x0 = self._new_vol(1.0 if method_run == "OSEM" else 0.0)
        use_os = self.OS_number > 1
        w = ops.pwls_weights(d["projection_data"], self.slab) if _data_["data_fidelity"] in ["PWLS", "SWLS"] else None
        # the TV operators' scratch arena is allocated and placed here, at set-up, not inside the first proximal step
        reserve_prox_scratch(self, rec_dim, r)
        return (d, a, r, x0, w, use_os)

    # ------------------------------------------------------------------ power method
    def powermethod(self, _data_: dict) -> float:
        """Largest eigenvalue of A^T A (of subset 0's operator when OS_number > 1) by 15 power iterations
        (reference: methodsIR_CuPy.py:311-354)."""
        if _data_.get("data_fidelity") is None:
            _data_["data_fidelity"] = "LS"
        A = self.Atools
        gen = None
        if self.power_seed is not None:
            gen = torch.Generator(device=A._device)
            gen.manual_seed(int(self.power_seed))
        x1 = torch.randn(A.vol_shape(), dtype=torch.float32, device=A._device, generator=gen)
        sub = 0 if self.OS_number > 1 else None
        y = A.forward(x1, sub)  # PWLS weights are all-ones here (:331-333,:344-346): no effect
        s = 1.0
        for _ in range(15):
            A.backward(y, sub, out=x1)
            s = ops.norm2(x1)
            if self.slab is not None:  # the eigenvector spans all slabs: global 2-norm
                s = float(np.sqrt(self.slab.allreduce_sum(s * s)))
            s = float32(s)
            ops.scale(float32(1.0) / s, x1, x1)
            A.forward(x1, sub, out=y)
        return float(s)

    # ------------------------------------------------------------------ FISTA
    def FISTA(self, _data_: dict, _algorithm_: Union[dict, None] = None,
              _regularisation_: Union[dict, None] = None) -> torch.Tensor:
        """Fast Iterative Shrinkage-Thresholding Algorithm with optional ordered subsets and ROF_TV / PD_TV proximal
        regularisation (reference: methodsIR_CuPy.py:401-484).  Returns the float32 volume ``[detY, N, N]``."""
        (d, a, r, x0, w, use_os) = self.__common_initialisation(_data_, _algorithm_, _regularisation_, "FISTA")
        A = self.Atools
        L_inv = float32(1.0 / a["lipschitz_const"])
        b = d["projection_data"]
        fid = self.data_fidelity
        nonneg = bool(a["nonnegativity"])
        has_prox = r["method"] is not None

        # ring-artefact data terms (BASELINE configs[4]; not in this reference version, see supp/dicts.py here):
        # Group-Huber offsets r [detY, detX] (one per detector pixel, constant over the angles) and / or SWLS weighting
        ring_lambda = d.get("ringGH_lambda")
        use_ring = ring_lambda is not None
        use_swls = fid == "SWLS"
        # Huber / Student's-t data terms: the (weighted) residual is re-weighted before the back projection
        robust = None
        if d.get("huber_threshold") is not None:
            robust = ("huber", float32(d["huber_threshold"]))
        elif d.get("studentst_threshold") is not None:
            robust = ("studentst", float32(d["studentst_threshold"]))
        if use_ring:
            ring_acc = float32(d["ringGH_accelerate"])
            r_shape = (A.nz, A.nu)
            r_cur = torch.zeros(r_shape, dtype=torch.float32, device=A._device)
            r_old = torch.zeros(r_shape, dtype=torch.float32, device=A._device)
            r_x = torch.zeros(r_shape, dtype=torch.float32, device=A._device)

        t = float32(1.0)
        X = x0                      # doubles as X_old at the start of every sub-iteration
        X_t = x0.clone()
        res = {}
        X_grad = self._new_vol() if has_prox else None
        X_prox = self._new_vol() if has_prox else None

        # the transposed X_t that A.momentum leaves in the projector context is a one-shot token for the NEXT residual of
        # this loop; nothing of it may survive this call (a later call's X_t can land at the same address)
        A.invalidate()
        # on the plain LS / PWLS / KL path nobody but the back projector reads the residual: producer and consumer use the
        # quad-interleaved layout (HipTools3D.set_residual_layout); the ring terms and the vertical CoR resampling read it
        # as [detY, angles, detX] and keep the planar one
        zquad = not (use_ring or use_swls) and not getattr(A, "has_vertical_shift", False)
        A.set_residual_layout("zquad" if zquad else "planar")
        n_sub_total = a["iterations"] * self.OS_number
        try:
            for it_no in range(a["iterations"]):
                for sub_ind in range(self.OS_number):
                    sub = sub_ind if use_os else None
                    t_old = t
                    if sub not in res:
                        res[sub] = A.residual_buffer(sub)
                    if use_ring:
                        # res = (A_s X_t - b_s) + accelerate * r_x ;  r = r_x - (1/L) sum_angles res ;  then the PWLS weights
                        A.residual_ring(X_t, b, r_x, ring_acc, sub, res[sub])
                        A.ring_reduce(res[sub], w if fid == "PWLS" else None, r_x, L_inv, sub, r_cur)
                        if robust is not None:
                            A.robust_apply(res[sub], *robust)
                    elif use_swls:
                        A.residual(X_t, b, None, "LS", sub, res[sub])
                        A.swls_apply(res[sub], w, float32(d["beta_SWLS"]), sub)
                        if robust is not None:
                            A.robust_apply(res[sub], *robust)
                    else:
                        A.residual(X_t, b, w, fid, sub, res[sub], robust=robust)
                    t = float32((float32(1.0) + np.sqrt(float32(1.0) + float32(4.0) * t * t)) * float32(0.5))
                    beta = float32((t_old - float32(1.0)) / t)
                    if not has_prox:
                        # X <- P+(X_t - grad/L) and X_t <- X + beta (X - X_old) inside the back-projection epilogue
                        A.grad_step_momentum(res[sub], X_t, X, L_inv, beta, nonneg, sub)
                    else:
                        A.grad_step(res[sub], X_t, X_grad, L_inv, nonneg, sub)
                        prox_regul(self, X_grad, r, out=X_prox)
                        if it_no * self.OS_number + sub_ind + 1 < n_sub_total:
                            # also leaves X_t transposed for the next forward projection; after the LAST sub-iteration X_t
                            # is never read again (the reference still computes it, methodsIR_CuPy.py:475): skipped
                            A.momentum(X_prox, X, X_t, beta)
                        X, X_prox = X_prox, X
                    if use_ring:
                        # r <- soft(r, lambda) ;  r_x = r + beta (r - r_old)
                        A.ring_update(r_cur, r_old, r_x, float32(ring_lambda), beta)
        finally:
            A.set_residual_layout("planar")
            A.invalidate()
        return self._finalise(X, a)

    # ------------------------------------------------------------------ ADMM
    def ADMM(self, _data_: dict, _algorithm_: Union[dict, None] = None,
             _regularisation_: Union[dict, None] = None) -> torch.Tensor:
        """Linearised, relaxed ADMM with optional ordered subsets (reference: methodsIR_CuPy.py:486-585).

        Divergence from the reference, on purpose: ``regul_param / ADMM_rho_const`` is applied to a private copy of
        the regularisation dictionary (the reference rewrites the caller's dictionary on every call, :526-528)."""
        (d, a, r, x0, w, use_os) = self.__common_initialisation(_data_, _algorithm_, _regularisation_, "ADMM")
        A = self.Atools
        b = d["projection_data"]
        fid = self.data_fidelity
        nonneg = bool(a["nonnegativity"])
        has_prox = r["method"] is not None
        rho, alpha = a["ADMM_rho_const"], a["ADMM_relax_par"]
        tau = float32(0.9 / (a["lipschitz_const"] + rho))
        r_local = dict(r)
        if has_prox:
            r_local["regul_param"] = r["regul_param"] / rho

        x = x0
        z = x0.clone()
        u = self._new_vol(0.0)
        zu = self._new_vol()
        res = {}
        # the residual goes from the forward projector straight into the fused z-update: quad-interleaved (see FISTA)
        A.set_residual_layout("planar" if getattr(A, "has_vertical_shift", False) else "zquad")
        try:
            for iter_no in range(a["iterations"]):
                for sub_ind in range(self.OS_number):
                    sub = sub_ind if use_os else None
                    if sub not in res:
                        res[sub] = A.residual_buffer(sub)
                    A.residual(z, b, w, fid, sub, res[sub])
                    # z-update, projection, over-relaxation (from the third outer iteration on) and zu = z + u
                    A.admm_z_update(res[sub], z, x, u, zu, tau, float32(rho), iter_no > 1, float32(1.0 - alpha),
                                    float32(alpha), nonneg, sub)
                    if has_prox:
                        prox_regul(self, zu, r_local, out=x)
                    else:
                        x, zu = zu, x
                ops.admm_dual(u, z, x)  # once per outer iteration (:566)
                if a["verbose"] and np.mod(iter_no, (round)(a["iterations"] / 5) + 1) == 0:
                    print("ADMM iteration (", iter_no + 1, ") using", r["method"], "regularisation")
        finally:
            A.set_residual_layout("planar")
        return self._finalise(x, a)

    # ------------------------------------------------------------------ simple iterative methods
    def Landweber(self, _data_: dict, _algorithm_: Union[dict, None] = None) -> torch.Tensor:
        """x <- x - tau A^T(Ax - b)   (reference: methodsIR_CuPy.py:128-172)."""
        d, a, _ = self._prepare_data(_data_, _algorithm_, None, "Landweber")
        A = self.Atools
        b = d["projection_data"]
        x = self._new_vol(0.0)
        step = float32(a["tau_step_lanweber"])
        # the residual goes straight from the forward projector into the fused gradient step: quad-interleaved (see FISTA)
        A.set_residual_layout("planar" if getattr(A, "has_vertical_shift", False) else "zquad")
        try:
            res = A.residual_buffer(None)
            for _ in range(a["iterations"]):
                A.residual(x, b, None, "LS", None, res)
                # x - tau*g with the clamp as a separate rounding step, like the reference's in-place ops
                A.grad_step(res, x, x, step, a["nonnegativity"], None)
        finally:
            A.set_residual_layout("planar")
        return self._finalise(x, a)

    def SIRT(self, _data_: dict, _algorithm_: Union[dict, None] = None) -> torch.Tensor:
        """x <- x + C A^T (R (b - Ax))   (reference: methodsIR_CuPy.py:174-231)."""
        d, a, _ = self._prepare_data(_data_, _algorithm_, None, "SIRT")
        A = self.Atools
        b = d["projection_data"]
        ones_v = self._new_vol(1.0)
        R = A.forward(ones_v)
        ops.recip_safe(R, R)
        ones_s = torch.empty_like(b)
        ops.fill(ones_s, 1.0)
        Cm = A.backward(ones_s)
        ops.recip_safe(Cm, Cm)
        del ones_s
        x = ones_v
        res = torch.empty_like(b)
        upd = self._new_vol()
        for _ in range(a["iterations"]):
            A.forward(x, None, out=res)
            ops.axpby(1.0, b, -1.0, res)   # b - Ax
            ops.mul(R, res)
            A.backward(res, None, out=upd)
            ops.mul(Cm, upd)
            ops.axpby(1.0, upd, 1.0, x)
            if a["nonnegativity"]:
                ops.clamp_min(x, 0.0)
        return self._finalise(x, a)

    def CGLS(self, _data_: dict, _algorithm_: Union[dict, None] = None) -> torch.Tensor:
        """Conjugate-gradient least squares (reference: methodsIR_CuPy.py:233-309)."""
        d, a, _ = self._prepare_data(_data_, _algorithm_, None, "CGLS")
        A = self.Atools
        x = self._new_vol(0.0)
        r_vec = d["projection_data"].clone()
        dvec = A.backward(r_vec)
        normr2 = float32(self._gdot(dvec, dvec))
        Ad = torch.empty_like(r_vec)
        s = self._new_vol()
        for _ in range(a["iterations"]):
            A.forward(dvec, None, out=Ad)
            alpha = float32(normr2 / float32(self._gdot(Ad, Ad)))
            ops.axpby(alpha, dvec, 1.0, x)
            ops.axpby(-alpha, Ad, 1.0, r_vec)
            A.backward(r_vec, None, out=s)
            normr2_new = float32(self._gdot(s, s))
            beta = float32(normr2_new / normr2)
            normr2 = normr2_new
            ops.axpby(1.0, s, beta, dvec)  # d = s + beta d
            if a["nonnegativity"]:
                ops.clamp_min(x, 0.0)
        return self._finalise(x, a)

    def OSEM(self, _data_: dict, _algorithm_: Union[dict, None] = None,
             _regularisation_: Union[dict, None] = None) -> torch.Tensor:
        """OSEM / MLEM for emission data as the reference writes it (methodsIR_CuPy.py:587-667): the multiplicative
        update uses ``backproj * normalisation`` with ``normalisation = clip(A_0^T 1, 1e-8)`` (:654; the textbook
        form divides -- kept as is so results match the reference)."""
        (d, a, r, x, w, use_os) = self.__common_initialisation(_data_, _algorithm_, _regularisation_, "OSEM")
        A = self.Atools
        b = d["projection_data"]
        eps = 1e-8
        sub0 = 0 if use_os else None
        ones_s = torch.empty(A.sino_shape(sub0), dtype=torch.float32, device=A._device)
        ops.fill(ones_s, 1.0)
        normalisation = A.backward(ones_s, sub0)
        ops.clamp_min(normalisation, eps)
        del ones_s
        ratio, back = {}, self._new_vol()
        for _ in range(a["iterations"]):
            for sub_ind in range(self.OS_number):
                sub = sub_ind if use_os else None
                if sub not in ratio:
                    ratio[sub] = torch.empty(A.sino_shape(sub), dtype=torch.float32, device=A._device)
                # ratio = b_s / clip(A_s x, eps) as the forward projector's epilogue
                A.residual(x, b, None, "RATIO", sub, ratio[sub])
                A.backward(ratio[sub], sub, out=back)
                ops.mul(normalisation, back)
                ops.mul(back, x)
                if r["method"] is not None:
                    x = prox_regul(self, x, r)
        return self._finalise(x, a)
