"""TV proximal operators on MI355X: same functions and argument meaning as the reference's
``tomobar/regularisersCuPy.py`` (``prox_regul`` :6-38, ``ROF_TV_cupy`` :41-167, ``PD_TV_cupy`` :170-296),
arrays are float32 ``torch.Tensor`` on the GPU instead of ``cupy.ndarray``.

The iteration loops run inside ``libtomo_mi355x.so`` (``tomo_pdtv`` / ``tomo_roftv``): one fused HIP kernel per
iteration, launched back to back on the caller's stream, scratch taken from the library's arena
(the reference allocates nine arrays and looks the CUDA module up on every call, :84,220-232).
"""

from __future__ import annotations

from typing import Tuple

import numpy as np
import torch

from . import ops


def prox_regul(self, X: torch.Tensor, _regularisation_: dict, out=None) -> torch.Tensor:
    """Dispatch on the ``method`` substring exactly like regularisersCuPy.py:16-38.

    One key beyond the reference's: ``_regularisation_["exact_roundings"] = True`` runs PD_TV with the rounding sequence of
    the reference's kernels for float32 duals too (``tomo_set_variant("pdtv", 22)``: bit-identical to the reference
    arithmetic, 5-16 % slower per launch) for this call; absent / False = the default (within 1e-5)."""
    if _regularisation_.get("exact_roundings") and "PD_TV" in _regularisation_["method"] and ops.get_variant("pdtv") == 0:
        with ops.variant("pdtv", 22):
            return _prox_regul(self, X, _regularisation_, out)
    return _prox_regul(self, X, _regularisation_, out)


def _prox_regul(self, X: torch.Tensor, _regularisation_: dict, out=None) -> torch.Tensor:
    method = _regularisation_["method"]
    slab = getattr(self, "slab", None)
    if slab is not None and X.dim() == 3 and min(X.shape) > 1:
        # the volume is one z-slab of a larger one: 3D TV with ghost planes exchanged between z-neighbours
        from .slab import pd_tv_slab, rof_tv_slab
        X = ops.contiguous(X)
        if "ROF_TV" in method:
            return rof_tv_slab(X, slab, _regularisation_["regul_param"], _regularisation_["iterations"],
                               _regularisation_["time_marching_step"], _regularisation_.get("half_precision", False),
                               out=out)
        if "PD_TV" in method:
            return pd_tv_slab(X, slab, _regularisation_["regul_param"], _regularisation_["iterations"],
                              _regularisation_["methodTV"], self.nonneg_regul, _regularisation_["PD_LipschitzConstant"],
                              _regularisation_.get("half_precision", False), out=out)
    if "ROF_TV" in method:
        return ROF_TV_cupy(X, _regularisation_["regul_param"], _regularisation_["iterations"],
                           _regularisation_["time_marching_step"], self.Atools.device_index,
                           _regularisation_.get("half_precision", False), out=out)
    if "PD_TV" in method:
        return PD_TV_cupy(X, _regularisation_["regul_param"], _regularisation_["iterations"],
                          _regularisation_["methodTV"], self.nonneg_regul, _regularisation_["PD_LipschitzConstant"],
                          self.Atools.device_index, _regularisation_.get("half_precision", False), out=out)
    raise ValueError(f"unknown regularisation method {method!r}: ROF_TV and PD_TV are supported")


def reserve_prox_scratch(self, vol_shape, _regularisation_: dict) -> None:
    """Set-up step of the iterative drivers: allocate -- and place, tomo_reserve_scratch -- the library's TV scratch arena
    for volumes of ``vol_shape`` before the loop starts, so that the placement search (0.1-4 s, transient footprint of up
    to `tries` x arena) is not part of the first proximal step.  No reference counterpart (CuPy's pool allocates the nine
    arrays inside every call, regularisersCuPy.py:220-232).  Nothing to do without a TV method, and in z-slab mode the
    slab drivers take their own placed block (slab.py)."""
    method = _regularisation_.get("method")
    if method is None or getattr(self, "slab", None) is not None:
        return
    shape = tuple(int(v) for v in vol_shape)
    if len(shape) == 3 and 1 in shape:       # a singleton axis runs the 2D kernels (_check_if_input_2d_or_3d)
        i = shape.index(1)
        shape = shape[:i] + shape[i + 1:]
    kind = "ROF_TV" if "ROF_TV" in method else ("PD_TV" if "PD_TV" in method else None)
    if kind is None:
        return
    ops.reserve_tv_scratch(shape, f"cuda:{self.Atools.device_index}", kind, bool(_regularisation_.get("half_precision", False)))


def _prepare(data, gpu_id: int):
    if gpu_id < 0:
        raise ValueError("The gpu_device must be a positive integer or zero")
    data = ops.to_device(data, gpu_id)
    if data.dtype != torch.float32:
        raise ValueError("The input data should be float32 data type")
    data, is2d, axis = _check_if_input_2d_or_3d(data)
    return ops.contiguous(data), is2d, axis


def _finish(result, is2d, axis, orig_shape, out, given=None):
    result = result.unsqueeze(axis) if is2d else result
    return ops.like(result, given) if out is None else out.view(orig_shape)


def ROF_TV_cupy(data, regularisation_parameter: float = 1e-05, iterations: int = 3000,
                time_marching_parameter: float = 0.001, gpu_id: int = 0, half_precision: bool = False,
                out=None) -> torch.Tensor:
    """Rudin-Osher-Fatemi TV by explicit time marching (reference: regularisersCuPy.py:41-167).

    ``half_precision`` reproduces the reference's binary16 storage of the D fields (they are rounded through
    half in registers; the fused kernel never writes them to memory)."""
    orig_shape = tuple(data.shape)
    d, is2d, axis = _prepare(data, gpu_id)
    res = torch.empty_like(d) if out is None else out.view(d.shape)
    ops.roftv(d, res, np.float32(regularisation_parameter), np.float32(time_marching_parameter), iterations,
              half_precision)
    return _finish(res, is2d, axis, orig_shape, out, data)


def PD_TV_cupy(data, regularisation_parameter: float = 1e-05, iterations: int = 1000, methodTV: int = 0,
               nonneg: int = 0, lipschitz_const: float = 8.0, gpu_id: int = 0, half_precision: bool = False,
               out=None) -> torch.Tensor:
    """Chambolle-Pock primal-dual TV (reference: regularisersCuPy.py:170-296)."""
    orig_shape = tuple(data.shape)
    d, is2d, axis = _prepare(data, gpu_id)
    # float32 scalar set-up of regularisersCuPy.py:215-218 (NumPy-2 weak-scalar promotion => float32 arithmetic)
    tau = np.float32(regularisation_parameter * 0.1)
    sigma = np.float32(1.0 / (lipschitz_const * tau))
    theta = np.float32(1.0)
    lt = np.float32(tau / regularisation_parameter)
    res = torch.empty_like(d) if out is None else out.view(d.shape)
    ops.pdtv(d, res, sigma, tau, lt, theta, iterations, methodTV, nonneg, half_precision)
    return _finish(res, is2d, axis, orig_shape, out, data)


def _check_if_input_2d_or_3d(data) -> Tuple[torch.Tensor, bool, int]:
    """(array, treated_as_2d, squeezed_axis): a 3D input with a singleton axis runs the 2D kernels
    (reference: regularisersCuPy.py:299-315)."""
    if data.ndim == 2:
        return (data, True, 0)
    if data.ndim == 3:
        for i, extent in enumerate(data.shape):
            if extent == 1:
                return (data.squeeze(i), True, i)
        return (data, False, 0)
    raise ValueError("2D or 3D arrays must be provided only")
