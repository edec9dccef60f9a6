"""Validation and default population of the ``_data_`` / ``_algorithm_`` / ``_regularisation_`` dictionaries.

Behavioural mirror of the reference's ``tomobar/supp/dicts.py:6-184`` (same keys, defaults, error types and the
same in-place effects on the caller's dictionaries), written table-driven.  Divergence, on purpose: the axis
swap materialises a C-contiguous device array (the reference keeps a strided view and later hands its raw
pointer to ASTRA, astra_base.py:533-535).
"""

from __future__ import annotations

from typing import Optional

from .. import ops
from .funcs import _data_dims_swapper

LABELS_3D = ["detY", "angles", "detX"]
LABELS_2D = ["angles", "detX"]

# default outer-iteration counts: method -> (classical, ordered-subsets)   (dicts.py:101-133)
_DEFAULT_ITERATIONS = {
    "SIRT": (200, 200), "CGLS": (30, 30), "power": (15, 15), "Landweber": (1500, 1500),
    "OSEM": (300, 15), "FISTA": (400, 20), "ADMM": (400, 10),
}
_NO_OS_METHODS = {"SIRT", "CGLS", "Landweber"}
_NO_LIPSCHITZ = {"SIRT", "CGLS", "power", "Landweber", "OSEM"}

_ALGORITHM_DEFAULTS = (("initialise", None), ("nonnegativity", False), ("recon_mask_radius", 1.0),
                       ("tolerance", 0.0), ("verbose", False))
_REGULARISATION_DEFAULTS = (("regul_param", 0.001), ("iterations", 150), ("tolerance", 0.0),
                            ("time_marching_step", 0.005), ("PD_LipschitzConstant", 12.0), ("methodTV", 0),
                            ("device_regulariser", 0))


def dicts_check(self, _data_: dict, _algorithm_: Optional[dict] = None, _regularisation_: Optional[dict] = None,
                method_run: str = "FISTA") -> tuple:
    """Populate the three dictionaries in place and return them (see module docstring)."""
    # ------------------------------------------------------------------ _data_
    if _data_ is None:
        raise NameError("The data dictionary must be always provided")
    if _data_.get("projection_data") is None:
        raise NameError("'projection_data' needs to be provided")
    device_index = self.Atools.device_index
    data = ops.to_device(_data_["projection_data"], device_index)
    is_2d = data.ndim == 2
    labels = _data_.setdefault("data_axes_labels_order", None)
    if labels is not None:
        data = _data_dims_swapper(data, labels, LABELS_2D if is_2d else LABELS_3D)
        _data_["data_axes_labels_order"] = None  # swapped once; never again
    if is_2d:
        data = data.unsqueeze(0)
    _data_["projection_data"] = ops.contiguous(data)

    if _data_.get("data_fidelity") is None:
        _data_["data_fidelity"] = "LS"
    # "SWLS" (stripe-weighted least squares) and the Group-Huber keys below are the ring-artefact data terms of the
    # reference's removed RecToolsIR class (docs/source/tutorials/real_data_recon.rst:100-151); the reference's current
    # dicts_check knows LS / PWLS / KL only
    if _data_["data_fidelity"] not in {"LS", "PWLS", "KL", "SWLS"}:
        raise ValueError("_data_['data_fidelity'] should be provided as 'LS', 'PWLS', 'KL' (or 'SWLS').")
    self.data_fidelity = _data_["data_fidelity"]
    _data_.setdefault("ringGH_lambda", None)        # Group-Huber offsets: off unless a threshold is given
    _data_.setdefault("ringGH_accelerate", 50)
    _data_.setdefault("beta_SWLS", 0.1)
    if _data_["ringGH_lambda"] is not None and _data_["data_fidelity"] not in {"LS", "PWLS"}:
        raise ValueError("the Group-Huber ring term (ringGH_lambda) combines with the 'LS' and 'PWLS' data fidelities only")
    if _data_["data_fidelity"] == "SWLS" and method_run != "FISTA":
        raise ValueError("the 'SWLS' data fidelity is available in FISTA only")
    if _data_["ringGH_lambda"] is not None and method_run != "FISTA":
        raise ValueError("the Group-Huber ring term (ringGH_lambda) is available in FISTA only")

    # Huber / Student's-t data terms (outlier-robust re-weighting of the residual): keys of the removed RecToolsIR class as
    # well (Demos/methods_IR_legacy/DemoFISTA_artifacts2D.py:197,263,307,348; docs/source/introduction/about.rst:38)
    _data_.setdefault("huber_threshold", None)
    _data_.setdefault("studentst_threshold", None)
    for key in ("huber_threshold", "studentst_threshold"):
        if _data_[key] is not None:
            if method_run != "FISTA":
                raise ValueError(f"the robust data term ({key}) is available in FISTA only")
            if _data_["data_fidelity"] == "KL":
                raise ValueError(f"the robust data term ({key}) combines with the 'LS', 'PWLS' and 'SWLS' data fidelities only")
            if not float(_data_[key]) > 0.0:
                raise ValueError(f"_data_['{key}'] must be a positive threshold")
    if _data_["huber_threshold"] is not None and _data_["studentst_threshold"] is not None:
        raise ValueError("give either 'huber_threshold' or 'studentst_threshold', not both")

    if self.OS_number > 1 and method_run in _NO_OS_METHODS:
        raise NameError(
            "There is no ordered-subsets implementation for this reconstruction method, please set OS_number=None")

    # ------------------------------------------------------------------ _algorithm_
    if _algorithm_ is None:
        _algorithm_ = {}
    if method_run in _NO_LIPSCHITZ:
        _algorithm_["lipschitz_const"] = 0  # these methods never need it
        if _algorithm_.get("tau_step_lanweber") is None:
            _algorithm_["tau_step_lanweber"] = 1e-05
    if _algorithm_.get("iterations") is None and method_run in _DEFAULT_ITERATIONS:
        classical, os_count = _DEFAULT_ITERATIONS[method_run]
        _algorithm_["iterations"] = os_count if self.OS_number > 1 else classical
    if method_run == "ADMM":
        _algorithm_.setdefault("ADMM_rho_const", 1.0)
        _algorithm_.setdefault("ADMM_relax_par", 1.6)
    for key, value in _ALGORITHM_DEFAULTS:
        _algorithm_.setdefault(key, value)
    if _algorithm_["nonnegativity"] not in [True, False]:
        raise ValueError("_algorithm_['nonnegativity'] should be set to True or False.")
    self.nonneg_regul = 1 if _algorithm_["nonnegativity"] else 0

    # ------------------------------------------------------------------ _regularisation_
    if _regularisation_ is None:
        _regularisation_ = {}
    if not _regularisation_:
        _regularisation_["method"] = None
    if method_run in {"FISTA", "ADMM", "OSEM"}:
        for key, value in _REGULARISATION_DEFAULTS:
            _regularisation_.setdefault(key, value)
    return (_data_, _algorithm_, _regularisation_)
