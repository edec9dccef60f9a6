"""Gradient of the data-fidelity term (reference: ``tomobar/data_fidelities.py:7-40``).

The reference spends one forward projection, one to three element-wise sinogram passes and one back projection
per call.  Here the residual (LS / PWLS / KL, including the ordered-subset gather of ``b`` and ``w`` by angle
index) is the epilogue of the forward-projection kernel, so only the two projector launches remain.
"""

from __future__ import annotations

from typing import Optional

import torch

from . import ops


def grad_data_term(self, x, b, use_os: bool, sub_ind: int, indVec=None, w: Optional[torch.Tensor] = None):
    """``A_s^T (w_s * (A_s x - b_s))`` for LS / PWLS, ``A_s^T (1 - b_s / max(A_s x, 1e-8))`` for KL.

    Same arguments as the reference: ``b`` is the subset's projection data ``[detY, len(indVec), detX]`` and
    ``w`` the FULL weight array (gathered by angle index inside the kernel)."""
    if self.data_fidelity not in ("LS", "PWLS", "KL"):
        raise ValueError("_data_['data_fidelity'] should be provided as 'LS', 'PWLS', 'KL'.")
    A = self.Atools
    os_index = sub_ind if use_os else None
    x = A._vol_in(x)
    b = A._sino_in(b, os_index)
    fid = self.data_fidelity
    if fid in ("LS", "PWLS"):
        fid = "PWLS" if w is not None else "LS"
    if w is not None:
        w = ops.contiguous(ops.to_device(w, A.device_index))
    res = torch.empty(A.sino_shape(os_index), dtype=torch.float32, device=x.device)
    A.residual(x, b, w if fid == "PWLS" else None, fid, os_index, res, gathered=1)
    return A.backward(res, os_index)
