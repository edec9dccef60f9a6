"""Direct methods on MI355X behind the reference's ``RecToolsDIRCuPy`` surface
(``tomobar/methodsDIR_CuPy.py:26-150``): ``FORWPROJ``, ``BACKPROJ`` and ``FBP`` reuse the projector kernels of the
iterative path.  ``FOURIER_INV`` (log-polar / USFFT, ``methodsDIR_CuPy.py:152-989``) is outside the FISTA / ADMM hot
path and raises ``NotImplementedError``.
"""

from __future__ import annotations

import numpy as np
import torch

from . import ops
from .projector import HipTools3D
from .supp.funcs import _data_dims_swapper
from .supp.suppTools import _apply_horiz_detector_padding, check_kwargs, perform_recon_crop


class RecToolsDIRCuPy:
    """Direct reconstruction / projection operators.

    Args mirror the reference (methodsDIR_CuPy.py:39-49): DetectorsDimH, DetectorsDimH_pad, DetectorsDimV,
    CenterRotOffset, AnglesVec, ObjSize, projector ('astra' keeps its meaning "the 3D parallel-beam projector"),
    device_projector (GPU index)."""

    def __init__(self, DetectorsDimH: int, DetectorsDimH_pad: int, DetectorsDimV, CenterRotOffset, AnglesVec,
                 ObjSize: int, projector: str = "astra", device_projector: int = 0):
        self.objsize_user_given = ObjSize if DetectorsDimH_pad != 0 else None
        if DetectorsDimH_pad > 0:
            ObjSize = DetectorsDimH + 2 * DetectorsDimH_pad
        if DetectorsDimV == 0 or DetectorsDimV is None:
            DetectorsDimV = 1
        self.projector = projector
        self.Atools = HipTools3D(DetectorsDimH, DetectorsDimH_pad, DetectorsDimV, AnglesVec, CenterRotOffset,
                                 ObjSize, "gpu", device_projector, None)

    def _canonical(self, data, labels):
        data = ops.to_device(data, self.Atools.device_index)
        if labels is not None:
            data = _data_dims_swapper(data, labels, ["detY", "angles", "detX"])
        return ops.contiguous(data)

    def FORWPROJ(self, data, **kwargs):
        """Forward projection of a volume ``[Z, Y, X]`` -> ``[detY, angles, detX]`` (methodsDIR_CuPy.py:70-90)."""
        return self.Atools._forwprojCuPy(data)

    def BACKPROJ(self, data, **kwargs):
        """Back projection of ``[detY, angles, detX]`` data (methodsDIR_CuPy.py:92-112).  The input is made
        contiguous first; the reference hands ASTRA the base pointer of a strided view (astra_base.py:533-535)."""
        data = self._canonical(data, kwargs.get("data_axes_labels_order"))
        return self.Atools._backprojCuPy(data)

    def FOURIER_INV(self, data, **kwargs):
        raise NotImplementedError("FOURIER_INV is outside the FISTA/ADMM hot path this package accelerates")
