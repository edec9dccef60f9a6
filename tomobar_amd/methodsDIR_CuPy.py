"""Direct methods on MI355X behind the reference's ``RecToolsDIRCuPy`` surface
(``tomobar/methodsDIR_CuPy.py:26-150``): ``FORWPROJ``, ``BACKPROJ`` and ``FBP`` reuse the projector kernels of the
iterative path.  ``FOURIER_INV`` (log-polar / USFFT, ``methodsDIR_CuPy.py:152-989``) is outside the FISTA / ADMM hot
path and raises ``NotImplementedError``.
"""

from __future__ import annotations

import torch

from . import _lib as L
from . import ops
from .projector import HipTools3D
from .supp.funcs import _data_dims_swapper
from .supp.suppTools import _apply_horiz_detector_padding, check_kwargs


class RecToolsDIRCuPy:
    """Direct reconstruction / projection operators.

    Args mirror the reference (methodsDIR_CuPy.py:39-49, methodsDIR.py:32-69): DetectorsDimH, DetectorsDimH_pad,
    DetectorsDimV, CenterRotOffset, AnglesVec, ObjSize, projector ('astra' keeps its meaning "the 3D parallel-beam
    projector"), device_projector (GPU index).  Unlike the iterative class the reconstruction grid stays ``ObjSize``
    when the detector is padded (methodsDIR.py:44-69)."""

    def __init__(self, DetectorsDimH: int, DetectorsDimH_pad: int, DetectorsDimV, CenterRotOffset, AnglesVec,
                 ObjSize: int, projector: str = "astra", device_projector: int = 0):
        if CenterRotOffset is None:
            CenterRotOffset = 0.0
        # 2D geometry (the reference's RecToolsDIR with DetectorsDimV=None, methodsDIR.py:44-69,322-371): one slice
        # through the same kernels; data ["angles", "detX"], images [Y, X]
        self.is2d = DetectorsDimV == 0 or DetectorsDimV is None
        if self.is2d:
            DetectorsDimV = 1
        self.detectors_x_pad = DetectorsDimH_pad
        self.centre_of_rotation = CenterRotOffset
        self.angles_vec = AnglesVec
        self.recon_size = ObjSize
        self.projector = projector
        self.geom = "2D" if self.is2d else "3D"
        self.Atools = HipTools3D(DetectorsDimH, DetectorsDimH_pad, DetectorsDimV, AnglesVec, CenterRotOffset,
                                 ObjSize, "gpu", device_projector, None)

    def _swap(self, data, labels, required):
        data = ops.to_device(data, self.Atools.device_index)
        if labels is not None:
            data = _data_dims_swapper(data, labels, required)
        return data

    def _lift(self, data, labels, required3, required2):
        """Bring projection data to the 3D layout ``required3``; 2D data ([angles, detX] by default) gain a detY axis."""
        data = ops.to_device(data, self.Atools.device_index)
        if self.is2d and data.dim() == 2:
            if labels is not None:
                data = _data_dims_swapper(data, labels, required2)
            return data.unsqueeze(required3.index("detY"))
        if labels is not None:
            data = _data_dims_swapper(data, labels, required3)
        return data

    def FORWPROJ(self, data, **kwargs):
        """Forward projection of a volume ``[Z, Y, X]`` -> ``[detY, angles, detX]`` (methodsDIR_CuPy.py:70-90); in 2D
        geometry an image ``[Y, X]`` -> ``[angles, detX]`` (methodsDIR.py:71-96)."""
        data = ops.to_device(data, self.Atools.device_index)
        if self.is2d and data.dim() == 2:
            return self.Atools._forwprojCuPy(data.unsqueeze(0)).squeeze(0)
        return self.Atools._forwprojCuPy(data)

    def BACKPROJ(self, data, **kwargs):
        """Back projection of ``[detY, angles, detX]`` data (methodsDIR_CuPy.py:92-112).  The input is made
        contiguous first; the reference hands ASTRA the base pointer of a strided view (astra_base.py:533-535)."""
        flat = self.is2d and ops.to_device(data, self.Atools.device_index).dim() == 2
        data = self._lift(data, kwargs.get("data_axes_labels_order"), ["detY", "angles", "detX"], ["angles", "detX"])
        data = _apply_horiz_detector_padding(ops.contiguous(data), self.Atools.detectors_x_pad, True)
        rec = self.Atools._backprojCuPy(data)
        return rec.squeeze(0) if flat else rec

    def FBP(self, data, **kwargs):
        """Filtered back projection with the sinc-ramp filter (reference: methodsDIR_CuPy.py:114-150).

        Keyword Args: ``data_axes_labels_order`` (the data are brought to ["angles", "detY", "detX"]),
        ``recon_mask_radius``, ``cutoff_freq`` (default 0.35)."""
        cutoff = kwargs.get("cutoff_freq")
        cutoff = 0.35 if cutoff is None else cutoff
        given = data
        flat = self.is2d and ops.to_device(data, self.Atools.device_index).dim() == 2
        data = ops.contiguous(self._lift(data, kwargs.get("data_axes_labels_order"), ["angles", "detY", "detX"],
                                         ["angles", "detX"]))
        if data.dtype != torch.float32 or data.dim() != 3:
            raise ValueError("FBP expects a float32 3D array")
        data = _apply_horiz_detector_padding(data, self.Atools.detectors_x_pad, True)
        if isinstance(given, torch.Tensor) and data.data_ptr() == given.data_ptr():
            data = data.clone()  # the filter works in place: never overwrite the caller's array
        na, nz, nu = data.shape
        with torch.cuda.device(data.device):
            L.check(L.lib().tomo_fbp_filter(data.device.index, ops.ptr(data), na * nz, nu, float(cutoff),
                                            float(1.0 / na / nu), ops.stream_ptr(data)))
        sino = ops.contiguous(data.transpose(0, 1))  # [detY, angles, detX]
        del data
        rec = self.Atools._backprojCuPy(sino)
        rec = check_kwargs(rec, cupyrun=True, recon_mask_radius=kwargs.get("recon_mask_radius"))
        return rec.squeeze(0) if flat else rec

    def FOURIER_INV(self, data, **kwargs):
        raise NotImplementedError("FOURIER_INV is outside the FISTA/ADMM hot path this package accelerates")
