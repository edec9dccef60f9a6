"""Direct methods on numpy arrays behind the reference's ``RecToolsDIR`` surface (``tomobar/methodsDIR.py:18-320``).

* ``device_projector="cpu"`` (BASELINE configs[0], the reference's numpy / ASTRA-CPU plumbing path): 2D geometry only, like
  the reference (its 3D CPU projector raises, ``astra_tools3d.py:56-59``; a non-zero CoR is rejected, ``astra_base.py:
  150-153``).  ``FBP`` is the customised sinc-ramp filter of ``_filtersinc2D`` (:295-320) followed by a back projection
  (:121-175).  ASTRA's CPU ``line`` projector is not part of this package: the host projector pair of
  ``libtomo_mi355x.so`` (``tomo_host_bp2d`` / ``tomo_host_fp2d``: the same voxel-driven / Joseph model and arithmetic as
  the GPU kernels) stands behind ``BACKPROJ`` / ``FORWPROJ``.  No GPU is needed for this device.
* ``device_projector="gpu"`` or a GPU index: numpy in, numpy out around :class:`tomobar_amd.methodsDIR_CuPy.RecToolsDIRCuPy`.
"""

from __future__ import annotations

import ctypes as C

import numpy as np
import scipy.fft

from . import _lib as L
from .supp.funcs import _data_dims_swapper, _parse_device_argument


def _sinc_ramp_filter(n: int, cutoff: float = 1.1) -> np.ndarray:
    """The frequency response of ``_filtersinc2D`` / ``_filtersinc3D`` (methodsDIR.py:257-320), ifftshift-ed:
    ``|2/a sin(a w / 2)| * (<sin(a w/2), a w/2> / ||a w/2||^2)^2`` on ``w = linspace(-pi, pi - 2 pi / n, n)``."""
    w = np.linspace(-np.pi, np.pi - (2 * np.pi) / n, n, dtype="float32")
    rd = (cutoff * w) / 2.0
    rn2 = np.sin(rd)
    # dot(rn2, pinv(row vector rd)) = <rn2, rd> / ||rd||^2, in float64 like the reference (its rd_c is a float64 array
    # holding the float32 values, methodsDIR.py:273-276,309-312), so the response itself is float64 too
    rd64, rn64 = rd.astype(np.float64), rn2.astype(np.float64)
    gain = (np.dot(rn64, rd64) / np.dot(rd64, rd64)) ** 2
    return scipy.fft.fftshift(np.abs(2.0 / cutoff * rn2) * gain)


def _filtersinc2D(sinogram: np.ndarray) -> np.ndarray:
    """Row-wise FBP filter of ``[angles, detX]`` data, scaled by 1/angles (methodsDIR.py:295-320), all rows at once."""
    na, nu = sinogram.shape
    f = _sinc_ramp_filter(nu)
    spec = scipy.fft.fft(np.asarray(sinogram), axis=1) * f[None, :]
    return np.float32((1.0 / na) * np.real(scipy.fft.ifft(spec, axis=1)))


def _circular_mask_(img: np.ndarray, radius) -> np.ndarray:
    """suppTools.py:364-396 on a numpy image / volume, in place."""
    if radius is None:
        return img
    n = img.shape[-1]
    h = n // 2
    Y, X = np.ogrid[:n, :n]
    dist = np.sqrt((X - h) ** 2 + (Y - h) ** 2)
    lim = h - abs(h - h / radius) if radius <= 1.0 else h + abs(h - h / radius)
    img *= dist <= lim
    return img


class _HostTools2D:
    """The ``Atools`` of a CPU 2D geometry: ``_forwproj`` / ``_backproj`` on numpy arrays (AstraTools2D's surface,
    astra_tools2d.py:8-123)."""

    def __init__(self, detectors_x, detectors_x_pad, angles_vec, centre_of_rotation, recon_size):
        if detectors_x <= 0:
            raise ValueError("The size of the horizontal detector cannot be negative or zero")
        if detectors_x_pad < 0:
            raise ValueError("The padding size of the horizontal detector cannot be negative")
        if len(angles_vec) == 0:
            raise ValueError("The length of angles array cannot be zero")
        if recon_size <= 0:
            raise ValueError("The size of the reconstruction object cannot be zero")
        cor = 0.0 if centre_of_rotation is None else centre_of_rotation
        if np.ndim(cor) != 0 or float(cor) != 0.0:
            raise ValueError("The CoR offset is not supported on the CPU device, please use the GPU")  # astra_base.py:150-153
        self.detectors_x, self.detectors_x_pad = int(detectors_x), int(detectors_x_pad)
        self.nu = self.detectors_x + 2 * self.detectors_x_pad
        self.recon_size = int(recon_size)
        self.angles_vec = np.ascontiguousarray(angles_vec, dtype=np.float64)
        self.processing_arch, self.device_index = "cpu", -1

    def _angles(self):
        return self.angles_vec.ctypes.data_as(C.POINTER(C.c_double))

    def _backproj(self, sinogram: np.ndarray) -> np.ndarray:
        s = np.ascontiguousarray(sinogram, dtype=np.float32)
        if s.shape != (self.angles_vec.size, self.nu):
            raise ValueError(f"projection data has shape {s.shape}, expected {(self.angles_vec.size, self.nu)}")
        img = np.empty((self.recon_size, self.recon_size), dtype=np.float32)
        L.check(L.lib().tomo_host_bp2d(s.ctypes.data, img.ctypes.data, self.recon_size, self.nu, self.angles_vec.size,
                                       self._angles(), 0.0))
        return img

    def _forwproj(self, image: np.ndarray) -> np.ndarray:
        x = np.ascontiguousarray(image, dtype=np.float32)
        if x.shape != (self.recon_size, self.recon_size):
            raise ValueError(f"object has shape {x.shape}, expected {(self.recon_size, self.recon_size)}")
        s = np.empty((self.angles_vec.size, self.nu), dtype=np.float32)
        L.check(L.lib().tomo_host_fp2d(x.ctypes.data, s.ctypes.data, self.recon_size, self.nu, self.angles_vec.size,
                                       self._angles(), 0.0))
        return s


class RecToolsDIR:
    """Reconstruction class using DIRect methods on numpy arrays (reference: methodsDIR.py:18-69).

    Args: DetectorsDimH, DetectorsDimH_pad, DetectorsDimV (0 / None for 2D), CenterRotOffset, AnglesVec, ObjSize,
    projector ("astra" keeps its meaning "the parallel-beam projector"), device_projector ('cpu', 'gpu' or a GPU index)."""

    def __init__(self, DetectorsDimH, DetectorsDimH_pad, DetectorsDimV, CenterRotOffset, AnglesVec, ObjSize,
                 projector: str = "astra", device_projector="gpu"):
        arch, index = _parse_device_argument(device_projector)
        self.geom = "2D" if (DetectorsDimV == 0 or DetectorsDimV is None) else "3D"
        self._gpu = None
        if arch == "cpu":
            if self.geom == "3D":
                raise ValueError("3D CPU reconstruction is not supported, please use GPU")  # astra_tools3d.py:56-59
            self.Atools = _HostTools2D(DetectorsDimH, DetectorsDimH_pad, AnglesVec, CenterRotOffset, ObjSize)
        else:
            from .methodsDIR_CuPy import RecToolsDIRCuPy
            self._gpu = RecToolsDIRCuPy(DetectorsDimH, DetectorsDimH_pad, DetectorsDimV, CenterRotOffset, AnglesVec, ObjSize,
                                        projector, index)
            self.Atools = self._gpu.Atools

    def _labels(self):
        return ["angles", "detX"] if self.geom == "2D" else ["detY", "angles", "detX"]

    def FORWPROJ(self, data: np.ndarray, **kwargs) -> np.ndarray:
        """Forward projection (methodsDIR.py:71-96); ``data_axes_labels_order`` orders the OUTPUT."""
        if self._gpu is not None:
            return self._gpu.FORWPROJ(data, **kwargs).cpu().numpy()
        projected = self.Atools._forwproj(data)
        labels = kwargs.get("data_axes_labels_order")
        return projected if labels is None else _data_dims_swapper(projected, labels, self._labels())

    def BACKPROJ(self, data: np.ndarray, **kwargs) -> np.ndarray:
        """Back projection (methodsDIR.py:98-119)."""
        if self._gpu is not None:
            return self._gpu.BACKPROJ(data, **kwargs).cpu().numpy()
        return self.Atools._backproj(self._prepare(data, kwargs))

    def FBP(self, data: np.ndarray, **kwargs) -> np.ndarray:
        """Filtered back projection (methodsDIR.py:121-175): on the CPU device the customised sinc filter + back
        projection; keyword ``recon_mask_radius`` applies the circular mask (``check_kwargs``, suppTools.py:462-467)."""
        if self._gpu is not None:
            return self._gpu.FBP(data, **kwargs).cpu().numpy()
        rec = self.Atools._backproj(_filtersinc2D(self._prepare(data, kwargs)))
        return _circular_mask_(rec, kwargs.get("recon_mask_radius"))

    def _prepare(self, data, kwargs):
        labels = kwargs.get("data_axes_labels_order")
        if labels is not None:
            data = _data_dims_swapper(data, labels, self._labels())
        pad = self.Atools.detectors_x_pad
        if pad > 0:
            data = np.pad(data, ((0, 0), (pad, pad)), mode="edge")   # _apply_horiz_detector_padding, suppTools.py:425-459
        return data
