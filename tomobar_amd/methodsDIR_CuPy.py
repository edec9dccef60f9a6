"""Direct methods on MI355X behind the reference's ``RecToolsDIRCuPy`` surface
(``tomobar/methodsDIR_CuPy.py:26-150``): ``FORWPROJ``, ``BACKPROJ`` and ``FBP`` reuse the projector kernels of the
iterative path.  ``FOURIER_INV`` (Fourier inversion on unequally spaced grids, ``methodsDIR_CuPy.py:152-989``) runs
through ``tomo_fourier_inv`` (csrc/fourier_inv.hip).
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
        geometry an image ``[Y, X]`` -> ``[angles, detX]`` (methodsDIR.py:71-96).

        Keyword Args: ``data_axes_labels_order`` -- axis order of the OUTPUT; the reference applies
        ``_data_dims_swapper(projected, value, ["detY", "angles", "detX"])`` (:84-88), so does this (the swapped
        result is returned C-contiguous)."""
        given = data
        data = ops.to_device(data, self.Atools.device_index)
        flat = self.is2d and data.dim() == 2
        projected = self.Atools._forwprojCuPy(data.unsqueeze(0) if flat else data)
        if flat:
            projected = projected.squeeze(0)
        labels = kwargs.get("data_axes_labels_order")
        if labels is not None:
            projected = ops.contiguous(_data_dims_swapper(projected, labels,
                                                          ["angles", "detX"] if flat else ["detY", "angles", "detX"]))
        return ops.like(projected, given)

    def BACKPROJ(self, data, **kwargs):
        """Back projection of ``[detY, angles, detX]`` data (methodsDIR_CuPy.py:92-112).  The input is made
        contiguous first; the reference hands ASTRA the base pointer of a strided view (astra_base.py:533-535)."""
        given = data
        flat = self.is2d and ops.to_device(data, self.Atools.device_index).dim() == 2
        data = self._lift(data, kwargs.get("data_axes_labels_order"), ["detY", "angles", "detX"], ["angles", "detX"])
        data = _apply_horiz_detector_padding(ops.contiguous(data), self.Atools.detectors_x_pad, True)
        rec = self.Atools._backprojCuPy(data)
        return ops.like(rec.squeeze(0) if flat else rec, given)

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
        if data.data_ptr() == ops.base_ptr(given):
            data = data.clone()  # the filter works in place: never overwrite the caller's (torch or cupy) array
        na, nz, nu = data.shape
        with torch.cuda.device(data.device):
            L.check(L.lib().tomo_fbp_filter(data.device.index, ops.ptr(data), na * nz, nu, float(cutoff),
                                            float(1.0 / na / nu), ops.stream_ptr(data)))
        sino = ops.contiguous(data.transpose(0, 1))  # [detY, angles, detX]
        del data
        rec = self.Atools._backprojCuPy(sino)
        rec = check_kwargs(rec, cupyrun=True, recon_mask_radius=kwargs.get("recon_mask_radius"))
        return ops.like(rec.squeeze(0) if flat else rec, given)

    def FOURIER_INV(self, data, **kwargs):
        """Fourier direct inversion on unequally spaced grids (reference: methodsDIR_CuPy.py:152-447, after V. Nikitin's
        radonusfft): oversampled FBP filter, 1D FFT of slice pairs, Gaussian gathering onto a 2n x 2n frequency grid, 2D
        inverse FFT, deconvolution.  Device work: ``tomo_fourier_inv`` (csrc/fourier_inv.hip).

        Keyword Args (the reference's): ``data_axes_labels_order`` (data are brought to ["detY", "angles", "detX"]),
        ``recon_mask_radius``, ``filter_type`` (none, ramp, shepp (default), cosine, cosine2, hamming, hann, parzen),
        ``cutoff_freq`` (1.0), ``center_size`` (32768), ``padding`` (0), ``power_of_2_oversampling`` (True),
        ``power_of_2_cropping`` (False).  The reference's launch-shape and memory-chunking knobs (``block_dim``,
        ``block_dim_center``, ``chunk_count``, ``min_mem_usage_filter``, ``min_mem_usage_ifft2``) are accepted and have
        no effect here: this implementation always works in chunks of 128 slices.

        Dry run: inside ``with DeviceMemStack():`` and called with the data SHAPE (tuple) plus ``data_dtype``, records the
        device memory it would allocate and returns the output shape (reference: the ``mem_stack`` branches)."""
        import math

        import numpy as np

        from . import fourier as FT
        from .supp.memory_estimator_helpers import DeviceMemStack

        cutoff_freq, filter_type = 1.0, "shepp"
        center_size, padding = 32768, 0
        power_of_2_oversampling, power_of_2_cropping = True, False
        oversampling_level = 4
        labels = None
        for key, value in kwargs.items():
            if value is None:
                continue
            if key == "data_axes_labels_order":
                labels = value
            elif key == "center_size":
                center_size = int(value)
            elif key == "cutoff_freq":
                cutoff_freq = value
            elif key == "filter_type":
                if value not in FT.FILTER_NAMES:
                    print("Unknown filter name, please use: none, ramp, shepp, cosine, cosine2, hamming, hann or parzen. "
                          "Set to shepp filter")
                else:
                    filter_type = value
            elif key == "power_of_2_oversampling":
                power_of_2_oversampling = bool(value)
            elif key == "power_of_2_cropping":
                power_of_2_cropping = bool(value)
            elif key == "padding":
                if not isinstance(value, int) or value < 0:
                    print(f"Invalid padding: {value}. Set to 0")
                else:
                    padding = value
            elif key == "chunk_count":
                if not isinstance(value, int) or value <= 0:
                    print(f"Invalid chunk count: {value}. Set to 1")

        mem_stack = DeviceMemStack.instance()
        dry = mem_stack is not None and isinstance(data, (tuple, list))
        if dry:
            shape = tuple(int(v) for v in data)
            if labels is not None:
                shape = tuple(shape[list(labels).index(k)] for k in ["detY", "angles", "detX"])
            nz, nproj, data_n = shape
        else:
            given = data
            data = ops.to_device(data, self.Atools.device_index)
            if labels is not None:
                data = _data_dims_swapper(data, labels, ["detY", "angles", "detX"])
            if data.dtype != torch.float32 or data.dim() != 3:
                raise ValueError("FOURIER_INV expects a float32 3D array")
            nz, nproj, data_n = (int(v) for v in data.shape)
        if nproj != len(self.angles_vec):
            raise ValueError(f"projection data has {nproj} angles, the geometry {len(self.angles_vec)}")
        recon_size = self.recon_size
        if recon_size > data_n:
            raise ValueError("The reconstruction size {} should not be larger than the size of the horizontal detector {}"
                             .format(recon_size, data_n))
        if np.ndim(self.centre_of_rotation) != 0:
            raise ValueError("FOURIER_INV needs a scalar CenterRotOffset")
        odd_horiz, odd_vert = data_n % 2, nz % 2
        raw_n, nz_even = data_n + odd_horiz, nz + odd_vert
        n = raw_n + self.detectors_x_pad * 2 + padding * 2
        if power_of_2_cropping:
            n_pow2 = 2 ** math.ceil(math.log2(n))
            if 0.9 < n / n_pow2:
                n = n_pow2
        center_size = min(center_size, n * 2)
        eps = 1e-4  # accuracy of the unequally spaced FFT
        mu = -np.log(eps) / (2 * n * n)
        m = FT.footprint_half_width(n, mu, eps)
        ne = FT.oversampled_width(raw_n, n, power_of_2_oversampling, oversampling_level)
        odd_recon = recon_size % 2
        unpad_m = (n - odd_horiz) // 2 - recon_size // 2
        unpad_p = (n - odd_horiz) // 2 + (recon_size + odd_recon) // 2
        size = unpad_p - unpad_m
        out_shape = (nz, size, size)
        if dry:
            itemsize = np.dtype(kwargs.get("data_dtype", np.float32)).itemsize
            zc = min(nz_even // 2, 64)
            rows_sub = max(1, min(zc * nproj, (64 << 20) // ne))
            sizes = [int(np.prod(shape)) * itemsize, nz_even * nproj * raw_n * 4 if (odd_horiz or odd_vert) else 0,
                     rows_sub * ne * 8, 0, zc * nproj * n * 8, nproj * n * 64 * 8,
                     (2 * n) * (2 * n) * zc * 8, int(np.prod(out_shape)) * 4]
            for b in sizes:
                if b:
                    mem_stack.malloc(b)
            for b in sizes[1:7]:
                if b:
                    mem_stack.free(b)
            return out_shape

        data = ops.contiguous(data)
        if odd_horiz or odd_vert:  # methodsDIR_CuPy.py:265-279: replicate the last column, add a zero slice
            padded = torch.zeros((nz_even, nproj, raw_n), dtype=torch.float32, device=data.device)
            padded[:nz, :, :data_n] = data
            if odd_horiz:
                padded[:nz, :, -1] = data[..., -1]
            data = padded
        w = FT.filter_with_phase(ne, filter_type, float(cutoff_freq), float(self.centre_of_rotation) + 0.5)
        theta = np.ascontiguousarray(-np.asarray(self.angles_vec, dtype=np.float64), dtype=np.float32)
        out = torch.empty(out_shape, dtype=torch.float32, device=data.device)
        with torch.cuda.device(data.device):
            L.check(L.lib().tomo_fourier_inv(data.device.index, ops.ptr(data), ops.ptr(out), nz_even, nz, nproj, raw_n, n, ne,
                                             unpad_m, size, w.ctypes.data, theta.ctypes.data, int(m), float(mu),
                                             int(center_size), ops.stream_ptr(data)))
        return ops.like(check_kwargs(out, cupyrun=True, recon_mask_radius=kwargs.get("recon_mask_radius")), given)
