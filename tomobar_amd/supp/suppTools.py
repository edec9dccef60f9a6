"""Pre/post-processing glue of the iterative methods (reference: ``tomobar/supp/suppTools.py:364-467``),
executed by HIP kernels of ``libtomo_mi355x.so`` on device tensors."""

from __future__ import annotations

import numpy as np
import torch

from .. import ops


def apply_circular_mask(data, recon_mask_radius, cupyrun=True):
    """Zero everything outside the disc of suppTools.py:364-396, in place (like the reference's ``data *= mask``)."""
    if isinstance(data, np.ndarray):
        raise ValueError("device arrays only: the numpy path of the reference is outside this package's scope")
    vol = data if data.dim() == 3 else data.unsqueeze(0)
    if not vol.is_contiguous():
        raise ValueError("the reconstruction must be C-contiguous")
    ops.circ_mask_(vol, recon_mask_radius)
    return data


def perform_recon_crop(data, croped_size: int):
    """Centre crop of the in-plane axes to ``croped_size`` (suppTools.py:399-422); returns a new contiguous array."""
    vol = data if data.dim() == 3 else data.unsqueeze(0)
    out = ops.crop_center(ops.contiguous(vol), int(croped_size))
    return out if data.dim() == 3 else out[0]


def _apply_horiz_detector_padding(data, detector_width_pad: int, cupyrun=True):
    """Edge-pad the detX axis of ``[detY, angles, detX]`` (or 2D ``[angles, detX]``) data (suppTools.py:425-459)."""
    if detector_width_pad <= 0:
        return data
    if isinstance(data, np.ndarray):
        width = ((0, 0),) * (data.ndim - 1) + ((detector_width_pad, detector_width_pad),)
        return np.pad(data, pad_width=width, mode="edge")
    b = data if data.dim() == 3 else data.unsqueeze(0)
    out = ops.pad_edge(ops.contiguous(b), int(detector_width_pad))
    return out if data.dim() == 3 else out[0]


def check_kwargs(reconstruction, **kwargs):
    """Optional post-processing switches (suppTools.py:462-467): only ``recon_mask_radius`` is acted upon."""
    radius = kwargs.get("recon_mask_radius")
    if radius is not None:
        apply_circular_mask(reconstruction, radius, kwargs.get("cupyrun", True))
    return reconstruction
