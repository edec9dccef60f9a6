"""TEST INFRASTRUCTURE: a stand-in for the ``cupy`` module, just large enough to drive the CuPy-in -> CuPy-out branches of
``tomobar_amd.ops`` (``to_device`` / ``like`` / ``base_ptr``) and the return-type rule of the classes built on them (the
reference returns ``cupy.ndarray``: /root/reference/tomobar/methodsIR_CuPy.py:484, regularisersCuPy.py:64-65,198-199).
CuPy-on-ROCm is not installed in this image; the product never needs it -- it only asks ``type(x).__module__`` and speaks
DLPack.  The stand-in's ``ndarray`` wraps a torch tensor (device memory stays torch's: plumbing), reports ``cupy`` as its
module, exports ``__dlpack__`` / ``__dlpack_device__`` / ``.data.ptr`` like ``cupy.ndarray`` and the module offers
``from_dlpack`` / ``asarray`` / ``asnumpy``.  Nothing here is reachable from the product package."""
import sys
import types

import numpy as np
import torch


class _MemoryPointer:
    def __init__(self, ptr):
        self.ptr = int(ptr)


class ndarray:
    """What ``cupy.ndarray`` offers at the boundary: shape / dtype / data.ptr / DLPack export."""

    def __init__(self, tensor: torch.Tensor):
        self._t = tensor

    shape = property(lambda self: tuple(self._t.shape))
    ndim = property(lambda self: self._t.dim())
    size = property(lambda self: self._t.numel())
    dtype = property(lambda self: np.dtype(str(self._t.dtype).replace("torch.", "")))
    data = property(lambda self: _MemoryPointer(self._t.data_ptr()))

    def __dlpack__(self, *args, **kwargs):
        return self._t.__dlpack__(*args, **kwargs)

    def __dlpack_device__(self):
        return self._t.__dlpack_device__()

    def get(self):
        return self._t.detach().cpu().numpy()

    def __repr__(self):
        return f"cupy-standin.ndarray(shape={self.shape}, dtype={self.dtype})"


ndarray.__module__ = "cupy"   # what tomobar_amd.ops.is_cupy looks at (cupy.ndarray lives in cupy._core.core: prefix "cupy")


def from_dlpack(x):
    return ndarray(torch.from_dlpack(x))


def asarray(x, dtype=None):
    t = torch.as_tensor(np.asarray(x, dtype=dtype))
    return ndarray(t.cuda() if torch.cuda.is_available() else t)


def asnumpy(x):
    return x.get() if isinstance(x, ndarray) else np.asarray(x)


def make_module():
    m = types.ModuleType("cupy")
    m.ndarray, m.from_dlpack, m.asarray, m.asnumpy = ndarray, from_dlpack, asarray, asnumpy
    m.__doc__ = "test stand-in for cupy (tests/_cupy_standin.py)"
    return m


def install(monkeypatch):
    """``import cupy`` resolves to the stand-in for the duration of the test."""
    module = make_module()
    monkeypatch.setitem(sys.modules, "cupy", module)
    return module
