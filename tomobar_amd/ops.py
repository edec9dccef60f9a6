"""Tensor-level wrappers over the C-ABI: device memory and streams come from torch (plumbing only),
every arithmetic operation is a HIP kernel in ``libtomo_mi355x.so``.

Arrays are ``torch.Tensor`` (float32, device ``cuda:<index>``) where the reference uses ``cupy.ndarray``.
"""

from __future__ import annotations

import contextlib
import ctypes as C
import threading

import numpy as np
import torch

from . import _lib as L


def _require_gpu(device_index: int) -> torch.device:
    if not torch.cuda.is_available():
        raise L.TomoRuntimeError(
            "no ROCm device is visible to torch: tomobar_amd runs on MI355X (gfx950) only and has no CPU fallback")
    return torch.device("cuda", int(device_index))


def to_device(x, device_index: int = 0) -> torch.Tensor:
    """numpy / torch / cupy (DLPack or __cuda_array_interface__) object -> float32-preserving device tensor (no dtype
    change, no copy for arrays that already live on the device)."""
    dev = _require_gpu(device_index)
    if isinstance(x, torch.Tensor):
        return x.to(dev)
    if isinstance(x, np.ndarray):
        return torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    if is_cupy(x) and hasattr(x, "__dlpack__"):
        return torch.from_dlpack(x).to(dev)
    return torch.as_tensor(x, device=dev)


def is_cupy(x) -> bool:
    return type(x).__module__.split(".")[0] == "cupy"


def base_ptr(x):
    """Device address of a torch / cupy array (None for anything else): used to detect zero-copy views of caller memory."""
    if isinstance(x, torch.Tensor):
        return x.data_ptr()
    if is_cupy(x):
        return int(x.data.ptr)
    return None


def like(result, given):
    """`result` (a device tensor) in the array library the caller used for `given`: the reference's classes return
    ``cupy.ndarray`` (methodsIR_CuPy.py:484), so a caller that hands CuPy arrays in gets CuPy arrays back (zero-copy through
    DLPack) wherever CuPy-on-ROCm is installed; every other caller gets the ``torch.Tensor``.  CuPy is never required."""
    if isinstance(result, torch.Tensor) and is_cupy(given):
        import cupy
        return cupy.from_dlpack(result)
    return result


def stream_ptr(t: torch.Tensor):
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _on(t: torch.Tensor):
    """Context making the tensor's device current for a launch (the launch stream belongs to that device)."""
    return torch.cuda.device(t.device)


def ptr(t):
    return C.c_void_p(0 if t is None else t.data_ptr())


def _chk_f32(t: torch.Tensor, name="array"):
    if t.dtype != torch.float32:
        raise ValueError(f"{name} must be float32")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be C-contiguous")
    if not t.is_cuda:
        raise ValueError(f"{name} must live on the GPU")


def contiguous(t: torch.Tensor) -> torch.Tensor:
    """Materialise a (possibly permuted) <=3-D float32 view as a C-contiguous tensor with tomo_permute3."""
    if t.is_contiguous():
        return t
    if t.dtype != torch.float32 or t.dim() > 3:
        raise ValueError("only float32 arrays of at most 3 dimensions are supported")
    shape = list(t.shape)
    strides = list(t.stride())
    while len(shape) < 3:
        shape.insert(0, 1)
        strides.insert(0, 0)
    out = torch.empty(t.shape, dtype=torch.float32, device=t.device)
    with torch.cuda.device(t.device):
        L.check(L.lib().tomo_permute3(ptr(t), ptr(out), shape[0], shape[1], shape[2],
                                      strides[0], strides[1], strides[2], stream_ptr(t)))
    return out


# ----------------------------------------------------------------------------- element-wise / reductions
def momentum(x, x_old, x_t, beta):
    with _on(x):
        L.check(L.lib().tomo_momentum(ptr(x), ptr(x_old), ptr(x_t), float(beta), x.numel(), stream_ptr(x)))


def admm_dual(u, z, x):
    with _on(u):
        L.check(L.lib().tomo_admm_dual(ptr(u), ptr(z), ptr(x), u.numel(), stream_ptr(u)))


def axpby(a, x, b, y):
    with _on(y):
        L.check(L.lib().tomo_axpby(float(a), ptr(x), float(b), ptr(y), y.numel(), stream_ptr(y)))


def scale(a, x, y):
    with _on(y):
        L.check(L.lib().tomo_scale(float(a), ptr(x), ptr(y), y.numel(), stream_ptr(y)))


def clamp_min(x, lo=0.0):
    with _on(x):
        L.check(L.lib().tomo_clamp_min(ptr(x), float(lo), x.numel(), stream_ptr(x)))


def mul(x, y):
    with _on(y):
        L.check(L.lib().tomo_mul(ptr(x), ptr(y), y.numel(), stream_ptr(y)))


def recip_safe(x, y):
    with _on(y):
        L.check(L.lib().tomo_recip_safe(ptr(x), ptr(y), y.numel(), stream_ptr(y)))


def fill(x, value):
    with _on(x):
        L.check(L.lib().tomo_fill(ptr(x), float(value), x.numel(), stream_ptr(x)))


def norm2(x) -> float:
    out = C.c_double(0.0)
    with torch.cuda.device(x.device):
        L.check(L.lib().tomo_norm2(ptr(x), x.numel(), C.byref(out), stream_ptr(x)))
    return out.value


def dot(x, y) -> float:
    out = C.c_double(0.0)
    with torch.cuda.device(x.device):
        L.check(L.lib().tomo_dot(ptr(x), ptr(y), x.numel(), C.byref(out), stream_ptr(x)))
    return out.value


def pwls_weights(b, slab=None):
    """w = max(b, 1e-6) / max(max(b, 1e-6)); with a SlabComm the maximum is taken over all z-slabs."""
    w = torch.empty_like(b)
    with torch.cuda.device(b.device):
        if slab is None:
            L.check(L.lib().tomo_pwls_weights(ptr(b), ptr(w), b.numel(), stream_ptr(b)))
        else:
            m = C.c_float(0.0)
            L.check(L.lib().tomo_pwls_max(ptr(b), b.numel(), C.byref(m), stream_ptr(b)))
            wmax = np.float32(slab.allreduce_max(float(m.value)))
            L.check(L.lib().tomo_pwls_weights_scaled(ptr(b), ptr(w), b.numel(), float(wmax), stream_ptr(b)))
    return w


# ----------------------------------------------------------------------------- pre / post
def pad_edge(b, pad):
    nz, na, nu0 = b.shape
    out = torch.empty((nz, na, nu0 + 2 * pad), dtype=torch.float32, device=b.device)
    with _on(b):
        L.check(L.lib().tomo_pad_edge(ptr(b), ptr(out), nz * na, nu0, pad, stream_ptr(b)))
    return out


def crop_center(vol, m):
    nz, n, _ = vol.shape
    out = torch.empty((nz, m, m), dtype=torch.float32, device=vol.device)
    with _on(vol):
        L.check(L.lib().tomo_crop_center(ptr(vol), ptr(out), nz, n, m, stream_ptr(vol)))
    return out


def circ_mask_(vol, radius):
    nz, n, _ = vol.shape
    with _on(vol):
        L.check(L.lib().tomo_circ_mask(ptr(vol), nz, n, float(radius), stream_ptr(vol)))
    return vol


# ----------------------------------------------------------------------------- TV
def _tv_dims(t):
    if t.dim() == 2:
        return t.shape[1], t.shape[0], 1, 2
    return t.shape[2], t.shape[1], t.shape[0], 3


def pdtv(data, out, sigma, tau, lt, theta, iterations, methodTV, nonneg, half):
    dx, dy, dz, nd = _tv_dims(data)
    with torch.cuda.device(data.device):
        L.check(L.lib().tomo_pdtv(data.device.index, ptr(data), ptr(out), dx, dy, dz, nd, float(sigma), float(tau),
                                  float(lt), float(theta), int(iterations), int(bool(methodTV)), int(bool(nonneg)),
                                  int(bool(half)), stream_ptr(data)))
    return out


def roftv(data, out, lam, tau, iterations, half):
    dx, dy, dz, nd = _tv_dims(data)
    with torch.cuda.device(data.device):
        L.check(L.lib().tomo_roftv(data.device.index, ptr(data), ptr(out), dx, dy, dz, nd, float(lam), float(tau),
                                   int(iterations), int(bool(half)), stream_ptr(data)))
    return out


_variant_state = threading.local()   # mirror of the library's per-thread switches, per flavour: lets `variant()` restore


def set_variant(kernel: str, variant: int):
    L.check(L.lib().tomo_set_variant(kernel.encode(), int(variant)))
    state = getattr(_variant_state, "v", None)
    if state is None:
        state = _variant_state.v = {}
    state[(L.flavour(), kernel)] = int(variant)


def get_variant(kernel: str) -> int:
    """The variant this thread last selected for `kernel` in the current library flavour (0 = default)."""
    return getattr(_variant_state, "v", {}).get((L.flavour(), kernel), 0)


class _DeviceBytes:
    """Zero-copy hand-over of library-owned device memory to torch (``torch.as_tensor`` reads ``__cuda_array_interface__``)."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False),
                                         "version": 2, "strides": None}


ARRAY_SKEW = 69888   # bytes between consecutive work arrays of a placed block (what the library uses between its own)


def placed_empty(specs, device, slot: int = 0):
    """Uninitialised work arrays inside ONE placed scratch block of the library (include/tomo_mi355x.h,
    tomo_placed_scratch): ``specs`` is a list of (shape, dtype); returns (one tensor per entry, 256-byte aligned and
    ``ARRAY_SKEW`` apart; a lease token).  The block belongs to the library (grow-only per (device, stream, slot), freed
    by ``tomo_release_scratch``): the tensors are views for the duration of ONE driver call, not allocations to keep, and
    a second ``placed_empty`` on the same (device, stream, slot) supersedes them -- ``lease_is_current(token)`` tells.
    Slots in use: 0 = PD_TV slab driver, 1 = ROF_TV slab driver (tomobar_amd/slab.py)."""
    device = torch.device(device)
    sizes, total = [], 0
    for shape, dtype in specs:
        nb = int(np.prod(shape)) * torch.empty((), dtype=dtype).element_size()
        sizes.append((total, nb))
        total += (nb + 255) // 256 * 256 + ARRAY_SKEW
    out = C.c_void_p(0)
    with torch.cuda.device(device):
        stream_id = torch.cuda.current_stream(device).cuda_stream
        L.check(L.lib().tomo_placed_scratch(device.index or 0, int(slot), total, C.c_void_p(stream_id), C.byref(out)))
        block = torch.as_tensor(_DeviceBytes(out.value, total), device=device)
    key = (L.flavour(), device.index or 0, stream_id, int(slot))
    with _lease_lock:
        _lease_generation[key] = _lease_generation.get(key, 0) + 1
        lease = (key, _lease_generation[key])
    arrays = [block[o:o + nb].view(dtype).view(tuple(shape)) for (o, nb), (shape, dtype) in zip(sizes, specs)]
    return arrays, lease


_lease_lock = threading.Lock()
_lease_generation = {}   # (flavour, device, stream, slot) -> how many times placed_empty handed the slot's block out


def lease_is_current(lease) -> bool:
    """True while nobody else has taken the same (device, stream, slot) block since ``placed_empty`` returned `lease`: the
    arrays of an older lease may have been overwritten -- or freed, if the newer request was larger -- and must not be
    read any more (the slab drivers check this before they copy their result out)."""
    key, gen = lease
    with _lease_lock:
        return _lease_generation.get(key) == gen


def reserve_tv_scratch(shape, device, method: str = "PD_TV", half: bool = False):
    """Allocate and place the library's TV scratch arena for volumes of `shape` on the current stream of `device` now
    (set-up time) instead of inside the first proximal step (include/tomo_mi355x.h, tomo_reserve_scratch)."""
    device = torch.device(device)
    nd = len(shape)
    dz, dy, dx = (1, *shape) if nd == 2 else shape
    lib = L.lib()
    nbytes = lib.tomo_pdtv_scratch_bytes(dx, dy, dz, nd, int(bool(half))) if method == "PD_TV" else lib.tomo_roftv_scratch_bytes(dx, dy, dz, nd)
    with torch.cuda.device(device):
        L.check(lib.tomo_reserve_scratch(device.index or 0, nbytes, C.c_void_p(torch.cuda.current_stream(device).cuda_stream)))


def set_placement_tries(tries: int):
    """Candidate allocations the library scores when it places a TV scratch arena of >= 1 GiB (include/tomo_mi355x.h,
    tomo_set_placement_tries; default 8 or TOMO_MI355X_PLACE_TRIES, at most 10; 1 = plain hipMalloc)."""
    L.check(L.lib().tomo_set_placement_tries(int(tries)))


def placement_tries() -> int:
    return int(L.lib().tomo_placement_tries())


def placement_last():
    """The library's most recent arena placement search: {"bytes", "chosen", "scores_GBps", "fast"} or None if none ran
    yet.  "fast": the kept block beat an earlier candidate by the 8 % that separates the fast class of blocks from the
    slow one (False: the tries ran out first -- either no candidate was in the fast class: expect the TV launches 4-10 % slower, or
    all of them were and none stood out: compare the scores, fast blocks of a 34 GB arena score >= 5.23 TB/s)."""
    nbytes, chosen = C.c_size_t(0), C.c_int(-1)
    scores = (C.c_double * 16)()
    n = L.lib().tomo_placement_last(C.byref(nbytes), C.byref(chosen), scores, 16)
    if n <= 0:
        return None
    return {"bytes": int(nbytes.value), "chosen": int(chosen.value), "scores_GBps": [round(float(scores[i]), 1) for i in range(n)],
            "fast": bool(L.lib().tomo_placement_last_fast() == 1)}


@contextlib.contextmanager
def variant(kernel: str, value: int):
    """`with ops.variant("pdtv", 22): ...` -- select a kernel variant for a block and restore what was selected before."""
    prev = get_variant(kernel)
    set_variant(kernel, value)
    try:
        yield
    finally:
        set_variant(kernel, prev)
