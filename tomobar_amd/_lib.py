"""ctypes binding of ``libtomo_mi355x.so`` (the C-ABI declared in ``include/tomo_mi355x.h``).

There is deliberately no fallback: if the shared library is missing or no gfx950 device is
visible, every operator raises.  Build the library with ``python -c "import __graft_entry__ as g; g.build()"``
or ``make -C tomobar_amd/csrc``.
"""

from __future__ import annotations

import contextvars
import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libtomo_mi355x.so")

OK, E_INVALID, E_RUNTIME, E_NOMEM, E_NODEVICE = 0, 1, 2, 3, 4
ABI_VERSION = 6  # TOMO_ABI_VERSION of include/tomo_mi355x.h (tests/test_host_logic.py keeps the two in step)
FLAG_LERP8 = 1
FID = {"LS": 0, "PWLS": 1, "KL": 2, "RATIO": 3}
ROBUST = {None: 0, "huber": 1, "studentst": 2}   # TOMO_ROBUST_* of include/tomo_mi355x.h
RESIDUAL_LAYOUT = {"planar": 0, "zquad": 1}   # TOMO_RESIDUAL_* of include/tomo_mi355x.h


class AngleRecord(C.Structure):
    _fields_ = [("cs", C.c_float), ("sn", C.c_float), ("cor", C.c_float), ("slope", C.c_float),
                ("inv", C.c_float), ("scale", C.c_float), ("dirx", C.c_int32), ("src", C.c_int32)]


_vp, _i, _f, _d, _sz = C.c_void_p, C.c_int, C.c_float, C.c_double, C.c_size_t

# name -> (restype, argtypes); kept in one table so tests can check it against the header
SIGNATURES = {
    "tomo_abi_version": (_i, []),
    "tomo_build_flavour": (C.c_char_p, []),
    "tomo_last_error": (C.c_char_p, []),
    "tomo_device_count": (_i, [C.POINTER(_i)]),
    "tomo_ctx_create": (_i, [_i, _i, _i, _i, _i, C.POINTER(_d), C.POINTER(_d), _i, _i, C.c_uint, C.POINTER(_vp)]),
    "tomo_ctx_destroy": (_i, [_vp]),
    "tomo_ctx_os_number": (_i, [_vp]),
    "tomo_ctx_num_bins": (_i, [_vp]),
    "tomo_ctx_newind_table": (_i, [_vp, C.POINTER(C.c_int64)]),
    "tomo_ctx_subset_size": (_i, [_vp, _i]),
    "tomo_ctx_angle_table": (_i, [_vp, _i, C.POINTER(AngleRecord), _i]),
    "tomo_ctx_release_scratch": (_i, [_vp]),
    "tomo_ctx_kernel_path": (C.c_char_p, [_vp, C.c_char_p]),
    "tomo_fp3d": (_i, [_vp, _i, _vp, _vp, _vp]),
    "tomo_bp3d": (_i, [_vp, _i, _vp, _vp, _vp]),
    "tomo_fp3d_residual": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _vp, _vp]),
    "tomo_fp3d_residual_robust": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _f, _vp, _vp]),
    "tomo_sino_robust": (_i, [_vp, _sz, _i, _f, _vp]),
    "tomo_fp3d_residual_ring": (_i, [_vp, _i, _vp, _vp, _vp, _f, _vp, _vp]),
    "tomo_ring_gh_reduce": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _f, _vp, _vp]),
    "tomo_swls_apply": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _f, _vp]),
    "tomo_ring_gh_update": (_i, [_vp, _vp, _vp, _f, _f, _sz, _vp]),
    "tomo_shift_rows": (_i, [_vp, _vp, _i, _i, _i, _vp, _f, _vp]),
    "tomo_sino_add_ring": (_i, [_vp, _vp, _f, _i, _i, _i, _vp]),
    "tomo_sino_residual": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "tomo_bp3d_fista": (_i, [_vp, _i, _vp, _vp, _vp, _f, _i, _vp]),
    "tomo_bp3d_fista_momentum": (_i, [_vp, _i, _vp, _vp, _vp, _f, _f, _i, _vp]),
    "tomo_bp3d_admm": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _f, _f, _i, _f, _f, _i, _vp]),
    "tomo_ctx_set_residual_layout": (_i, [_vp, _i]),
    "tomo_ctx_residual_layout": (_i, [_vp]),
    "tomo_ctx_residual_elems": (_sz, [_vp, _i]),
    "tomo_momentum": (_i, [_vp, _vp, _vp, _f, _sz, _vp]),
    "tomo_momentum_transposed": (_i, [_vp, _vp, _vp, _vp, _f, _vp]),
    "tomo_ctx_invalidate": (_i, [_vp]),
    "tomo_admm_dual": (_i, [_vp, _vp, _vp, _sz, _vp]),
    "tomo_axpby": (_i, [_f, _vp, _f, _vp, _sz, _vp]),
    "tomo_scale": (_i, [_f, _vp, _vp, _sz, _vp]),
    "tomo_clamp_min": (_i, [_vp, _f, _sz, _vp]),
    "tomo_mul": (_i, [_vp, _vp, _sz, _vp]),
    "tomo_recip_safe": (_i, [_vp, _vp, _sz, _vp]),
    "tomo_fill": (_i, [_vp, _f, _sz, _vp]),
    "tomo_norm2": (_i, [_vp, _sz, C.POINTER(_d), _vp]),
    "tomo_dot": (_i, [_vp, _vp, _sz, C.POINTER(_d), _vp]),
    "tomo_max": (_i, [_vp, _sz, C.POINTER(_f), _vp]),
    "tomo_pwls_weights": (_i, [_vp, _vp, _sz, _vp]),
    "tomo_pwls_max": (_i, [_vp, _sz, C.POINTER(_f), _vp]),
    "tomo_pwls_weights_scaled": (_i, [_vp, _vp, _sz, _f, _vp]),
    "tomo_diag_stream": (_i, [C.POINTER(_vp), _i, C.POINTER(_vp), _i, _sz, _i, _i, _vp]),
    "tomo_pad_edge": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "tomo_crop_center": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "tomo_circ_mask": (_i, [_vp, _i, _i, _d, _vp]),
    "tomo_permute3": (_i, [_vp, _vp, _i, _i, _i, C.c_int64, C.c_int64, C.c_int64, _vp]),
    "tomo_pdtv": (_i, [_i, _vp, _vp, _i, _i, _i, _i, _f, _f, _f, _f, _i, _i, _i, _i, _vp]),
    "tomo_roftv": (_i, [_i, _vp, _vp, _i, _i, _i, _i, _f, _f, _i, _i, _vp]),
    "tomo_pdtv_scratch_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "tomo_roftv_scratch_bytes": (_sz, [_i, _i, _i, _i]),
    "tomo_release_scratch": (_i, [_i]),
    "tomo_reserve_scratch": (_i, [_i, _sz, _vp]),
    "tomo_set_placement_tries": (_i, [_i]),
    "tomo_placement_tries": (_i, []),
    "tomo_placement_last_fast": (_i, []),
    "tomo_placed_scratch": (_i, [_i, _i, _sz, _vp, C.POINTER(_vp)]),
    "tomo_placement_last": (_i, [C.POINTER(C.c_size_t), C.POINTER(_i), C.POINTER(C.c_double), _i]),
    "tomo_pdtv_iter_slab": (_i, [_i, _vp, _vp, _vp, C.POINTER(_vp), C.POINTER(_vp), _i, _i, _i, _i, _i,
                                 _f, _f, _f, _f, _i, _i, _i, _vp]),
    "tomo_pdtv_pair_slab": (_i, [_i, _vp, _vp, _vp, C.POINTER(_vp), C.POINTER(_vp), _i, _i, _i, _i, _i,
                                 _f, _f, _f, _f, _i, _i, _i, _vp]),
    "tomo_pdtv_pair_slab_range": (_i, [_i, _vp, _vp, _vp, C.POINTER(_vp), C.POINTER(_vp), _i, _i, _i, _i, _i, _i, _i,
                                       _f, _f, _f, _f, _i, _i, _i, _vp]),
    "tomo_pdtv_multi_slab_range": (_i, [_i, _vp, _vp, _vp, C.POINTER(_vp), C.POINTER(_vp), _i, _i, _i, _i, _i, _i, _i, _i,
                                        _f, _f, _f, _f, _i, _i, _i, _vp]),
    "tomo_roftv_iter_slab": (_i, [_i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _f, _i, _vp]),
    "tomo_roftv_iter_slab_range": (_i, [_i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _f, _i, _vp]),
    "tomo_halo_staging_bytes": (_sz, [C.POINTER(_sz), _i]),
    "tomo_halo_pack": (_i, [C.POINTER(_vp), C.POINTER(_sz), _i, _vp, _vp]),
    "tomo_halo_unpack": (_i, [_vp, C.POINTER(_vp), C.POINTER(_sz), _i, _vp]),
    "tomo_pdtv_iters_per_launch": (_i, [_i]),
    "tomo_fbp_filter": (_i, [_i, _vp, _sz, _i, _f, _f, _vp]),
    "tomo_fourier_inv": (_i, [_i, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _i, _f, _i, _vp]),
    "tomo_host_bp2d": (_i, [_vp, _vp, _i, _i, _i, C.POINTER(_d), _d]),
    "tomo_host_fp2d": (_i, [_vp, _vp, _i, _i, _i, C.POINTER(_d), _d]),
    "tomo_set_variant": (_i, [C.c_char_p, _i]),
    "tomo_profile_enable": (_i, [_i]),
    "tomo_profile_read": (_i, [C.c_char_p, C.POINTER(C.c_longlong), C.POINTER(_d)]),
}

LIB_PATHS = {"shipped": LIB_PATH, "dev": os.path.join(_HERE, "libtomo_mi355x_dev.so")}
_handles = {}
_DEFAULT_FLAVOUR = "dev" if os.environ.get("TOMO_MI355X_FLAVOUR", "shipped") == "dev" else "shipped"
# the flavour in force is per host THREAD (and per asyncio task): one thread's `with use_flavour("dev")` never redirects
# another thread's calls to the other library and its arenas
_flavour_var = contextvars.ContextVar("tomo_mi355x_flavour", default=_DEFAULT_FLAVOUR)


class TomoRuntimeError(RuntimeError):
    pass


def _load(flavour: str):
    path = LIB_PATHS[flavour]
    if not os.path.exists(path):
        raise ImportError(
            f"{path} not found: the HIP extension is required (no CPU fallback). "
            f"Build it with `make -C tomobar_amd/csrc{' dev' if flavour == 'dev' else ''}` or `__graft_entry__.build()`.")
    handle = C.CDLL(path)
    missing = [name for name in SIGNATURES if not hasattr(handle, name)]
    if missing:
        raise ImportError(f"{path} is stale: it does not export {missing[:4]}{'...' if len(missing) > 4 else ''}; "
                          "rebuild it (make -C tomobar_amd/csrc all)")
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(handle, name)
        fn.restype = res
        fn.argtypes = args
    if handle.tomo_abi_version() != ABI_VERSION:
        raise ImportError(f"{os.path.basename(path)} has ABI version {handle.tomo_abi_version()}, this package binds "
                          f"version {ABI_VERSION}: rebuild it (make -C tomobar_amd/csrc all)")
    if handle.tomo_build_flavour().decode() != flavour:
        raise ImportError(f"{path} reports flavour {handle.tomo_build_flavour().decode()!r}, expected {flavour!r}")
    return handle


_load_lock = threading.Lock()


def lib():
    """The loaded shared library of the current flavour (loaded once).  Raises ImportError with build instructions if it
    is absent.  The product always runs "shipped" (libtomo_mi355x.so); `with use_flavour("dev")` -- or TOMO_MI355X_FLAVOUR=dev
    in the environment -- points the calling thread at libtomo_mi355x_dev.so, the build that also carries the independent
    kernel implementations and measurement switches tests/ and tools/ compare against."""
    name = _flavour_var.get()
    h = _handles.get(name)
    if h is None:
        with _load_lock:
            h = _handles.get(name)
            if h is None:
                h = _handles[name] = _load(name)
    return h


def flavour() -> str:
    return _flavour_var.get()


class use_flavour:
    """Context manager: make `lib()` return the given flavour for the calling thread inside the `with` block (the switch
    happens in ``__enter__``, not at construction; ``enter()`` / ``exit()`` are the same for fixtures that cannot use
    `with`).  The two libraries are independent (their own contexts, scratch arenas and variant switches): objects created
    under one flavour keep calling it (HipTools3D stores its handle), so flavours can be mixed in one process as long as
    native handles are not passed across."""

    def __init__(self, name: str):
        if name not in LIB_PATHS:
            raise ValueError(f"unknown library flavour {name!r}")
        self.name = name
        self._token = None

    def __enter__(self):
        self._token = _flavour_var.set(self.name)
        return lib()

    def __exit__(self, *exc):
        if self._token is not None:
            _flavour_var.reset(self._token)
            self._token = None
        return False

    enter, exit = __enter__, __exit__


def check(rc: int, handle=None):
    """Map a C-ABI status to the exception type the reference raises for the same condition.  `handle`: the library the
    call went to (default: the current flavour)."""
    if rc == OK:
        return
    msg = (handle or lib()).tomo_last_error().decode("utf-8", "replace")
    if rc == E_INVALID:
        raise ValueError(msg)
    if rc == E_NOMEM:
        raise MemoryError(msg)
    raise TomoRuntimeError(msg)
