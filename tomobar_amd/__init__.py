"""tomobar_amd -- MI355X (gfx950) engine for ToMoBAR's ordered-subsets FISTA / ADMM hot path.

Drop-in surface (same names and argument meaning as the reference package ``tomobar``):

    from tomobar_amd.methodsIR_CuPy import RecToolsIRCuPy      # FISTA / ADMM / OSEM / SIRT / CGLS / Landweber
    from tomobar_amd.methodsDIR_CuPy import RecToolsDIRCuPy    # FORWPROJ / BACKPROJ / FBP
    from tomobar_amd.regularisersCuPy import PD_TV_cupy, ROF_TV_cupy

Arrays are float32 ``torch.Tensor`` on the GPU.  All arithmetic runs in hand-written HIP kernels of
``libtomo_mi355x.so`` (C-ABI: ``include/tomo_mi355x.h``); importing this package never builds or falls back to
anything: without the library or without a GPU the operators raise.
"""

__version__ = "0.1.0"

from . import _lib  # noqa: F401


def library_path() -> str:
    return _lib.LIB_PATH
