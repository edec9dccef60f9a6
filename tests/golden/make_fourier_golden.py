"""Generate tests/golden/fourier_golden.npz -- run in the build container only (needs /root/reference).

Runs the REFERENCE's ``RecToolsDIRCuPy.FOURIER_INV`` (tomobar/methodsDIR_CuPy.py:152-447 and the helpers :449-989,
tomobar/fourier.py:81-159) unmodified.  What the image lacks is supplied as plumbing, exactly as in
make_outer_golden.py: ``cupy`` forwards to numpy, ``cupyx.scipy.fft`` to ``scipy.fft``, ``astra`` is a geometry-dict
stub (the Fourier method never calls a projector), and the RawModule that ``load_cuda_module("fft_us_kernels")`` asks
for is served by the reference's own ``fft_us_kernels.cu`` executed on the host (oracle/ref_fft, built by
``make -C oracle ref``).  Fixtures: inputs + the reference's reconstruction for a handful of small geometries.

    cd /root/repo && make -C oracle ref && python tests/golden/make_fourier_golden.py
"""
import ctypes as C
import os
import sys
import types

import numpy as np
import scipy.fft as sfft

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import make_outer_golden as MOG  # noqa: E402  (cupy / astra plumbing shims)


class NdGet(np.ndarray):
    """numpy array with CuPy's ``.get()``"""

    def get(self):
        return np.asarray(self)


def _as(a):
    return np.asarray(a).view(NdGet)


class HostKernel:
    def __init__(self, lib, name):
        self.fn, self.name = getattr(lib, "ref_" + name), name

    def __call__(self, grid, block, args):
        g = (C.c_int * 3)(*[int(v) for v in grid])
        b = (C.c_int * 3)(*[int(v) for v in block])
        conv = []
        for a in args:
            if isinstance(a, np.ndarray):
                assert a.flags["C_CONTIGUOUS"], self.name
                conv.append(C.c_void_p(a.ctypes.data))
            elif isinstance(a, (float, np.floating)):
                conv.append(C.c_float(float(a)))
            else:
                conv.append(C.c_int(int(a)))
        self.fn(g, b, *conv)


class HostModule:
    def __init__(self, **_):
        self.lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_fft.so"))

    def get_function(self, name):
        return HostKernel(self.lib, name)


def install():
    MOG.install_shims()
    cp = sys.modules["cupy"]
    cp.RawModule = HostModule
    cp.RawKernel = HostKernel
    cp.array = lambda a, dtype=None: _as(np.array(a, dtype=dtype))
    cp.sort = lambda a: _as(np.sort(a))
    cp.argsort = lambda a: _as(np.argsort(a, kind="stable"))
    cp.ndarray = np.ndarray
    # gather_kernel_partial atomically ADDS into an array the reference allocates with cp.empty (methodsDIR_CuPy.py:662-666);
    # it relies on fresh device-pool memory being zero.  Make that explicit so the fixture is reproducible.
    cp.empty = lambda shape, dtype=np.float32: np.zeros(shape, dtype=dtype)
    cupyx = types.ModuleType("cupyx")
    cx_scipy = types.ModuleType("cupyx.scipy")
    cx_fft = types.ModuleType("cupyx.scipy.fft")
    for name in ("fft", "ifft2", "rfftfreq", "rfft", "irfft"):
        setattr(cx_fft, name, getattr(sfft, name))
    cx_fftpack = types.ModuleType("cupyx.scipy.fftpack")
    cx_fftpack.get_fft_plan = lambda *a, **k: None
    cx_scipy.fft, cx_scipy.fftpack = cx_fft, cx_fftpack
    cupyx.scipy = cx_scipy
    sys.modules.update({"cupyx": cupyx, "cupyx.scipy": cx_scipy, "cupyx.scipy.fft": cx_fft,
                        "cupyx.scipy.fftpack": cx_fftpack})


def phantom_sino(nz, nproj, n, angles, seed):
    """analytic parallel projections of a few ellipses + a little noise, [detY, angles, detX] float32"""
    rng = np.random.default_rng(seed)
    s = (np.arange(n) - (n - 1) / 2) / (n / 2)
    out = np.zeros((nz, nproj, n), np.float32)
    for k in range(nz):
        for (a, b, x0, y0, rho) in ((0.7, 0.5, 0.0, 0.05 * k, 1.0), (0.2, 0.3, 0.25, -0.2, 0.7), (0.15, 0.1, -0.3, 0.3, -0.4)):
            for j, th in enumerate(angles):
                c, sn = np.cos(th), np.sin(th)
                t = s - (x0 * c + y0 * sn)
                aa = (a * c) ** 2 + (b * sn) ** 2
                out[k, j] += (rho * 2 * a * b * np.sqrt(np.maximum(aa - t * t, 0)) / aa).astype(np.float32)
    out += 0.01 * rng.standard_normal(out.shape).astype(np.float32)
    return out


CASES = [
    # name, nz, nproj, data_n, recon_size, cor, angle span, kwargs
    ("default_even", 4, 40, 32, 32, 0.0, np.pi, {}),
    ("odd_sizes_hann_mask", 5, 37, 33, 30, 0.3, np.pi, {"filter_type": "hann", "cutoff_freq": 0.8, "recon_mask_radius": 0.9}),
    ("scatter_small_center", 2, 30, 40, 40, -0.25, np.pi, {"center_size": 64, "filter_type": "ramp"}),
    ("center_plus_partial", 2, 45, 128, 100, 0.0, np.pi, {"center_size": 192}),
    ("center_whole_grid", 3, 60, 100, 100, 0.4, np.pi, {}),
    ("center_unsorted_angles", 2, 48, 96, 90, 0.0, -np.pi, {"filter_type": "cosine"}),
    ("full_turn_padding", 4, 50, 36, 36, 0.0, 2 * np.pi, {"padding": 6, "filter_type": "parzen", "cutoff_freq": 0.6}),
    ("pow2_cropping_axes", 2, 33, 60, 48, 0.0, np.pi, {"power_of_2_cropping": True,
                                                       "data_axes_labels_order": ["angles", "detY", "detX"]}),
]


def main():
    install()
    from tomobar.methodsDIR_CuPy import RecToolsDIRCuPy
    out = {}
    for name, nz, nproj, dn, rs, cor, span, kw in CASES:
        angles = np.linspace(0, span, nproj, endpoint=False).astype(np.float64)
        sino = phantom_sino(nz, nproj, dn, angles, seed=len(name))
        rt = RecToolsDIRCuPy(DetectorsDimH=dn, DetectorsDimH_pad=0, DetectorsDimV=nz, CenterRotOffset=cor,
                             AnglesVec=angles, ObjSize=rs, device_projector=0)
        data = sino
        kwargs = dict(kw)
        if kwargs.get("data_axes_labels_order") == ["angles", "detY", "detX"]:
            data = np.ascontiguousarray(np.swapaxes(sino, 0, 1))
        rec = np.asarray(rt.FOURIER_INV(data.copy(), **kwargs), dtype=np.float32)
        print(name, rec.shape, float(np.abs(rec).max()), float(rec.mean()))
        out[name + "_sino"] = sino
        out[name + "_angles"] = angles
        out[name + "_rec"] = rec
    np.savez_compressed(os.path.join(HERE, "fourier_golden.npz"), **out)


if __name__ == "__main__":
    main()
