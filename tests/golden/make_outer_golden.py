"""Generate tests/golden/outer_golden.npz -- run in the build container only (needs /root/reference).

Imports the REFERENCE's Python driver (/root/reference/tomobar/methodsIR_CuPy.py with its dicts_check,
data_fidelities, suppTools glue) and runs its FISTA / ADMM / powermethod loops unmodified.  Two things the
image lacks are supplied so that the import works, and both are *array plumbing*, not algorithm:
  * ``cupy``  -> a module object that forwards to numpy (the reference only uses the numpy-compatible
                 subset on this path) plus the handful of CUDA-runtime names touched at import time;
  * ``astra`` -> geometry-dict constructors only.  The projector calls (direct_FP3D/direct_BP3D live in the
                 un-vendored astra-toolbox==2.4.*) are replaced at the ``Atools._forwproj*CuPy/_backproj*CuPy``
                 seam by the oracle's CPU projector pair (oracle/tomo_oracle.c), and the TV kernels are the
                 reference's own .cu sources executed on the host (oracle/ref_tv, fma build).
So these fixtures pin the OUTER-LOOP ALGEBRA of the reference (t-sequence, momentum, OS subset order and
trimming, PWLS/KL residuals, ADMM relaxation/dual placement, pad/crop/mask, axis swapping, 2D handling)
for a given projector; they do not pin ASTRA's element-wise output.

    cd /root/repo && make -C oracle ref && python tests/golden/make_outer_golden.py
"""
import ctypes as C
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import tomo_oracle as O  # noqa: E402

REF = "/root/reference"


# ----------------------------------------------------------------------------- plumbing shims
def install_shims():
    cp = types.ModuleType("cupy")

    def _getattr(name):
        return getattr(np, name)

    cp.__getattr__ = _getattr

    class _Pool:
        def free_all_blocks(self):
            pass

    cp._default_memory_pool = _Pool()
    cp.get_default_memory_pool = lambda: _Pool()
    cp.get_array_module = lambda *a: cp
    cp.asnumpy = np.asarray

    class _Dev:
        compute_capability = "00"

        def __init__(self, *_):
            pass

        def use(self):
            pass

    cuda = types.SimpleNamespace(Device=_Dev, runtime=types.SimpleNamespace(CUDARuntimeError=RuntimeError))
    cp.cuda = cuda
    rng = np.random.default_rng(7)
    cp.random = types.SimpleNamespace(randn=lambda *shape, dtype=np.float32: rng.standard_normal(shape).astype(dtype))
    cp.RawModule = None
    sys.modules["cupy"] = cp

    astra = types.ModuleType("astra")
    astra.create_vol_geom = lambda Y, X, Z=None: {"kind": "vol", "Y": Y, "X": X, "Z": Z}

    def create_proj_geom(kind, *args):
        if kind == "parallel3d_vec":
            rows, cols, vectors = args
            return {"kind": kind, "rows": rows, "cols": cols, "vectors": vectors}
        return {"kind": kind, "args": args}

    def geom_size(g):
        if g["kind"] == "vol":
            return (g["Z"], g["Y"], g["X"])
        return (g["rows"], g["vectors"].shape[0], g["cols"])

    astra.create_proj_geom = create_proj_geom
    astra.geom_size = geom_size
    astra.create_projector = lambda *a, **k: 0
    astra.data3d = types.SimpleNamespace(delete=lambda *_: None, link=lambda *_: 0)
    exp = types.ModuleType("astra.experimental")
    exp.direct_BP3D = exp.direct_FP3D = None
    pu = types.ModuleType("astra.pythonutils")
    pu.GPULink = None
    astra.experimental, astra.pythonutils = exp, pu
    sys.modules.update({"astra": astra, "astra.experimental": exp, "astra.pythonutils": pu})
    sys.path.insert(0, REF)


def ref_tv_lib():
    fp = C.POINTER(C.c_float)
    L = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_tv_fma.so"))
    L.ref_pdtv.argtypes = [fp, fp] + [C.c_int] * 4 + [C.c_float] * 4 + [C.c_int] * 4
    L.ref_roftv.argtypes = [fp, fp] + [C.c_int] * 4 + [C.c_float] * 2 + [C.c_int] * 2
    return L


def setup():
    """Install the plumbing shims, import the reference driver and return ``make(detH, pad, detV, cor, angles, objsize,
    os_number)``: a factory of REFERENCE RecToolsIRCuPy objects with the oracle projector plugged in at the Atools seam
    and the reference's own TV kernel sources (host-executed) behind PD_TV_cupy / ROF_TV_cupy."""
    install_shims()
    import tomobar.regularisersCuPy as reg_mod
    from tomobar.methodsIR_CuPy import RecToolsIRCuPy

    L = ref_tv_lib()

    def _dims(d):
        return (d.shape[1], d.shape[0], 1, 2) if d.ndim == 2 else (d.shape[2], d.shape[1], d.shape[0], 3)

    def PD(data, lam=1e-5, iterations=1000, methodTV=0, nonneg=0, lipschitz_const=8.0, gpu_id=0,
           half_precision=False):
        d, is2d, ax = O._squeeze_2d(data)
        d = np.ascontiguousarray(d)
        sg, tau, lt, th = O.pd_scalars(lam, lipschitz_const)
        out = np.empty_like(d)
        dx, dy, dz, nd = _dims(d)
        L.ref_pdtv(O._fptr(d), O._fptr(out), dx, dy, dz, nd, sg, tau, lt, th, iterations, int(methodTV),
                   int(nonneg), int(half_precision))
        return np.expand_dims(out, ax) if is2d else out

    def ROF(data, lam=1e-5, iterations=3000, tms=0.001, gpu_id=0, half_precision=False):
        d, is2d, ax = O._squeeze_2d(data)
        d = np.ascontiguousarray(d)
        out = np.empty_like(d)
        dx, dy, dz, nd = _dims(d)
        L.ref_roftv(O._fptr(d), O._fptr(out), dx, dy, dz, nd, np.float32(lam), np.float32(tms), iterations,
                    int(half_precision))
        return np.expand_dims(out, ax) if is2d else out

    reg_mod.PD_TV_cupy = PD
    reg_mod.ROF_TV_cupy = ROF

    def make(detH, pad, detV, cor, angles, objsize, os_number=None):
        R = RecToolsIRCuPy(detH, pad, detV, cor, angles, objsize, 0, os_number)
        nz = detV if detV else 1
        n = detH + 2 * pad if pad > 0 else objsize
        P = O.Projector(nz, n, detH + 2 * pad, angles, cor, os_number if os_number else 1)
        A = R.Atools
        A._forwprojCuPy = lambda x: P.fp(np.ascontiguousarray(x))
        A._backprojCuPy = lambda b: P.bp(np.ascontiguousarray(b))
        A._forwprojOSCuPy = lambda x, os_index: P.fp(np.ascontiguousarray(x), os_index)
        A._backprojOSCuPy = lambda b, os_index: P.bp(np.ascontiguousarray(b), os_index)
        return R

    return make


def main():
    make = setup()
    store = {}
    rng = np.random.default_rng(11)
    nz, n, na = 4, 24, 30
    angles = np.linspace(0, np.pi, na, endpoint=False)
    sino = O.shepp_logan_sino(n, nz, n, angles) / n  # ~O(1) values
    sino = (sino + 0.02 * rng.standard_normal(sino.shape)).astype(np.float32)
    store["angles"] = angles
    store["sino"] = sino

    def run(name, method, rt, data, alg, reg=None):
        out = getattr(rt, method)(dict(data), dict(alg), None if reg is None else dict(reg))
        store[name] = np.ascontiguousarray(out, dtype=np.float32)
        print(name, store[name].shape, float(store[name].min()), float(store[name].max()))

    d = {"projection_data": sino, "data_axes_labels_order": ["detY", "angles", "detX"]}
    # Lipschitz constants from the reference's own power method (random start, shim-seeded)
    store["L_full"] = np.float64(make(n, 0, nz, 0.0, angles, n).powermethod(dict(d)))
    store["L_os4"] = np.float64(make(n, 0, nz, 0.0, angles, n, 4).powermethod(dict(d)))
    store["L_os7"] = np.float64(make(n, 0, nz, 0.0, angles, n, 7).powermethod(dict(d)))
    Lf, L4, L7 = float(store["L_full"]), float(store["L_os4"]), float(store["L_os7"])
    print("L", Lf, L4, L7)

    run("fista_plain", "FISTA", make(n, 0, nz, 0.0, angles, n), d, {"iterations": 6, "lipschitz_const": Lf})
    run("fista_nonneg_mask", "FISTA", make(n, 0, nz, 0.0, angles, n), d,
        {"iterations": 5, "lipschitz_const": Lf, "nonnegativity": True, "recon_mask_radius": 0.85})
    run("fista_os4_pdtv", "FISTA", make(n, 0, nz, 0.0, angles, n, 4), d,
        {"iterations": 3, "lipschitz_const": L4, "nonnegativity": True},
        {"method": "PD_TV", "regul_param": 0.002, "iterations": 8})
    run("fista_os7_roftv", "FISTA", make(n, 0, nz, 0.0, angles, n, 7), d,
        {"iterations": 2, "lipschitz_const": L7},
        {"method": "ROF_TV", "regul_param": 0.002, "iterations": 8, "time_marching_step": 0.002})
    run("fista_os4_pdtv_half_aniso", "FISTA", make(n, 0, nz, 0.0, angles, n, 4), d,
        {"iterations": 2, "lipschitz_const": L4},
        {"method": "PD_TV", "regul_param": 0.002, "iterations": 6, "methodTV": 1, "half_precision": True})
    dp = dict(d, data_fidelity="PWLS")
    run("fista_pwls_os4", "FISTA", make(n, 0, nz, 0.0, angles, n, 4), dp, {"iterations": 3, "lipschitz_const": L4})
    raw = np.exp(-np.clip(sino, 0, None)).astype(np.float32)
    store["raw_kl"] = raw
    dk = {"projection_data": raw, "data_axes_labels_order": ["detY", "angles", "detX"], "data_fidelity": "KL"}
    x_kl = np.full((nz, n, n), 0.02, dtype=np.float32)  # KL needs Ax > 0: warm start (zero start divides by the 1e-8 clip)
    store["x0_kl"] = x_kl
    run("fista_kl", "FISTA", make(n, 0, nz, 0.0, angles, n), dk,
        {"iterations": 3, "lipschitz_const": Lf, "initialise": x_kl, "nonnegativity": True})
    # axis permutation + centre of rotation offset
    d_perm = {"projection_data": np.ascontiguousarray(np.swapaxes(sino, 0, 1)),
              "data_axes_labels_order": ["angles", "detY", "detX"]}
    run("fista_perm_cor", "FISTA", make(n, 0, nz, 1.5, angles, n), d_perm, {"iterations": 4, "lipschitz_const": Lf})
    # padded detector -> larger grid, crop back, no mask
    rt = make(n, 4, nz, 0.0, angles, n, 4)
    store["L_pad_os4"] = np.float64(rt.powermethod(dict(d)))
    run("fista_pad_os4", "FISTA", make(n, 4, nz, 0.0, angles, n, 4), d,
        {"iterations": 3, "lipschitz_const": float(store["L_pad_os4"]), "recon_mask_radius": 2.0})
    # 2D input
    d2 = {"projection_data": np.ascontiguousarray(sino[1]), "data_axes_labels_order": ["angles", "detX"]}
    run("fista_2d_os4_pdtv", "FISTA", make(n, 0, None, 0.0, angles, n, 4), d2,
        {"iterations": 3, "lipschitz_const": L4}, {"method": "PD_TV", "regul_param": 0.002, "iterations": 8})
    # warm start
    x0 = store["fista_plain"].copy()
    run("fista_warm", "FISTA", make(n, 0, nz, 0.0, angles, n), d,
        {"iterations": 2, "lipschitz_const": Lf, "initialise": x0})
    # ADMM: relaxation starts at iter_no > 1, dual update per outer iteration
    run("admm_plain", "ADMM", make(n, 0, nz, 0.0, angles, n), d, {"iterations": 5, "lipschitz_const": Lf})
    run("admm_pdtv", "ADMM", make(n, 0, nz, 0.0, angles, n), d,
        {"iterations": 4, "lipschitz_const": Lf, "ADMM_rho_const": 2.0, "ADMM_relax_par": 1.5, "nonnegativity": True},
        {"method": "PD_TV", "regul_param": 0.004, "iterations": 8})
    run("admm_os4_roftv", "ADMM", make(n, 0, nz, 0.0, angles, n, 4), d,
        {"iterations": 4, "lipschitz_const": L4},
        {"method": "ROF_TV", "regul_param": 0.004, "iterations": 8, "time_marching_step": 0.002})
    run("admm_os4_pwls", "ADMM", make(n, 0, nz, 0.0, angles, n, 4), dp,
        {"iterations": 3, "lipschitz_const": L4, "recon_mask_radius": 0.9})
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "outer_golden.npz"), **store)
    print("wrote", len(store), "arrays")


if __name__ == "__main__":
    main()
