"""Generate tests/golden/tv_golden.npz -- run in the build container only (needs /root/reference).

Executes the REFERENCE's own TV kernel sources
(/root/reference/tomobar/cuda_kernels/{primal_dual_for_total_variation,rudin_osher_fatemi_total_variation}.cu)
on the host through oracle/ref_tv (see oracle/ref_tv/cuda_host_exec.h) with the launch/iteration structure
of tomobar/regularisersCuPy.py, on small seeded inputs, and stores inputs + parameters + outputs.
Two builds are recorded: ``off`` (-ffp-contract=off) and ``fma`` (-ffp-contract=fast, what NVRTC's default
--fmad=true produces).  The committed .npz holds data only.

    cd /root/repo && make -C oracle ref && python tests/golden/make_tv_golden.py
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import tomo_oracle as O  # noqa: E402  (scalar set-up + squeeze helper only)

fp = C.POINTER(C.c_float)


def load(name):
    L = C.CDLL(os.path.join(ROOT, "oracle", "_ref", name))
    L.ref_pdtv.argtypes = [fp, fp] + [C.c_int] * 4 + [C.c_float] * 4 + [C.c_int] * 4
    L.ref_roftv.argtypes = [fp, fp] + [C.c_int] * 4 + [C.c_float] * 2 + [C.c_int] * 2
    return L


def dims(d):
    if d.ndim == 2:
        return d.shape[1], d.shape[0], 1, 2
    return d.shape[2], d.shape[1], d.shape[0], 3


def ref_pd(L, data, lam, iters, mtv, nn, lip, half):
    d, is2d, ax = O._squeeze_2d(data)
    d = np.ascontiguousarray(d)
    sg, tau, lt, th = O.pd_scalars(lam, lip)
    out = np.empty_like(d)
    dx, dy, dz, nd = dims(d)
    L.ref_pdtv(O._fptr(d), O._fptr(out), dx, dy, dz, nd, sg, tau, lt, th, iters, mtv, nn, half)
    return np.expand_dims(out, ax) if is2d else out


def ref_rof(L, data, lam, iters, tms, half):
    d, is2d, ax = O._squeeze_2d(data)
    d = np.ascontiguousarray(d)
    out = np.empty_like(d)
    dx, dy, dz, nd = dims(d)
    L.ref_roftv(O._fptr(d), O._fptr(out), dx, dy, dz, nd, np.float32(lam), np.float32(tms), iters, half)
    return np.expand_dims(out, ax) if is2d else out


def main():
    libs = {"off": load("libref_tv.so"), "fma": load("libref_tv_fma.so")}
    rng = np.random.default_rng(20260929)
    store = {}
    cases = []
    shapes = [(6, 9, 13), (1, 20, 17), (12, 1, 70), (10, 11, 1), (8, 8, 8), (3, 5, 131), (24, 19)]
    cid = 0
    for shape in shapes:
        base = rng.random(shape).astype(np.float32)
        step = (np.indices(shape)[-1] > shape[-1] // 2).astype(np.float32)
        x = (base * 0.3 + step).astype(np.float32)
        store[f"in_{len(cases)}"] = x
        in_id = len(cases)
        cases.append(shape)
        for half in (0, 1):
            for mtv in (0, 1):
                for nn in (0, 1):
                    xi = (x - 0.6).astype(np.float32) if nn else x
                    for k, L in libs.items():
                        store[f"pd_{cid}_{k}"] = ref_pd(L, xi, 0.04, 12, mtv, nn, 8.0, half)
                    store[f"pd_{cid}_meta"] = np.array([in_id, half, mtv, nn, 12, 0.04, 8.0], dtype=np.float64)
                    cid += 1
            for k, L in libs.items():
                store[f"rof_{cid}_{k}"] = ref_rof(L, x, 0.05, 12, 0.005, half)
            store[f"rof_{cid}_meta"] = np.array([in_id, half, 12, 0.05, 0.005], dtype=np.float64)
            cid += 1
    # a longer run (error growth over iterations) and the dicts.py default Lipschitz 12
    x = store["in_4"]
    for k, L in libs.items():
        store[f"pd_{cid}_{k}"] = ref_pd(L, (x - 0.6).astype(np.float32), 0.0005, 60, 0, 1, 12.0, 0)  # nn=1 -> shifted
    store[f"pd_{cid}_meta"] = np.array([4, 0, 0, 1, 60, 0.0005, 12.0], dtype=np.float64)
    cid += 1
    for k, L in libs.items():
        store[f"rof_{cid}_{k}"] = ref_rof(L, x, 0.02, 60, 0.001, 0)
    store[f"rof_{cid}_meta"] = np.array([4, 0, 60, 0.02, 0.001], dtype=np.float64)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "tv_golden.npz"), **store)
    print("wrote", len(store), "arrays")


if __name__ == "__main__":
    main()
