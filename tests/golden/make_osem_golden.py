"""Generate tests/golden/osem_golden.npz -- run in the build container only (needs /root/reference).

The REFERENCE's own ``RecToolsIRCuPy.OSEM`` (/root/reference/tomobar/methodsIR_CuPy.py:587-667) is run unmodified on
emission-like data through the same seam as make_outer_golden.py (numpy-forwarding ``cupy``, geometry-only ``astra``,
oracle projector at ``Atools._forwproj*/_backproj*``, the reference's TV kernel sources executed on the host).  Pins the
multiplicative update ``x *= A_s^T(b_s / clip(A_s x, 1e-8)) * clip(A_0^T 1, 1e-8)`` (:648-654), the subset order and
trim, the ones start vector and the prox placement.

    make -C oracle ref && python tests/golden/make_osem_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_outer_golden as G  # noqa: E402
from oracle import tomo_oracle as O  # noqa: E402


def main():
    make = G.setup()
    nz, n, na = 4, 24, 30
    angles = np.linspace(0, np.pi, na, endpoint=False)
    rng = np.random.default_rng(23)
    # emission-like data: non-negative line integrals of the phantom with Poisson noise, scaled so that the reference's
    # (multiplying, :654) normalisation keeps the iterates in float32 range for a few iterations
    clean = np.clip(O.shepp_logan_sino(n, nz, n, angles), 0, None)
    counts = rng.poisson(clean * 4.0).astype(np.float32) / 4.0
    sino = (counts * np.float32(1e-3)).astype(np.float32)
    store = {"angles": angles, "sino": sino}

    def run(name, rt, alg, reg=None):
        d = {"projection_data": sino.copy(), "data_axes_labels_order": ["detY", "angles", "detX"]}
        out = rt.OSEM(d, dict(alg), None if reg is None else dict(reg))
        store[name] = np.ascontiguousarray(out, dtype=np.float32)
        print(name, store[name].shape, float(store[name].min()), float(store[name].max()))

    run("mlem", make(n, 0, nz, 0.0, angles, n), {"iterations": 3})
    run("osem_os4", make(n, 0, nz, 0.0, angles, n, 4), {"iterations": 2})
    run("osem_os7_mask", make(n, 0, nz, 0.0, angles, n, 7), {"iterations": 1, "recon_mask_radius": 0.9})
    run("osem_os4_pdtv", make(n, 0, nz, 0.0, angles, n, 4), {"iterations": 2, "nonnegativity": True},
        {"method": "PD_TV", "regul_param": 0.002, "iterations": 6})
    run("mlem_roftv", make(n, 0, nz, 0.0, angles, n), {"iterations": 2},
        {"method": "ROF_TV", "regul_param": 0.002, "iterations": 5, "time_marching_step": 0.002})
    np.savez_compressed(os.path.join(HERE, "osem_golden.npz"), **store)
    print("wrote", len(store), "arrays")


if __name__ == "__main__":
    main()
