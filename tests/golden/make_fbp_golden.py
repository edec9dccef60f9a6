"""Generate tests/golden/fbp_golden.npz -- run in the build container only (needs /root/reference).

Imports the reference's numpy filter ``tomobar.methodsDIR._filtersinc2D`` (methodsDIR.py:295-320; the same sinc-ramp
formula as the CuPy path's generate_filtersinc kernel, with a fixed cut-off a = 1.1 and the 1/angles factor) through the
same plumbing shims as make_outer_golden.py and applies it to seeded sinograms.  Data only is stored.

    python tests/golden/make_fbp_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_outer_golden import install_shims  # noqa: E402


def main():
    install_shims()
    from tomobar.methodsDIR import _filtersinc2D
    rng = np.random.default_rng(2026)
    store = {}
    for i, (na, nu) in enumerate(((24, 64), (17, 45), (30, 128))):
        s = rng.random((na, nu)).astype(np.float32)
        store[f"sino_{i}"] = s
        store[f"filt_{i}"] = np.asarray(_filtersinc2D(s), dtype=np.float32)
    np.savez_compressed(os.path.join(HERE, "fbp_golden.npz"), **store)
    print("wrote", sorted(store))


if __name__ == "__main__":
    main()
