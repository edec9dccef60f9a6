"""Generate tests/golden/cfg1_golden.npz -- run in the build container only (needs /root/reference).

BASELINE configs[0]: 2D phantom 256^2, 180 angles, FBP through the REFERENCE's ``RecToolsDIR(..., device_projector="cpu")``
(tomobar/methodsDIR.py:121-175: edge padding, ``_filtersinc2D`` :295-320, ``Atools._backproj``, circular mask), imported
through the plumbing shims of make_outer_golden.py.  ASTRA's CPU `line` back projection is not available, so the seam
``Atools._backproj`` is served by the oracle's voxel-driven back projector (one slice); what the fixture pins is the
reference's filter, scaling, padding and masking around it.

    python tests/golden/make_cfg1_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_outer_golden as G  # noqa: E402
from oracle import tomo_oracle as O  # noqa: E402


def main():
    G.install_shims()
    import astra  # the geometry stub: the 2D CPU branch also asks for these
    astra.create_projector = lambda *a, **k: 0
    from tomobar.methodsDIR import RecToolsDIR
    n, na = 256, 180
    angles = np.linspace(0, np.pi, na, endpoint=False)
    sino = (O.shepp_logan_sino(n, 1, n, angles)[0] / n).astype(np.float32)          # [angles, detX]
    sino = (sino + 0.01 * np.random.default_rng(3).standard_normal(sino.shape)).astype(np.float32)
    store = {"angles": angles, "sino": sino}
    for name, pad, radius in (("fbp", 0, 0.95), ("fbp_pad", 16, None)):
        rt = RecToolsDIR(n, pad, None, 0.0, angles, n, device_projector="cpu")
        P = O.Projector(1, n, n + 2 * pad, angles, 0.0, 1)
        rt.Atools._backproj = lambda s, P=P: P.bp(np.ascontiguousarray(s, dtype=np.float32)[None])[0]
        kw = {} if radius is None else {"recon_mask_radius": radius}
        store[name] = np.asarray(rt.FBP(sino.copy(), **kw), dtype=np.float32)
        print(name, store[name].shape, float(store[name].min()), float(store[name].max()))
    np.savez_compressed(os.path.join(HERE, "cfg1_golden.npz"), **store)


if __name__ == "__main__":
    main()
