"""Bit-equality of PD_TV kernel variants that must compute the same roundings: python tools/archive/probes/pd_variant_equal.py A B [A B ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
os.environ.setdefault("TOMO_MI355X_FLAVOUR", "dev")  # A/B variants and measurement switches live in libtomo_mi355x_dev.so
from tomobar_amd import ops
from tomobar_amd.regularisersCuPy import PD_TV_cupy
pairs = [(int(sys.argv[i]), int(sys.argv[i + 1])) for i in range(1, len(sys.argv) - 1, 2)]
g = torch.Generator(device="cuda").manual_seed(1)
bad = 0
for shape in [(40, 200, 300), (7, 64, 130), (96, 257, 515), (300, 33, 70), (5, 1024, 1024), (129, 120, 121)]:
    vol = torch.rand(shape, device="cuda", generator=g) * 2 - 0.5
    for half in (False, True):
        for nonneg in (0, 1):
            for mtv in (0, 1):
                for iters in (3, 9, 7):
                    for va, vb in pairs:
                        outs = []
                        for v in (va, vb):
                            ops.set_variant("pdtv", v)
                            outs.append(PD_TV_cupy(vol, 0.05, iters, mtv, nonneg, 12.0, 0, half).clone())
                        if not torch.equal(outs[0], outs[1]):
                            bad += 1
                            d = (outs[0] - outs[1]).abs().max().item()
                            print(f"MISMATCH v{va} vs v{vb} shape {shape} half {half} nonneg {nonneg} mtv {mtv} iters {iters}: max diff {d:.3e}")
ops.set_variant("pdtv", 0)
print("mismatches:", bad)
