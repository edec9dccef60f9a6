"""Overhead of the slab (multi-GPU) PD_TV driver relative to the single-call operator, measured on ONE GPU with a
communicator that has no neighbours (so: the Python-level pair loop, the ghost-less PdSlab buffers, the final copy)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
from tomobar_amd.regularisersCuPy import PD_TV_cupy
from tomobar_amd.slab import pd_tv_slab


class NoNeighbours:
    has_lo = has_hi = False

    def exchange(self, *a):
        pass

    def exchange_start(self, *a):
        return []

    @staticmethod
    def exchange_wait(reqs):
        pass


N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
vol = torch.rand((N, N, N), device="cuda")
out = torch.empty_like(vol)


def timed(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); fn(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 2


a = timed(lambda: PD_TV_cupy(vol, 0.01, 30, 0, 1, 12.0, 0, False, out=out))
b = timed(lambda: pd_tv_slab(vol, NoNeighbours(), 0.01, 30, 0, 1, 12.0, False, out=out))
print(f"PD_TV 30 iterations {N}^3: single call {a:.1f} ms, slab driver {b:.1f} ms (+{(b / a - 1) * 100:.1f} %)")
