"""PD_TV on 2D images: the fused rows-in-registers kernel (pd_rows2d.inl, three iterations per launch).  28 B/pixel is what
ONE iteration moves when launched on its own (read Input, U, P1, P2; write U, P1, P2): the fused launch moves it once per
three.  usage: python tools/tv2d_bench.py [N ...]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tomobar_amd.regularisersCuPy import PD_TV_cupy

for n in [int(v) for v in sys.argv[1:]] or [1024, 2048, 4096, 8192]:
    img = torch.rand((n, n), device="cuda")
    out = torch.empty_like(img)
    for half in (False, True):
        PD_TV_cupy(img, 0.01, 30, 0, 1, 12.0, 0, half, out=out); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            PD_TV_cupy(img, 0.01, 30, 0, 1, 12.0, 0, half, out=out)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5 / 30
        b = 20 if half else 28
        print(f"{n}^2 {'f16' if half else 'f32'} duals: {ms*1e3:8.1f} us per iteration (30-iteration prox, 10 launches); one iteration alone would "
              f"move {b} B/pixel = {b*n*n/1e6:.0f} MB -> {b*n*n/ms/1e6:7.1f} GB/s per-iteration-equivalent", flush=True)
