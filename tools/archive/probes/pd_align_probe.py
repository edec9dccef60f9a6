"""Does the PD_TV time depend on where the caller's Input / output arrays sit relative to the library's arena?
The volume is a view at different byte offsets into one larger allocation.  usage: python tools/archive/probes/pd_align_probe.py [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import statistics
import torch
from tomobar_amd.regularisersCuPy import PD_TV_cupy
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
IT = 30
V = N * N * N
PAD = 1 << 20   # floats of slack
big = torch.rand(V + PAD, device="cuda")
bigo = torch.empty(V + PAD, device="cuda")
print("base pointers: in %x  out %x" % (big.data_ptr(), bigo.data_ptr()), flush=True)
for off_b in (0, 4096, 16384, 34944, 69888, 139776, 262144, 524288, 1048576, 2097152 + 69888):
    off = off_b // 4
    vol = big[off:off + V].view(N, N, N)
    out = bigo[off:off + V].view(N, N, N)
    ts = []
    PD_TV_cupy(vol, 0.01, 3, 0, 1, 12.0, 0, False, out=out)
    for _ in range(3):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); PD_TV_cupy(vol, 0.01, IT, 0, 1, 12.0, 0, False, out=out); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / IT)
    print(f"offset {off_b:8d} B: median {statistics.median(ts):6.3f} ms/iter", flush=True)
