"""Achievable HBM rate for an (nin reads, nout writes) streaming mix: calibrates roofline expectations."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
from tomobar_amd import _lib
lib = _lib.lib()
lib.tomo_diag_stream.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_void_p), C.c_int, C.c_size_t, C.c_int, C.c_int, C.c_void_p]
n = 1024 ** 3
skew = int(os.environ.get("SKEW", "0"))
def alloc(k):
    big = torch.empty(n * k + k * skew // 4 + 1024, dtype=torch.float32, device="cuda")
    big.normal_()
    return big, [big[i * (n + skew // 4): i * (n + skew // 4) + n] for i in range(k)]
for nin, nout in ((2, 1), (1, 1), (5, 4), (3, 1), (4, 4)):
    bi, ins = alloc(nin)
    bo, outs = alloc(nout)
    pi = (C.c_void_p * 8)(*[t.data_ptr() for t in ins])
    po = (C.c_void_p * 8)(*[t.data_ptr() for t in outs])
    for vec in (4, 1):
        for grid in (2048, 8192, 0x7fff):
            st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            for _ in range(2):
                lib.tomo_diag_stream(pi, nin, po, nout, n, vec, grid, st)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                lib.tomo_diag_stream(pi, nin, po, nout, n, vec, grid, st)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 3
            print(f"reads={nin} writes={nout} vec={vec} grid={grid:6d} skew={skew}: {ms:7.3f} ms  {(nin+nout)*4*n/ms/1e6:7.1f} GB/s")
    del bi, bo, ins, outs
