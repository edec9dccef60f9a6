"""Which placement matters to a PD_TV launch: the scratch arena's or the Input / output arrays'?  S arenas (one per torch
stream, plain hipMalloc: TOMO_MI355X_PLACE_TRIES=1) x S (Input, output) pairs allocated in between; the 30-iteration prox is
timed for every combination.  usage: TOMO_MI355X_PLACE_TRIES=1 python tools/archive/probes/pd_input_arena_matrix.py [N] [S]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
from tomobar_amd.regularisersCuPy import PD_TV_cupy
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
S = int(sys.argv[2]) if len(sys.argv) > 2 else 5
streams = [torch.cuda.Stream() for _ in range(S)]
vols, outs = [], []
for i, st in enumerate(streams):   # interleave: (Input, output) pair i, then arena i
    vols.append(torch.rand((N, N, N), device="cuda")); outs.append(torch.empty_like(vols[-1]))
    with torch.cuda.stream(st):
        PD_TV_cupy(vols[i], 0.01, 3, 0, 1, 12.0, 0, False, out=outs[i]); st.synchronize()
    free, total = torch.cuda.mem_get_info()
    print(f"pair {i} + arena {i} allocated, {(total - free) / 1e9:6.1f} GB in use", flush=True)
def run(st, v, o):
    with torch.cuda.stream(st):
        ts = []
        for _ in range(2):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st); PD_TV_cupy(v, 0.01, 30, 0, 1, 12.0, 0, False, out=o); e1.record(st); st.synchronize()
            ts.append(e0.elapsed_time(e1) / 10)
    return min(ts)
print("ms per launch (mean over the 10 launches of a prox); rows = (Input, output) pair, columns = arena")
print("          " + " ".join(f"arena{j:<2d}" for j in range(S)))
for i in range(S):
    print(f"pair {i:<2d}  " + " ".join(f"{run(streams[j], vols[i], outs[i]):7.3f}" for j in range(S)), flush=True)
