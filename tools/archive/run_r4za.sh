# round 4: arena placement search -- PD prox time with the search on (default 4 tries) and off, several processes each; then the bench
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4za; mkdir -p $O
for i in 1 2 3 4; do
  for tries in 1 4; do
    echo "== process $i, TOMO_MI355X_PLACE_TRIES=$tries" >> $O/pd_time.txt
    TOMO_MI355X_PLACE_TRIES=$tries timeout 200 python - >> $O/pd_time.txt 2>/dev/null <<'PY'
import sys, time, torch
sys.path.insert(0, ".")
from tomobar_amd import ops
from tomobar_amd.regularisersCuPy import PD_TV_cupy
vol = torch.rand((1024, 1024, 1024), device="cuda"); out = torch.empty_like(vol)
t0 = time.perf_counter(); PD_TV_cupy(vol, 0.01, 30, 0, 1, 12.0, 0, False, out=out); torch.cuda.synchronize(); first = time.perf_counter() - t0
ts = []
for _ in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); PD_TV_cupy(vol, 0.01, 30, 0, 1, 12.0, 0, False, out=out); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 10)
print(f"{min(ts):7.3f} ms per three-iteration launch; first call {first*1e3:7.1f} ms; placement {ops.placement_last()}")
PY
  done
done
cat $O/pd_time.txt
python bench.py --no-north-star > $O/bench_line.json 2> $O/bench_err.txt; cat $O/bench_line.json | cut -c1-400
python -c "
import json; l=json.load(open('$O/bench_line.json')); print(l['value'], l['roofline']['frac'], l.get('placement'), l['kernels'] if 'kernels' in l else '')"
