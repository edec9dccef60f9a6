# round 5: quad-interleaved residual layout between FP epilogue and BP staging -- parity, per-call timing, bench line
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5b; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log
tail -6 $O/pytest_gpu.log
for shape in "1024 1024 75" "2560 270 150" "2048 256 1500"; do
  echo "== BP epilogues, N NZ NA = $shape" >> $O/bp_epi.txt
  timeout 300 python tools/archive/probes/bp_epi_bench.py $shape 2>/dev/null | grep -v amdgpu >> $O/bp_epi.txt
done
cat $O/bp_epi.txt
timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu --no-pmc > $O/bench_line.json 2> $O/bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5b/bench_line.json"))
print(d["value"], d["ms_per_step"], {k: round(v["avg_ms"], 3) for k, v in d["kernels"].items()}, d.get("placement"))
PY
