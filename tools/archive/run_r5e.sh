# round 5: planar sinograms re-laid quad-interleaved inside tomo_bp3d* (every back projection stages by LDS-DMA)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5e; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log
tail -6 $O/pytest_gpu.log
for shape in "1024 1024 75" "2560 270 150" "2048 256 1500"; do
  echo "== BP epilogues, N NZ NA = $shape" >> $O/bp_epi.txt
  timeout 300 python tools/archive/probes/bp_epi_bench.py $shape 2>/dev/null | grep -v amdgpu >> $O/bp_epi.txt
done
cat $O/bp_epi.txt
timeout 300 python tools/fbp_bench.py > $O/fbp_bench.txt 2>&1; grep -v amdgpu $O/fbp_bench.txt
timeout 300 python tools/archive/probes/ir_methods_probe.py > $O/ir_methods.txt 2>&1; grep -v amdgpu $O/ir_methods.txt
timeout 900 python bench.py --config cfg5-share --steps 2 --warmup 1 --no-cpu > $O/bench_cfg5_share.json 2> $O/bench_cfg5_share.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5e/bench_cfg5_share.json"))
print(d["value"], d["ms_per_step"], {k: round(v["avg_ms"], 3) for k, v in d["kernels"].items()})
PY
