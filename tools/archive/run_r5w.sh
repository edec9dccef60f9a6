# round 5: 64 x 64 float4 momentum / transpose kernels -- parity subset, kernel bench, headline
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5w; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_recon.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -5 > $O/pytest_subset.log
timeout 300 python tools/kernel_bench.py 1024 1024 75 3 2>&1 | grep -v amdgpu > $O/kernel_bench_1024.txt
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu --no-pmc > $O/bench_10_steps.json 2> $O/bench_10_steps.err
tail -3 $O/pytest_subset.log; grep -i "momentum\|transpose\|FP\|fp " $O/kernel_bench_1024.txt | head; cut -c1-330 $O/bench_10_steps.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5w/bench_10_steps.json'))
print({k:(v.get('ms') if isinstance(v,dict) else v) for k,v in d.get('kernels',{}).items()})
PY
