# round 5: kernel-timed placement of the PD_TV arena -- placement test, then the default bench in fresh processes (what gets picked, PD_TV ms)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5zc; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "placement or pdtv or tv_" 2>&1 | tail -4 > $O/pytest_subset.log
for i in 1 2 3 4 5; do
  timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu --no-pmc > $O/bench_$i.json 2> $O/bench_$i.err
done
tail -3 $O/pytest_subset.log
python - <<'PY'
import json
for i in range(1,6):
    try:
        d=json.loads(open(f'gpurun_out/r5zc/bench_{i}.json').read().strip().splitlines()[-1])
        print(i, round(d['value'],4), round(d['kernels']['pdtv']['avg_ms'],3), d['placement'])
    except Exception as e: print(i,'ERR',e)
PY
