# round 5: does the ORDER of set-up matter for the placement search in bench.py?  arena reserved first (device empty) vs after the data synthesis
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5f; mkdir -p $O; rm -f $O/summary.txt
for i in 1 2 3 4 5; do
  for late in 0 1; do
    BENCH_RESERVE_LATE=$late timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu --no-pmc > $O/bench_${late}_$i.json 2>/dev/null
    python - $O/bench_${late}_$i.json $late >> $O/summary.txt <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
p = d.get("placement") or {}
print(f"reserve {'after the data' if sys.argv[2] == '1' else 'first          '}: {d['value']:.4f} it/s  PD_TV {d['kernels']['pdtv']['avg_ms']:.3f} ms  fast {p.get('fast')}  tries {len(p.get('scores_GBps', []))}  scores {p.get('scores_GBps')}")
PY
  done
done
cat $O/summary.txt
