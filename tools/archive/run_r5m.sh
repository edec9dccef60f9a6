# round 5: the other BASELINE shapes with live counter traffic and rocprofv3 kernel statistics
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5m; mkdir -p $O
timeout 900 python bench.py --config cfg5-share --steps 2 --warmup 1 --no-cpu --live-pmc > $O/bench_cfg5_share_live_pmc.json 2> $O/a.err
timeout 900 python bench.py --config cfg3-share --steps 2 --warmup 1 --no-cpu --live-pmc > $O/bench_cfg3_share_live_pmc.json 2> $O/b.err
for c in cfg3-share cfg5-share; do
  rocprofv3 --kernel-trace --stats -d $O/prof_$c -o bench -- python bench.py --config $c --steps 2 --warmup 1 --no-cpu --no-pmc > $O/bench_${c}_prof_line.json 2> $O/prof_$c.err
  DB=$(find $O/prof_$c -name "*.db" | head -1)
  python tools/rocpd_summary.py "$DB" $O/kernel_stats_$c.txt | head -8
  find $O/prof_$c -type f -size +1M -delete
done
python - <<'PY'
import json
for f in ("bench_cfg5_share_live_pmc", "bench_cfg3_share_live_pmc"):
    d = json.load(open(f"gpurun_out/r5m/{f}.json"))
    r = d["roofline"]
    print(f, d["value"], r["kernel"], r["frac"], r.get("traffic"), str(r.get("traffic_source"))[:300])
PY
