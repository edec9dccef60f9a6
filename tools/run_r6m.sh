# round 6: what the driver runs at round end, on the final tree
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6m; mkdir -p $O
timeout 2700 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -6 > $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_20_steps.json 2> $O/bench_20_steps.err
tail -3 $O/pytest.log; tail -1 $O/smoke.log; cut -c1-250 $O/bench_20_steps.json
