# round 6: final sources (after the forward projector's scalar row origin): PMC traffic refresh, GPU suite, smoke, bench lines
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
bash tools/run_pmc_refresh.sh r6x > gpurun_out/r6x_refresh.log 2>&1
O=gpurun_out/r6x; mkdir -p $O
timeout 2700 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -5 > $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_20_steps.json 2> $O/bench_20_steps.err
timeout 900 python bench.py --config cfg3-share --steps 2 --warmup 1 --no-cpu > $O/bench_cfg3_share.json 2> $O/bench_cfg3_share.err
timeout 900 python bench.py --config cfg5-share --steps 2 --warmup 1 --no-cpu > $O/bench_cfg5_share.json 2> $O/bench_cfg5_share.err
BENCH_CONFIG=cfg3 timeout 1500 python bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu > $O/bench_cfg3_full.json 2> $O/bench_cfg3_full.err
tail -3 gpurun_out/r6x_refresh.log | cut -c1-200; tail -3 $O/pytest.log; tail -1 $O/smoke.log; for f in bench_20_steps bench_cfg3_share bench_cfg5_share bench_cfg3_full; do cut -c1-160 $O/$f.json; done
