# round 6: supervised north-star child job end to end (2 ranks on the one GPU), robust terms, two-process slab cases; the
# 2048^3-class line (BENCH_CONFIG=cfg3: whole configs[3] on one GPU) plain and under rocprofv3 --kernel-trace --stats
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6d; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_bench_child.py tests/test_robust_terms.py tests/test_gpu_slab_fista.py -m gpu -q --durations=5 2>&1 | tail -25 > $O/pytest.log
BENCH_CONFIG=cfg3 timeout 1500 python bench.py --gpus 1 --steps 3 --warmup 1 > $O/bench_cfg3_env.json 2> $O/bench_cfg3_env.err
BENCH_CONFIG=cfg3 rocprofv3 --kernel-trace --stats -d $O/prof -o cfg3 -- python bench.py --steps 2 --warmup 1 --no-cpu --no-pmc > $O/bench_cfg3_prof_line.json 2> $O/bench_cfg3_prof.err
DB=$(find $O/prof -name "*.db" | head -1)
python tools/rocpd_summary.py "$DB" $O/bench_cfg3_kernel_stats.txt | head -12
find $O/prof -type f -size +1M -delete
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu --no-pmc > $O/bench_20_steps.json 2> $O/bench_20_steps.err
tail -12 $O/pytest.log; cut -c1-700 $O/bench_cfg3_env.json; tail -3 $O/bench_cfg3_env.err; cut -c1-300 $O/bench_20_steps.json
