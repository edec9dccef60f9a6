# round 5, after the 64 x 64 non-temporal momentum / transpose kernels: whole GPU suite, PMC refresh (the projector sources changed), default bench line
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5x; mkdir -p $O
timeout 2700 python -m pytest tests -m gpu -q 2>&1 | tail -8 > $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
bash tools/run_pmc_refresh.sh r5x > $O/pmc_refresh.log 2>&1
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu --no-pmc > $O/bench_prof_line.json 2> $O/bench_prof.err
DB=$(find $O/prof -name "*.db" | head -1)
python tools/rocpd_summary.py "$DB" $O/bench_kernel_stats.txt | head -12
find $O/prof -type f -size +1M -delete
tail -4 $O/pytest.log; tail -1 $O/smoke.log; tail -5 $O/pmc_refresh.log; cut -c1-200 $O/bench_default.json
