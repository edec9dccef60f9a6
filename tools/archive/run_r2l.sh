cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r2l
timeout 1700 python -m pytest tests -m gpu -q 2>&1 | tail -6 > gpurun_out/r2l/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2l/smoke.log 2>&1
rocprofv3 --kernel-trace --stats -d gpurun_out/r2l/prof -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu > gpurun_out/r2l/bench_prof_line.json 2> gpurun_out/r2l/bench_prof.err
DB=$(find gpurun_out/r2l/prof -name "*.db" | head -1)
python tools/rocpd_summary.py "$DB" gpurun_out/r2l/bench_kernel_stats.txt | head -8
find gpurun_out/r2l/prof -type f -size +1M -delete
PMC_GROUPS="FETCH_SIZE;WRITE_SIZE;TCC_HIT_sum TCC_MISS_sum" PMC_PD_ITERS=9 bash tools/pmc_run.sh r2l_a pdtv0 2>&1 | grep -v native | tail -3
PMC_GROUPS="FETCH_SIZE;WRITE_SIZE" bash tools/pmc_run.sh r2l_b pdtv0h roftv bp0 fp 2>&1 | grep -v native | tail -8
timeout 600 python bench.py --steps 3 --warmup 1 > gpurun_out/r2l/bench_n1.json 2> gpurun_out/r2l/bench_n1.err
timeout 300 python tools/kernel_bench.py 1024 1024 75 3 > gpurun_out/r2l/kernel_bench_1024.txt 2>&1
tail -3 gpurun_out/r2l/pytest.log; tail -1 gpurun_out/r2l/smoke.log; cut -c1-200 gpurun_out/r2l/bench_n1.json
