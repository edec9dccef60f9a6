cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r2h
rocprofv3 --kernel-trace --stats -d gpurun_out/r2h/prof -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu > gpurun_out/r2h/bench_prof_line.json 2> gpurun_out/r2h/bench_prof.err
DB=$(find gpurun_out/r2h/prof -name "*.db" | head -1)
python tools/rocpd_summary.py "$DB" gpurun_out/r2h/bench_kernel_stats.txt | head -12
find gpurun_out/r2h/prof -type f -size +1M -delete
PMC_GROUPS="FETCH_SIZE;WRITE_SIZE;TCC_HIT_sum TCC_MISS_sum" bash tools/pmc_run.sh r2h pdtv0 pdtv0h bp0 fp 2>&1 | grep -v native | tail -8
