cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r2i
rocprofv3 --kernel-trace --stats -d gpurun_out/r2i/prof -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu > gpurun_out/r2i/bench_prof_line.json 2> gpurun_out/r2i/bench_prof.err
DB=$(find gpurun_out/r2i/prof -name "*.db" | head -1)
python tools/rocpd_summary.py "$DB" gpurun_out/r2i/bench_kernel_stats.txt | head -8
find gpurun_out/r2i/prof -type f -size +1M -delete
PMC_GROUPS="FETCH_SIZE;WRITE_SIZE;TCC_HIT_sum TCC_MISS_sum" bash tools/pmc_run.sh r2i pdtv0 pdtv0h bp0 fp roftv 2>&1 | grep -v native | tail -8
timeout 600 python bench.py --steps 3 --warmup 1 > gpurun_out/r2i/bench_n1.json 2> gpurun_out/r2i/bench_n1.err
timeout 600 python bench.py --steps 2 --warmup 1 --half --no-cpu > gpurun_out/r2i/bench_half.json 2>/dev/null
timeout 600 python bench.py --gpus 2 --steps 1 --warmup 1 --nz 128 --no-cpu > gpurun_out/r2i/bench_n2_dry.json 2> gpurun_out/r2i/bench_n2_dry.err
timeout 600 python bench.py --gpus 2 --strong --steps 1 --warmup 1 --nz 256 --no-cpu > gpurun_out/r2i/bench_n2_strong_dry.json 2>/dev/null
timeout 600 python bench.py --steps 2 --warmup 1 --ring 1e-4 --no-cpu > gpurun_out/r2i/bench_ring.json 2>/dev/null
timeout 600 python tools/archive/probes/admm_cfg3_probe.py > gpurun_out/r2i/admm_cfg3.txt 2>&1
for f in bench_n1 bench_half bench_n2_dry bench_n2_strong_dry bench_ring; do cut -c1-170 gpurun_out/r2i/$f.json; done; tail -3 gpurun_out/r2i/admm_cfg3.txt
