# full measurement set of a source state: usage  bash tools/run_full_set.sh <tag>   (round 5: r5z, round 6: r6z)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
T=${1:-r6z}; O=gpurun_out/$T; mkdir -p $O
timeout 2700 python -m pytest tests -m gpu -q 2>&1 | tail -8 > $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu --no-pmc > $O/bench_prof_line.json 2> $O/bench_prof.err
DB=$(find $O/prof -name "*.db" | head -1)
python tools/rocpd_summary.py "$DB" $O/bench_kernel_stats.txt | head -10
find $O/prof -type f -size +1M -delete
PMC_GROUPS="FETCH_SIZE;WRITE_SIZE;TCC_HIT_sum TCC_MISS_sum" PMC_PD_ITERS=9 bash tools/pmc_run.sh ${T}_a pdtv0 pdtv0h 2>&1 | grep -v native | tail -6
PMC_GROUPS="FETCH_SIZE;WRITE_SIZE" bash tools/pmc_run.sh ${T}_b roftv bpq fpq 2>&1 | grep -v native | tail -8
python tools/update_pmc_traffic.py gpurun_out/pmc_${T}_a gpurun_out/pmc_${T}_b profiles/${T}_pmc_fetch_write.txt > $O/pmc_update.log 2>&1
cp profiles/pmc_traffic.json $O/pmc_traffic.json
cat gpurun_out/pmc_${T}_a/summary.txt gpurun_out/pmc_${T}_b/summary.txt > $O/pmc_fetch_write.txt
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu --no-pmc > $O/bench_20_steps.json 2> $O/bench_20_steps.err
timeout 600 python bench.py --exact-tv --steps 3 --warmup 1 --no-cpu --no-pmc > $O/bench_exact_tv.json 2> $O/bench_exact_tv.err
timeout 600 python bench.py --half --steps 3 --warmup 1 --no-cpu --no-pmc > $O/bench_half.json 2> $O/bench_half.err
timeout 600 python bench.py --config cfg1 --steps 50 --warmup 5 > $O/bench_cfg1.json 2> $O/bench_cfg1.err
timeout 900 python bench.py --config cfg3-share --steps 2 --warmup 1 > $O/bench_cfg3_share.json 2> $O/bench_cfg3_share.err
timeout 900 python bench.py --config cfg5-share --steps 2 --warmup 1 > $O/bench_cfg5_share.json 2> $O/bench_cfg5_share.err
timeout 1500 python bench.py --config cfg3 --steps 3 --warmup 1 > $O/bench_cfg3_full.json 2> $O/bench_cfg3_full.err
timeout 300 python tools/kernel_bench.py 1024 1024 75 3 > $O/kernel_bench_1024.txt 2>&1
# the 2048^3-class line as a driver that only varies --gpus reaches it, under rocprofv3 as well (round 6)
BENCH_CONFIG=cfg3 rocprofv3 --kernel-trace --stats -d $O/prof3 -o cfg3 -- python bench.py --steps 2 --warmup 1 --no-cpu --no-pmc > $O/bench_cfg3_prof_line.json 2> $O/bench_cfg3_prof.err
DB3=$(find $O/prof3 -name "*.db" | head -1)
python tools/rocpd_summary.py "$DB3" $O/bench_cfg3_kernel_stats.txt | head -8
find $O/prof3 -type f -size +1M -delete
# multi-rank dry runs on the one GPU (gloo transport, "oversubscribed": functional only) against the final library
timeout 300 python tools/rccl_preflight.py --gpus 2 --n 1024 --nz 64 --reps 3 > $O/rccl_preflight_dryrun.json 2> $O/rccl_preflight_dryrun.err
timeout 600 python bench.py --gpus 2 --strong --n 512 --nz 128 --angles 360 --steps 2 --warmup 1 > $O/bench_2ranks_strong_dryrun.json 2> $O/bench_2ranks_strong_dryrun.err
timeout 600 python bench.py --gpus 2 --n 512 --nz 128 --angles 360 --steps 2 --warmup 1 > $O/bench_2ranks_weak_dryrun.json 2> $O/bench_2ranks_weak_dryrun.err
BENCH_NORTH_STAR_TEST=1 timeout 900 python bench.py --gpus 2 --north-star --n 512 --nz 128 --angles 360 --steps 2 --warmup 1 > $O/bench_2ranks_north_star_child_dryrun.json 2> $O/bench_2ranks_north_star_child_dryrun.err
tail -4 $O/pytest.log; tail -1 $O/smoke.log; cat $O/pmc_update.log
for f in bench_default bench_20_steps bench_exact_tv bench_half bench_cfg1 bench_cfg3_share bench_cfg5_share bench_cfg3_full bench_cfg3_prof_line rccl_preflight_dryrun bench_2ranks_strong_dryrun bench_2ranks_weak_dryrun bench_2ranks_north_star_child_dryrun; do cut -c1-150 $O/$f.json; done
