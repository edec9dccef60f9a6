# full measurement set of a source state: usage  bash tools/run_r3p.sh <tag>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
T=${1:-r3p}; O=gpurun_out/$T; mkdir -p $O
timeout 1700 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu > $O/bench_prof_line.json 2> $O/bench_prof.err
DB=$(find $O/prof -name "*.db" | head -1)
python tools/rocpd_summary.py "$DB" $O/bench_kernel_stats.txt | head -8
find $O/prof -type f -size +1M -delete
PMC_GROUPS="FETCH_SIZE;WRITE_SIZE;TCC_HIT_sum TCC_MISS_sum" PMC_PD_ITERS=9 bash tools/pmc_run.sh ${T}_a pdtv0 pdtv0h 2>&1 | grep -v native | tail -6
PMC_GROUPS="FETCH_SIZE;WRITE_SIZE" bash tools/pmc_run.sh ${T}_b roftv bp0 fp 2>&1 | grep -v native | tail -8
python tools/update_pmc_traffic.py gpurun_out/pmc_${T}_a gpurun_out/pmc_${T}_b profiles/${T}_pmc_fetch_write.txt > $O/pmc_update.log 2>&1
cp profiles/pmc_traffic.json $O/pmc_traffic.json
timeout 600 python bench.py --steps 3 --warmup 1 > $O/bench_n1.json 2> $O/bench_n1.err
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 600 python bench.py --config cfg1 --steps 50 --warmup 5 > $O/bench_cfg1.json 2> $O/bench_cfg1.err
timeout 900 python bench.py --config cfg3-share --steps 2 --warmup 1 > $O/bench_cfg3_share.json 2> $O/bench_cfg3_share.err
timeout 900 python bench.py --config cfg5-share --steps 2 --warmup 1 > $O/bench_cfg5_share.json 2> $O/bench_cfg5_share.err
timeout 600 python bench.py --half --steps 2 --warmup 1 --no-cpu > $O/bench_half.json 2> $O/bench_half.err
timeout 600 python bench.py --gpus 2 --strong --n 512 --nz 128 --angles 360 --steps 2 --warmup 1 > $O/bench_2ranks_strong_dryrun.json 2> $O/bench_2ranks_strong_dryrun.err
timeout 300 python tools/kernel_bench.py 1024 1024 75 3 > $O/kernel_bench_1024.txt 2>&1
timeout 300 python tools/archive/probes/bp_epi_bench.py > $O/bp_epilogues.txt 2>&1
timeout 300 python tools/archive/probes/pd_halo_probe.py 1024 30 > $O/pd_probes.txt 2>&1
tail -3 $O/pytest.log; tail -1 $O/smoke.log; cat $O/pmc_update.log; for f in bench_n1 bench_default bench_cfg1 bench_cfg3_share bench_cfg5_share bench_half bench_2ranks_strong_dryrun; do cut -c1-160 $O/$f.json; done
