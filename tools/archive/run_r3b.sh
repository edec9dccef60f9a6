cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3b; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15 > $O/pytest.log
timeout 300 python tools/archive/probes/fp_stage_probe.py 1024 1024 900 12 > $O/fp_stage_probe.txt 2>&1
timeout 300 python tools/archive/probes/fp_stage_probe.py 2048 256 1500 1 >> $O/fp_stage_probe.txt 2>&1
timeout 600 python bench.py --steps 3 --warmup 1 > $O/bench_cfg2.json 2> $O/bench_cfg2.err
timeout 600 python bench.py --config cfg1 --steps 50 --warmup 5 > $O/bench_cfg1.json 2> $O/bench_cfg1.err
timeout 600 python bench.py --gpus 2 --strong --n 256 --nz 64 --angles 180 --os 6 --inner 9 --steps 2 --warmup 1 > $O/bench_2ranks_dryrun.json 2> $O/bench_2ranks_dryrun.err
PMC_GROUPS="FETCH_SIZE;WRITE_SIZE;TCC_HIT_sum TCC_MISS_sum" PMC_PD_ITERS=9 PMC_PROBE=0 bash tools/pmc_run.sh r3b_probe0 pdtv0 2>&1 | grep -v native | tail -4 > $O/pmc_probe0.txt
PMC_GROUPS="FETCH_SIZE;WRITE_SIZE;TCC_HIT_sum TCC_MISS_sum" PMC_PD_ITERS=9 PMC_PROBE=3 bash tools/pmc_run.sh r3b_probe3 pdtv0 2>&1 | grep -v native | tail -4 > $O/pmc_probe3.txt
tail -4 $O/pytest.log; cat $O/fp_stage_probe.txt; cut -c1-400 $O/bench_cfg2.json; cut -c1-300 $O/bench_cfg1.json; cut -c1-300 $O/bench_2ranks_dryrun.json; tail -3 $O/bench_2ranks_dryrun.err; cat $O/pmc_probe0.txt $O/pmc_probe3.txt
