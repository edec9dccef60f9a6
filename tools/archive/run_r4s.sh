# round 4, call S: rocprofv3 kernel statistics of the configs[3] share bench; LDS counters of the dense-angle FP form at that shape
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
T=${1:-r4s}; O=gpurun_out/$T; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python bench.py --config cfg3-share --steps 2 --warmup 1 --no-cpu --no-pmc > $O/bench_cfg3_share_prof_line.json 2> $O/bench_prof.err
DB=$(find $O/prof -name "*.db" | head -1)
python tools/rocpd_summary.py "$DB" $O/bench_cfg3_share_kernel_stats.txt | head -8
find $O/prof -type f -size +1M -delete
PMC_N=2048 PMC_NZ=256 PMC_NA=1500 PMC_GROUPS="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS;SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES;FETCH_SIZE;WRITE_SIZE" bash tools/pmc_run.sh ${T}_fp fp > $O/pmc_fp_cfg3.txt 2>&1
grep -v native $O/pmc_fp_cfg3.txt | tail -4
