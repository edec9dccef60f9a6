# round 4, call B: v_rcp all-ones check, PD stream-mix ceilings, FP lane->pixel permutation A/B (ab/lib_head.so vs
# ab/lib_perm.so), its LDS conflict counters, and the parity tests on the tree's library.  usage: bash tools/run_r4b.sh [tag]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
T=${1:-r4b}; O=gpurun_out/$T; mkdir -p $O
timeout 300 tools/probes/_build/recip_allones_probe > $O/recip_allones_probe.txt 2>&1
timeout 900 tools/probes/_build/hbm_copy_probe > $O/hbm_copy_probe.txt 2>&1
cat > /tmp/fpab.sh <<'EOS'
python tools/archive/probes/fp_conflict_probe.py 1024 1024
python tools/kernel_bench.py 1024 1024 75 3 | grep -E "^(BP|FP)  variant 0"
python tools/kernel_bench.py 2048 128 750 2 | grep -E "^(BP|FP)  variant 0"
python tools/kernel_bench.py 2560 128 150 2 | grep -E "^(BP|FP)  variant 0"
EOS
bash tools/run_ab.sh ${T}_ab bash /tmp/fpab.sh > $O/fp_ab.txt 2>&1
timeout 1700 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -8 > $O/pytest.log
PMC_GROUPS="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS;SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES" bash tools/pmc_run.sh ${T}_perm fp bp0 > $O/pmc_perm.txt 2>&1
cp ab/lib_head.so tomobar_amd/libtomo_mi355x.so
PMC_GROUPS="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS;SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES" bash tools/pmc_run.sh ${T}_head fp > $O/pmc_head.txt 2>&1
cat $O/recip_allones_probe.txt; tail -22 $O/hbm_copy_probe.txt; cat $O/fp_ab.txt; cat $O/pytest.log; grep -v native $O/pmc_perm.txt | tail -4; grep -v native $O/pmc_head.txt | tail -3
