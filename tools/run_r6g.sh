# round 6: same-box A/B of the forward projector before / after the waterfall fix (ab/lib_a_before.so = commit bcf47de,
# ab/lib_b_after.so = the tree), and the LDS / VALU counters of the forward projector with and without the lane multipliers
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6g; mkdir -p $O
bash tools/run_ab.sh r6g_ab python tools/fp_time.py 5 > $O/fp_waterfall_ab.txt 2>&1
PMC_GROUPS="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS;SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU;SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" bash tools/pmc_run.sh r6g_mult fp fp4 > $O/pmc_fp_mult.log 2>&1
cp gpurun_out/pmc_r6g_mult/summary.txt $O/pmc_fp_mult_summary.txt
for L in a_before b_after; do
  cp tomobar_amd/libtomo_mi355x.so /tmp/tree.so; cp ab/lib_$L.so tomobar_amd/libtomo_mi355x.so
  PMC_N=2048 PMC_NZ=64 PMC_NA=1500 PMC_GROUPS="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS;SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE;SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" bash tools/pmc_run.sh r6g_dense_$L fp > $O/pmc_dense_$L.log 2>&1
  cp gpurun_out/pmc_r6g_dense_$L/summary.txt $O/pmc_dense_${L}_summary.txt
  cp /tmp/tree.so tomobar_amd/libtomo_mi355x.so
done
cat $O/fp_waterfall_ab.txt | cut -c1-260; cat $O/pmc_fp_mult_summary.txt $O/pmc_dense_a_before_summary.txt $O/pmc_dense_b_after_summary.txt | cut -c1-400
