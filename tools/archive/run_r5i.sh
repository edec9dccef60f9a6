# round 5: non-temporal hint on the back projector's one-touch epilogue streams (X_t read, X written): A/B of four builds, one box
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5i; mkdir -p $O
cp tomobar_amd/libtomo_mi355x.so /tmp/tree.so
for rep in 1 2; do
  for L in ab/lib_*.so; do
    n=$(basename $L .so); cp $L tomobar_amd/libtomo_mi355x.so
    echo "== $n (pass $rep)" >> $O/bp_epi.txt
    TOMO_MI355X_FLAVOUR=shipped timeout 200 python tools/archive/probes/bp_epi_bench.py 1024 1024 75 2>/dev/null | grep "quad\|plain" >> $O/bp_epi.txt
  done
done
for L in ab/lib_*.so; do
  n=$(basename $L .so); cp $L tomobar_amd/libtomo_mi355x.so
  PMC_GROUPS="FETCH_SIZE;WRITE_SIZE" bash tools/pmc_run.sh r5i_$n bpq 2>&1 | grep bp_brick | sed "s/^/$n /" >> $O/pmc.txt
done
cp /tmp/tree.so tomobar_amd/libtomo_mi355x.so
cat $O/bp_epi.txt; cut -c1-200 $O/pmc.txt
