# refresh profiles/pmc_traffic.json for the current kernel sources: usage  bash tools/run_pmc_refresh.sh <tag>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
T=${1:-pmcr}; O=gpurun_out/$T; mkdir -p $O
PMC_GROUPS="FETCH_SIZE;WRITE_SIZE;TCC_HIT_sum TCC_MISS_sum" PMC_PD_ITERS=9 bash tools/pmc_run.sh ${T}_a pdtv0 pdtv0h 2>&1 | grep -v native | tail -4
PMC_GROUPS="FETCH_SIZE;WRITE_SIZE" bash tools/pmc_run.sh ${T}_b roftv bpq fpq 2>&1 | grep -v native | tail -6
python tools/update_pmc_traffic.py gpurun_out/pmc_${T}_a gpurun_out/pmc_${T}_b profiles/${T}_pmc_fetch_write.txt > $O/pmc_update.log 2>&1
cp profiles/pmc_traffic.json $O/pmc_traffic.json
cat gpurun_out/pmc_${T}_a/summary.txt gpurun_out/pmc_${T}_b/summary.txt > $O/pmc_fetch_write.txt   # -> profiles/${T}_pmc_fetch_write.txt
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
cat $O/pmc_update.log; cut -c1-200 $O/bench_default.json
