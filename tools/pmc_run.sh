#!/bin/bash
# usage: tools/pmc_run.sh <tag> <what...>   -- separate rocprofv3 --pmc passes per probe (kernel-trace only, csv)
# PMC_GROUPS="A B;C D" overrides the counter groups (one pass per ';'-separated group).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1; shift
mkdir -p gpurun_out/pmc_${tag}
groups=${PMC_GROUPS:-"FETCH_SIZE;WRITE_SIZE;SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES;TCC_HIT_sum TCC_MISS_sum"}
IFS=';' read -ra GR <<< "$groups"
for what in "$@"; do
  i=0
  for pmc in "${GR[@]}"; do
    d=gpurun_out/pmc_${tag}/${what}_g${i}
    timeout ${PMC_TIMEOUT:-300} rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $d -o p -- python tools/pmc_probe.py $what ${PMC_N:-1024} ${PMC_NZ:-1024} ${PMC_NA:-75} > $d.log 2>&1
    i=$((i+1))
  done
done
python tools/pmc_summary.py gpurun_out/pmc_${tag} | tee gpurun_out/pmc_${tag}/summary.txt
find gpurun_out/pmc_${tag} -name "*kernel_trace.csv" -delete; find gpurun_out/pmc_${tag} -name "*agent_info.csv" -delete
