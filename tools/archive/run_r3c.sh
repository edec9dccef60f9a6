cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3c; mkdir -p $O
timeout 600 python tools/archive/probes/rof_variants_probe.py > $O/rof_variants.txt 2>&1
timeout 300 python tools/archive/probes/fp_stage_probe.py 1024 1024 900 12 > $O/fp_stage_probe.txt 2>&1
cat $O/rof_variants.txt $O/fp_stage_probe.txt
