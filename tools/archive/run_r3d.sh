cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3d; mkdir -p $O
timeout 300 python tools/archive/probes/fp_stage_probe.py 1024 1024 900 12 > $O/fp_stage_probe.txt 2>&1
timeout 300 python tools/archive/probes/fp_stage_probe.py 2048 256 1500 1 >> $O/fp_stage_probe.txt 2>&1
timeout 300 python tools/archive/probes/fp_stage_probe.py 2560 270 1800 12 >> $O/fp_stage_probe.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15 > $O/pytest.log
cat $O/fp_stage_probe.txt; tail -5 $O/pytest.log
