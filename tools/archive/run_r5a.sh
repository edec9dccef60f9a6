# round 5: PD_TV tile-shape request-stream probe (review item 1) beside the shipped kernel on the same box
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5a; mkdir -p $O
timeout 200 python tools/archive/probes/pd_time.py 1024 3 2>/dev/null | grep -v amdgpu > $O/pd_time.txt
timeout 600 tools/probes/_build/pd_shape_probe 2 > $O/pd_shape_probe.txt 2>&1
cat $O/pd_time.txt $O/pd_shape_probe.txt
