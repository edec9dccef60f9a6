# round 4: PD_TV access-pattern testbed + which box
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4x; mkdir -p $O
timeout 120 python tools/archive/probes/pd_time.py 1024 3 2>/dev/null | grep -v amdgpu > $O/pd_time.txt
timeout 400 tools/probes/_build/pd_stream_probe > $O/pd_stream_probe.txt 2>&1
cat $O/pd_time.txt $O/pd_stream_probe.txt
