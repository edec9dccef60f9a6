# round 4: is the fast / slow state of the PD_TV launch a property of where the process's arrays were allocated?
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4z; mkdir -p $O
for i in 1 2 3 4; do
  echo "== process $i" >> $O/placement.txt
  timeout 200 tools/probes/_build/pd_stream_probe place 6 >> $O/placement.txt 2>&1
  timeout 100 python tools/archive/probes/pd_time.py 1024 3 2>/dev/null | grep "default" >> $O/placement.txt
done
cat $O/placement.txt
