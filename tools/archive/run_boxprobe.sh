# which kind of box is this?  copy ceiling, PD_TV launch, clocks -- one line each (round 4: the lease pool has fast and slow boxes for the PD_TV launch)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
T=${1:-box}; O=gpurun_out/$T; mkdir -p $O
{ rocm-smi --showcomputepartition --showmemorypartition 2>/dev/null | grep -iE "partition" | head -4; rocm-smi --showclocks 2>/dev/null | grep -E "sclk|mclk|fclk|socclk" | head -6; rocm-smi --showpower --showtemp 2>/dev/null | grep -E "Power|Temperature" | head -6;
  timeout 200 tools/probes/_build/hbm_copy_probe 2>/dev/null | grep -E "flat float4 copy +threads= 256|flat float4 copy, nt.*threads= 256|mix 5r/4w vec1 flat, skew 69888 +threads= 256|mix 5r/4w vec1 grid" | head -6;
  timeout 200 python tools/archive/probes/pd_time.py 1024 3 2>/dev/null | grep -v amdgpu;
  rocm-smi --showclocks 2>/dev/null | grep -E "sclk|mclk|fclk" | head -4; } > $O/boxprobe.txt 2>&1
cat $O/boxprobe.txt
