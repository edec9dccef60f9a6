# round 5 EXPERIMENT: FP whole-row form staged by LDS-DMA from quad-interleaved volume copies, 8 / 16 angles per workgroup
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5l; mkdir -p $O
for shape in "1024 1024 900 12" "1024 256 900 1" "2048 256 1500 1" "512 512 360 1" "1024 512 450 6"; do
  timeout 300 python tools/archive/probes/fp_qv_probe.py $shape 2>&1 | grep -v amdgpu >> $O/fp_qv_probe.txt
done
cat $O/fp_qv_probe.txt
