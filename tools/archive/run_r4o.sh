cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
T=${1:-r4o}; O=gpurun_out/$T; mkdir -p $O
timeout 300 python tools/archive/probes/ir_methods_probe.py > $O/ir_methods.txt 2>&1
timeout 300 python tools/fbp_bench.py > $O/fbp_bench.txt 2>&1
timeout 300 python tools/fourier_bench.py > $O/fourier_bench.txt 2>&1
grep -v amdgpu $O/ir_methods.txt; grep -v amdgpu $O/fbp_bench.txt; grep -v amdgpu $O/fourier_bench.txt
