# round 4, call I: the halo / cache-resident / one-workgroup probes of the PD_TV kernel once more (which box do we get?), and the new 2D tests
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
T=${1:-r4i}; O=gpurun_out/$T; mkdir -p $O
timeout 300 python tools/archive/probes/pd_time.py 1024 3 > $O/pd_time.txt 2>&1
timeout 600 python tools/archive/probes/pd_halo_probe.py 1024 30 0 > $O/pd_probes_default.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "2d_large or default_arithmetic" 2>&1 | tail -5 > $O/pytest_2d.log
grep -v amdgpu $O/pd_time.txt; grep -v amdgpu $O/pd_probes_default.txt; cat $O/pytest_2d.log
