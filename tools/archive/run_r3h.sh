cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3h; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_slab.py -m gpu -q -x -k "pdtv or tv_random or halo or thin" 2>&1 | tail -8 > $O/pytest.log
PD_IT=30 timeout 600 python tools/archive/probes/pd_sweep.py 1024 0 22 21 2 > $O/pd_sweep.txt 2>&1
tail -4 $O/pytest.log; cat $O/pd_sweep.txt
