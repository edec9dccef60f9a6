cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ref_tv.py tests/test_gpu_slab.py -m gpu -q -x 2>&1 | tail -3
bash tools/run_ab.sh r3m python tools/rof_bench.py
