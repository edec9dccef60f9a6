# round 5: vertical CoR component in z-slab mode (two processes on one GPU) + reserve_scratch test + suite
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5q; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_slab_fista.py tests/test_vertical_cor.py tests/test_gpu_recon.py tests/test_ring_terms.py -m gpu -x -q 2>&1 | tail -15 > $O/pytest_subset.log; cat $O/pytest_subset.log
