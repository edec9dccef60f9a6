# round 6, first GPU call: FP combination probe (review item 1, step 1), co-scheduling probe (item 3), the new full-share tests (item 2)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6a; mkdir -p $O
timeout 600 tools/probes/_build/fp_combo_probe > $O/fp_combo_probe.txt 2>&1
timeout 600 python tools/cosched_probe.py 1024 1024 75 > $O/cosched_probe.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_shares.py "tests/test_gpu_fullsize.py::test_full_size_pdtv_z_varying_cone" -q -s --durations=0 2>&1 | tail -60 > $O/pytest_shares.log
cat $O/fp_combo_probe.txt; cat $O/cosched_probe.txt; tail -40 $O/pytest_shares.log
