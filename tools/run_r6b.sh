# round 6: FP combination probe re-measured (warm-up + round robin), the full-share tests with the corrected sensitivity check
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6b; mkdir -p $O
timeout 900 tools/probes/_build/fp_combo_probe > $O/fp_combo_probe.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_shares.py "tests/test_gpu_fullsize.py::test_full_size_pdtv_z_varying_cone" -q -s --durations=0 2>&1 | tail -40 > $O/pytest_shares.log
cat $O/fp_combo_probe.txt; tail -30 $O/pytest_shares.log
