# round 6: lane multipliers on odd wide detectors (new parity test)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6l; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "lane_multipliers or residual_buffer_must" 2>&1 | tail -6 > $O/pytest.log
cat $O/pytest.log
