# round 6: forward projector, chunk row origin through readfirstlane (uniform counters stay scalar): parity, same-box A/B against HEAD
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6t; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_shares.py -m gpu -q -x 2>&1 | tail -4 > $O/pytest.log
bash tools/run_ab.sh r6t_ab python tools/fp_time.py 7 > $O/fp_k0_scalar_ab.txt 2>&1
tail -3 $O/pytest.log; cut -c1-170 $O/fp_k0_scalar_ab.txt
