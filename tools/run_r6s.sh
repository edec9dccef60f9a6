# round 6: back projector reads (cos, sin, offset, window origin) of an angle with ONE ds_read_b128 (was ds_read_b96 + ds_read_b32): parity, then same-box A/B against HEAD
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6s; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_shares.py tests/test_gpu_recon.py tests/test_fbp.py -m gpu -q -x 2>&1 | tail -5 > $O/pytest.log
bash tools/run_ab.sh r6s_ab python tools/bp_time.py 7 > $O/bp_b128_records_ab.txt 2>&1
tail -3 $O/pytest.log; cut -c1-230 $O/bp_b128_records_ab.txt
