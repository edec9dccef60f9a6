# round 6: dwordx4 staging loads in the forward projector (whole-row + dense forms): parity, then same-box A/B against HEAD
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6o; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_shares.py tests/test_gpu_recon.py tests/test_robust_terms.py tests/test_ring_terms.py tests/test_fbp.py -m gpu -q -x 2>&1 | tail -8 > $O/pytest.log
timeout 600 python tools/fuzz_campaign.py --minutes 4 --seed0 9000 > $O/fuzz_campaign.txt 2>&1
bash tools/run_ab.sh r6o_ab python tools/fp_time.py 5 > $O/fp_x4_ab.txt 2>&1
tail -5 $O/pytest.log; tail -2 $O/fuzz_campaign.txt; cut -c1-250 $O/fp_x4_ab.txt
