# round 6: forward projector staging descriptors end with the window (no compare / select per item): parity + fuzz, same-box A/B against HEAD
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6v; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_shares.py tests/test_gpu_recon.py tests/test_fbp.py tests/test_vertical_cor.py -m gpu -q -x 2>&1 | tail -4 > $O/pytest.log
timeout 500 python tools/fuzz_campaign.py --minutes 5 --seed0 12000 > $O/fuzz_campaign.txt 2>&1
bash tools/run_ab.sh r6v_ab python tools/fp_time.py 7 > $O/fp_window_descriptor_ab.txt 2>&1
tail -3 $O/pytest.log; tail -2 $O/fuzz_campaign.txt | cut -c1-200; cut -c1-170 $O/fp_window_descriptor_ab.txt
