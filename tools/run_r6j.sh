# round 6: LDS-DMA staging in the whole-row forward projector: parity, then same-box A/B against the previous commit
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6j; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_shares.py tests/test_gpu_recon.py tests/test_robust_terms.py tests/test_ring_terms.py -m gpu -q -x 2>&1 | tail -8 > $O/pytest.log
bash tools/run_ab.sh r6j_ab python tools/fp_time.py 5 > $O/fp_lds_dma_ab.txt 2>&1
tail -5 $O/pytest.log; cut -c1-250 $O/fp_lds_dma_ab.txt
