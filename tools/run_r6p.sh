# round 6: EXPERIMENT -- staggered LDS writes of the next chunk during the sampling of the current one (-DTOMO_FP_STAGGER):
# parity of that build, then same-box A/B against the tree's build
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6p; mkdir -p $O
cp tomobar_amd/libtomo_mi355x.so /tmp/tree.so; cp ab/lib_b_after.so tomobar_amd/libtomo_mi355x.so
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_shares.py -m gpu -q -x -k "not dev" 2>&1 | tail -6 > $O/pytest_stagger.log
cp /tmp/tree.so tomobar_amd/libtomo_mi355x.so
bash tools/run_ab.sh r6p_ab python tools/fp_time.py 5 > $O/fp_stagger_ab.txt 2>&1
tail -4 $O/pytest_stagger.log; cut -c1-200 $O/fp_stagger_ab.txt
