# round 6: per-angle lane multipliers in the whole-row forward projector (A/B + parity), Huber / Student's-t terms
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6c; mkdir -p $O
TOMO_MI355X_FLAVOUR=dev timeout 900 python tools/fp_mult_ab.py > $O/fp_mult_ab.txt 2>&1
timeout 1500 python -m pytest tests/test_robust_terms.py tests/test_gpu_parity.py tests/test_ring_terms.py -m gpu -q -x 2>&1 | tail -15 > $O/pytest_a.log
timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_shares.py -m gpu -q 2>&1 | tail -15 > $O/pytest_b.log
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu --no-pmc > $O/bench_5.json 2> $O/bench_5.err
cat $O/fp_mult_ab.txt; tail -8 $O/pytest_a.log; tail -8 $O/pytest_b.log; cut -c1-600 $O/bench_5.json
