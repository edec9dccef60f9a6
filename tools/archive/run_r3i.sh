cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3i; mkdir -p $O
timeout 1700 python -m pytest tests -m gpu -q 2>&1 | tail -8 > $O/pytest.log
timeout 600 python bench.py --half --steps 2 --warmup 1 --no-cpu > $O/bench_half.json 2> $O/bench_half.err
tail -4 $O/pytest.log; cut -c1-200 $O/bench_half.json
