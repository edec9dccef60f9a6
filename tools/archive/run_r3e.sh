cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3e; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15 > $O/pytest.log
timeout 300 python tools/kernel_bench.py 1024 1024 75 3 > $O/kernel_bench_1024.txt 2>&1
tail -5 $O/pytest.log; cat $O/kernel_bench_1024.txt
