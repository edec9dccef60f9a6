cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3l; mkdir -p $O
timeout 1700 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $O/pytest.log
timeout 600 python tools/fuzz_tv.py > $O/fuzz_tv.txt 2>&1
tail -3 $O/pytest.log; tail -2 $O/fuzz_tv.txt
