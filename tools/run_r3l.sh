cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3l; mkdir -p $O
timeout 1700 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $O/pytest.log
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu > $O/bench.json 2> $O/bench.err
timeout 900 python bench.py --config cfg3-share --steps 2 --warmup 1 --no-cpu > $O/bench_cfg3.json 2> $O/bench_cfg3.err
timeout 300 python tools/fuzz_tv.py > $O/fuzz_tv.txt 2>&1
tail -3 $O/pytest.log; cut -c1-150 $O/bench.json; cut -c1-150 $O/bench_cfg3.json; tail -2 $O/fuzz_tv.txt
