cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
T=${1:-r4t}; O=gpurun_out/$T; mkdir -p $O
timeout 2700 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
cat $O/pytest.log; tail -1 $O/smoke.log; cut -c1-200 $O/bench_default.json
