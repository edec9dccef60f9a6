# round 6: final-tree check after the clean rebuild: residual-layout check, child-job tests, smoke, the whole GPU suite
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6h; mkdir -p $O
timeout 2700 python -m pytest tests -m gpu -q 2>&1 | tail -8 > $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -4 $O/pytest.log; tail -1 $O/smoke.log; cut -c1-200 $O/bench_default.json
