cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r3a
timeout 600 python tools/archive/probes/pd_halo_probe.py 1024 30 > gpurun_out/r3a/pd_halo_probe.txt 2>&1
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 > gpurun_out/r3a/pytest.log
timeout 600 python bench.py --steps 3 --warmup 1 > gpurun_out/r3a/bench_n1.json 2> gpurun_out/r3a/bench_n1.err
cat gpurun_out/r3a/pd_halo_probe.txt; tail -3 gpurun_out/r3a/pytest.log; cut -c1-300 gpurun_out/r3a/bench_n1.json
