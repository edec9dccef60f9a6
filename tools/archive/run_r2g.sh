cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r2g
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r2g/pytest.log
timeout 600 python bench.py --steps 3 --warmup 1 > gpurun_out/r2g/bench_n1.json 2> gpurun_out/r2g/bench_n1.err
timeout 300 python tools/kernel_bench.py 1024 1024 75 3 > gpurun_out/r2g/kernel_bench_1024.txt 2>&1
tail -4 gpurun_out/r2g/pytest.log; cut -c1-300 gpurun_out/r2g/bench_n1.json; grep -v amdgpu gpurun_out/r2g/kernel_bench_1024.txt | grep -v "^BP\|^FP"
