cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r2a
timeout 1500 python -m pytest tests -m gpu -x -q -s 2>&1 | tail -40 > gpurun_out/r2a/pytest.log
timeout 600 python bench.py --steps 3 --warmup 1 > gpurun_out/r2a/bench_n1.json 2> gpurun_out/r2a/bench_n1.err
timeout 600 python bench.py --gpus 2 --steps 1 --warmup 1 --nz 128 --no-cpu > gpurun_out/r2a/bench_n2_dry.json 2> gpurun_out/r2a/bench_n2_dry.err
timeout 600 python bench.py --gpus 2 --strong --steps 1 --warmup 1 --nz 128 --no-cpu > gpurun_out/r2a/bench_n2_strong_dry.json 2> gpurun_out/r2a/bench_n2_strong_dry.err
timeout 300 python tools/kernel_bench.py 1024 1024 75 3 > gpurun_out/r2a/kernel_bench_1024.txt 2>&1
tail -5 gpurun_out/r2a/pytest.log; cat gpurun_out/r2a/bench_n1.json | cut -c1-600; cat gpurun_out/r2a/bench_n2_dry.json | cut -c1-400; tail -3 gpurun_out/r2a/bench_n2_dry.err
