# round 4, call C: the whole GPU suite on the two-flavour build (shipped default = exact PD_TV arithmetic), smoke, per-kernel
# bench of both flavours, the default bench line under rocprofv3, 2D fused TV timing, multi-rank dry runs.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
T=${1:-r4c}; O=gpurun_out/$T; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
timeout 300 python tools/tv2d_bench.py > $O/tv2d_bench.txt 2>&1
timeout 400 python tools/kernel_bench.py 1024 1024 75 3 > $O/kernel_bench_1024.txt 2>&1
TOMO_MI355X_FLAVOUR=dev timeout 600 python tools/kernel_bench.py 1024 1024 75 3 > $O/kernel_bench_1024_dev.txt 2>&1
timeout 600 python bench.py --steps 3 --warmup 1 > $O/bench_n1.json 2> $O/bench_n1.err
timeout 300 python tools/rccl_preflight.py --gpus 2 --n 1024 --nz 64 --reps 3 > $O/rccl_preflight_dryrun.json 2> $O/rccl_preflight_dryrun.err
BENCH_NORTH_STAR_TEST=1 timeout 600 python bench.py --gpus 2 --strong --n 512 --nz 128 --angles 360 --steps 2 --warmup 1 > $O/bench_2ranks_dryrun.json 2> $O/bench_2ranks_dryrun.err
tail -12 $O/pytest.log; tail -1 $O/smoke.log; cat $O/tv2d_bench.txt; grep -v amdgpu $O/kernel_bench_1024.txt; grep -E "PD_TV|ROF" $O/kernel_bench_1024_dev.txt; cut -c1-300 $O/bench_n1.json; cut -c1-600 $O/rccl_preflight_dryrun.json; tail -2 $O/rccl_preflight_dryrun.err; python -c "
import json;d=json.load(open('$O/bench_2ranks_dryrun.json'));print(d['value'], d['config']['backend'], json.dumps(d.get('north_star'))[:600])"; tail -3 $O/bench_2ranks_dryrun.err
