# round 5, final tree: whole GPU suite + smoke + the multi-rank dry runs (two ranks on the one GPU, gloo transport)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5r; mkdir -p $O
timeout 2700 python -m pytest tests -m gpu -q 2>&1 | tail -8 > $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
timeout 300 python tools/rccl_preflight.py --gpus 2 --n 1024 --nz 64 --reps 3 > $O/rccl_preflight_dryrun.json 2> $O/rccl_preflight_dryrun.err
timeout 600 python bench.py --gpus 2 --strong --n 512 --nz 128 --angles 360 --steps 2 --warmup 1 > $O/bench_2ranks_strong_dryrun.json 2> $O/bench_2ranks_strong_dryrun.err
timeout 600 python bench.py --gpus 2 --n 512 --nz 128 --angles 360 --steps 2 --warmup 1 > $O/bench_2ranks_weak_dryrun.json 2> $O/bench_2ranks_weak_dryrun.err
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu --no-pmc > $O/bench_20_steps.json 2> $O/bench_20_steps.err
tail -4 $O/pytest.log; tail -1 $O/smoke.log; cut -c1-400 $O/rccl_preflight_dryrun.json; for f in bench_2ranks_strong_dryrun bench_2ranks_weak_dryrun bench_20_steps; do cut -c1-300 $O/$f.json; tail -2 $O/$f.err | cut -c1-200; done
