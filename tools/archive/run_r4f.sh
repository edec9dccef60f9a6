# round 4, call F: details of three failing tests; FP lane permutation A/B on BASELINE configs[3]'s dense angle set; the default
# bench line with the live PMC passes
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
T=${1:-r4f}; O=gpurun_out/$T; mkdir -p $O
timeout 1200 python -m pytest "tests/test_gpu_fullsize.py::test_full_size_tv_on_z_invariant_volume" "tests/test_gpu_fullsize.py::test_config5_shape_per_gpu" "tests/test_gpu_fullsize.py::test_config4_geometry_fista_ring_end_to_end_against_oracle" -m gpu -q --tb=short 2>&1 | tail -60 > $O/pytest_failing.log
cat > /tmp/fpab.sh <<'EOS'
python tools/kernel_bench.py 2048 256 1500 2 | grep -E "^(BP|FP)  variant 0"
python tools/kernel_bench.py 2048 128 750 2 | grep -E "^(BP|FP)  variant 0"
python bench.py --config cfg3-share --steps 2 --warmup 1 --no-cpu --no-pmc | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg3-share', round(d['ms_per_step'],1), {k:round(v['avg_ms'],2) for k,v in d['kernels'].items()})"
EOS
bash tools/run_ab.sh ${T}_ab bash /tmp/fpab.sh > $O/fp_ab_cfg3.txt 2>&1
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
cat $O/pytest_failing.log; cat $O/fp_ab_cfg3.txt; python -c "
import json;d=json.load(open('$O/bench_default.json'));print(d['value'], json.dumps(d['roofline'])[:1500])"; tail -3 $O/bench_default.err
