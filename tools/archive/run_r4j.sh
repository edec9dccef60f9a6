# round 4, call J: the dense-angle form of the forward projector (256 pixels x 16 angles): parity tests, then a same-box A/B on
# BASELINE configs[3]'s angle set (ab/lib_nodense.so = -DTOMO_FP_NO_DENSE16, ab/lib_dense.so = as shipped)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
T=${1:-r4j}; O=gpurun_out/$T; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -x -k "dense_angle or config3 or forward_projection or random_geometries or config5 or full_size_projector" 2>&1 | tail -8 > $O/pytest_fp.log
cat > /tmp/fpab.sh <<'EOS'
python tools/kernel_bench.py 2048 256 1500 2 | grep -E "^(BP|FP)  variant 0"
python tools/kernel_bench.py 1024 256 900 2 | grep -E "^(BP|FP)  variant 0"
python tools/kernel_bench.py 2560 64 1800 2 | grep -E "^(BP|FP)  variant 0"
python bench.py --config cfg3-share --steps 2 --warmup 1 --no-cpu --no-pmc | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg3-share', round(d['ms_per_step'],1), {k:round(v['avg_ms'],2) for k,v in d['kernels'].items()})"
EOS
bash tools/run_ab.sh ${T}_ab bash /tmp/fpab.sh > $O/fp_dense_ab.txt 2>&1
cat $O/pytest_fp.log; grep -v amdgpu $O/fp_dense_ab.txt
