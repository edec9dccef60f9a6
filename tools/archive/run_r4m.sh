# round 4, call M: when does the dense-angle FP form pay on SMALL problems?  lib_nodense (never), lib_thresh768 (>= 768 workgroups per
# class launch), lib_thresh3072 (>= 3072: as shipped)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
T=${1:-r4m}; O=gpurun_out/$T; mkdir -p $O
cat > /tmp/fpab.sh <<'EOS'
for shape in "256 256 360" "512 512 360" "512 128 720" "1024 128 1800" "768 768 900"; do echo "shape $shape"; python tools/kernel_bench.py $shape 3 | grep -E "^FP  variant 0"; done
python bench.py --config cfg1 --steps 50 --warmup 5 --no-cpu --no-pmc | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg1', round(d['value'],1), {k:round(v['avg_ms'],3) for k,v in d['kernels'].items()})"
EOS
bash tools/run_ab.sh ${T}_ab bash /tmp/fpab.sh > $O/fp_dense_small_ab.txt 2>&1
grep -v amdgpu $O/fp_dense_small_ab.txt
