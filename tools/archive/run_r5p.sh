# round 5: exact PD_TV arithmetic with a wave-uniform skip of the correction chain where no lane is over the |p|^2 > 1 threshold
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5p; mkdir -p $O
cp tomobar_amd/libtomo_mi355x.so /tmp/tree.so
for rep in 1 2; do
  for L in ab/lib_a_base.so ab/lib_exact_skip.so; do
    n=$(basename $L .so); cp $L tomobar_amd/libtomo_mi355x.so
    for mode in "--exact-tv" ""; do
      timeout 300 python bench.py $mode --steps 3 --warmup 1 --no-cpu --no-pmc > $O/b.json 2>/dev/null
      python - "$n" "$mode" >> $O/summary.txt <<'PY'
import json, sys
d = json.load(open("gpurun_out/r5p/b.json"))
print(f"{sys.argv[1]:16s} {sys.argv[2] or 'default (relaxed)':18s}: {d['value']:.4f} it/s  PD_TV {d['kernels']['pdtv']['avg_ms']:.3f} ms  fast block {d['placement']['fast']}")
PY
    done
    TOMO_MI355X_FLAVOUR=shipped timeout 200 python tools/archive/probes/pd_time.py 1024 3 2>/dev/null | grep "exact\|default" | sed "s/^/$n  uniform-random volume: /" >> $O/summary.txt
  done
done
cp ab/lib_exact_skip.so tomobar_amd/libtomo_mi355x.so
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_recon.py tests/test_gpu_fullsize.py -m gpu -q -k "pd or tv or PD or fista or FISTA or osem or admm" 2>&1 | tail -3 >> $O/summary.txt
cp /tmp/tree.so tomobar_amd/libtomo_mi355x.so
cat $O/summary.txt
