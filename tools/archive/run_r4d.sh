# round 4, call D: same-box A/B of the headline with exact (default) vs relaxed PD_TV arithmetic; the one-workgroup-per-CU
# probe of the K = 3 PD_TV kernel (what an LDS-DMA prefetch tiling would have to live with); LDS probe with BP's VALU:LDS ratio;
# one-Newton-step reciprocal; 2D TV timing; preflight dry run.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
T=${1:-r4d}; O=gpurun_out/$T; mkdir -p $O
for rep in 1 2; do
  timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu > $O/bench_exact_$rep.json 2> $O/bench_exact_$rep.err
  timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu --relaxed-tv > $O/bench_relaxed_$rep.json 2> $O/bench_relaxed_$rep.err
done
timeout 600 python tools/archive/probes/pd_halo_probe.py 1024 30 0 > $O/pd_probes_exact.txt 2>&1
timeout 600 python tools/archive/probes/pd_halo_probe.py 1024 30 3 > $O/pd_probes_relaxed.txt 2>&1
timeout 300 tools/probes/_build/lds_rate_probe > $O/lds_rate_probe.txt 2>&1
timeout 300 tools/probes/_build/recip_allones_probe > $O/recip_allones_probe.txt 2>&1
timeout 300 python tools/tv2d_bench.py > $O/tv2d_bench.txt 2>&1
timeout 300 python tools/rccl_preflight.py --gpus 2 --n 1024 --nz 64 --reps 3 > $O/rccl_preflight_dryrun.json 2> $O/rccl_preflight_dryrun.err
for f in bench_exact_1 bench_relaxed_1 bench_exact_2 bench_relaxed_2; do python -c "
import json;d=json.load(open('$O/$f.json'));print('$f', round(d['value'],4), {k:round(v['avg_ms'],3) for k,v in d['kernels'].items()})"; done
grep -v amdgpu $O/pd_probes_exact.txt; grep -v amdgpu $O/pd_probes_relaxed.txt; grep "BP mix\|reads only, stride 1 " $O/lds_rate_probe.txt; cat $O/recip_allones_probe.txt; grep -v amdgpu $O/tv2d_bench.txt; cut -c1-900 $O/rccl_preflight_dryrun.json; tail -2 $O/rccl_preflight_dryrun.err
