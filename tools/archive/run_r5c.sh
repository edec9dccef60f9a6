# round 5: CuPy-surface + slab tests on the GPU; FP breakdown on the configs[3] share; placement hit rate over 12 processes;
# one-GPU dry runs of the 2-rank paths
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5c; mkdir -p $O
timeout 600 python -m pytest tests/test_cupy_surface.py tests/test_gpu_slab.py tests/test_gpu_slab_fista.py tests/test_host_logic.py -x -q > $O/pytest_subset.log 2>&1; echo "pytest rc $?" >> $O/pytest_subset.log
tail -5 $O/pytest_subset.log
# ---- FP on the configs[3] share: staging / LDS writes / sampling split, then counters of the same call
timeout 600 python tools/archive/probes/fp_stage_probe.py 2048 256 1500 1 2>/dev/null | grep -v amdgpu > $O/fp_stage_probe_cfg3_share.txt
cat $O/fp_stage_probe_cfg3_share.txt
PMC_N=2048 PMC_NZ=256 PMC_NA=1500 PMC_TIMEOUT=240 PMC_GROUPS="SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES;GRBM_GUI_ACTIVE;FETCH_SIZE;WRITE_SIZE;SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES;TCC_HIT_sum TCC_MISS_sum" bash tools/pmc_run.sh r5c_fp fp > $O/pmc_fp_cfg3_share.txt 2>&1
grep -v native $O/pmc_fp_cfg3_share.txt | tail -8
# ---- placement: P(a block of the fast class within the tries) over 12 processes
rm -f $O/placement_hit_rate.txt
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do
  timeout 200 python - >> $O/placement_hit_rate.txt 2>/dev/null <<'PY'
import sys, time, torch
sys.path.insert(0, ".")
from tomobar_amd import ops
from tomobar_amd.regularisersCuPy import PD_TV_cupy
vol = torch.rand((1024, 1024, 1024), device="cuda"); out = torch.empty_like(vol)
t0 = time.perf_counter(); ops.reserve_tv_scratch((1024, 1024, 1024), "cuda:0", "PD_TV", False); torch.cuda.synchronize(); search = time.perf_counter() - t0
PD_TV_cupy(vol, 0.01, 30, 0, 1, 12.0, 0, False, out=out); torch.cuda.synchronize()
ts = []
for _ in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); PD_TV_cupy(vol, 0.01, 30, 0, 1, 12.0, 0, False, out=out); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 10)
p = ops.placement_last()
print(f"{min(ts):7.3f} ms per launch; search {search:5.2f} s; tries {len(p['scores_GBps']):2d}; fast {str(p['fast']):5s}; chosen {p['chosen']}; scores {p['scores_GBps']}")
PY
done
cat $O/placement_hit_rate.txt
# ---- the 2-rank paths as functional dry runs on one GPU (gloo, host-staged halos)
timeout 300 python tools/rccl_preflight.py --gpus 2 --n 1024 --nz 64 --reps 3 > $O/rccl_preflight_dryrun.json 2> $O/rccl_preflight_dryrun.err; echo "preflight rc $?"
timeout 600 python bench.py --gpus 2 --strong --n 512 --nz 128 --angles 360 --steps 2 --warmup 1 > $O/bench_2ranks_dryrun.json 2> $O/bench_2ranks_dryrun.err; echo "bench 2 ranks rc $?"
BENCH_NORTH_STAR_TEST=1 timeout 600 python bench.py --gpus 2 --strong --n 512 --nz 128 --angles 360 --steps 2 --warmup 1 > $O/bench_2ranks_northstar_dryrun.json 2> $O/bench_2ranks_northstar_dryrun.err; echo "bench 2 ranks + north-star block rc $?"
cut -c1-300 $O/rccl_preflight_dryrun.json $O/bench_2ranks_dryrun.json $O/bench_2ranks_northstar_dryrun.json
