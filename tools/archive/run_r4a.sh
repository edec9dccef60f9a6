# round 4, call A: the two ceiling probes (HBM copy, ds_read_b128 issue), the per-kernel A/B of the PD_TV variants, and the
# whole BASELINE configs[3] on ONE GPU (the N = 1 point of the 2048^2-class curve).  usage: bash tools/run_r4a.sh [tag]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
T=${1:-r4a}; O=gpurun_out/$T; mkdir -p $O
timeout 600 tools/probes/_build/hbm_copy_probe > $O/hbm_copy_probe.txt 2>&1
timeout 300 tools/probes/_build/lds_rate_probe > $O/lds_rate_probe.txt 2>&1
timeout 300 python tools/archive/probes/stream_probe.py > $O/stream_probe.txt 2>&1
timeout 400 python tools/kernel_bench.py 1024 1024 75 3 > $O/kernel_bench_1024.txt 2>&1
timeout 1500 python bench.py --config cfg3 --gpus 1 --steps 3 --warmup 1 > $O/bench_cfg3_full.json 2> $O/bench_cfg3_full.err
rocm-smi --showmeminfo vram > $O/vram_after.txt 2>&1
cat $O/hbm_copy_probe.txt; cat $O/lds_rate_probe.txt; grep -v amdgpu.ids $O/kernel_bench_1024.txt; cut -c1-400 $O/bench_cfg3_full.json; tail -3 $O/bench_cfg3_full.err
