# round 6: forward projector: wave-uniform row index made provable (no waterfall loops around the staging loads of the
# dense-angle form): parity, then the bench lines where the forward projector weighs most
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6f; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_shares.py tests/test_gpu_recon.py -m gpu -q -x 2>&1 | tail -8 > $O/pytest.log
timeout 900 python tools/fuzz_campaign.py > $O/fuzz_campaign.txt 2>&1
timeout 300 python tools/kernel_bench.py 1024 1024 75 3 > $O/kernel_bench_1024.txt 2>&1
timeout 900 python bench.py --config cfg3-share --steps 3 --warmup 1 --no-cpu > $O/bench_cfg3_share.json 2> $O/bench_cfg3_share.err
timeout 1500 python bench.py --config cfg3 --steps 3 --warmup 1 --no-cpu > $O/bench_cfg3_full.json 2> $O/bench_cfg3_full.err
timeout 900 python bench.py --config cfg5-share --steps 2 --warmup 1 --no-cpu > $O/bench_cfg5_share.json 2> $O/bench_cfg5_share.err
timeout 600 python bench.py --config cfg1 --steps 50 --warmup 5 --no-cpu > $O/bench_cfg1.json 2> $O/bench_cfg1.err
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu --no-pmc > $O/bench_10_steps.json 2> $O/bench_10_steps.err
tail -5 $O/pytest.log; tail -5 $O/fuzz_campaign.txt; grep -E "^FP|^BP" $O/kernel_bench_1024.txt
python - <<'PY'
import json
for f in ("bench_cfg3_share","bench_cfg3_full","bench_cfg5_share","bench_cfg1","bench_10_steps"):
    try:
        l=json.load(open(f"gpurun_out/r6f/{f}.json"))
        print(f, round(l["value"],4), "it/s", round(l["ms_per_step"],1), "ms", {k:round(v["avg_ms"],2) for k,v in l["kernels"].items()})
    except Exception as e: print(f, "failed", e)
PY
