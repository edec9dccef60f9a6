cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
T=${1:-r4k}; O=gpurun_out/$T; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -8 > $O/pytest.log
timeout 900 python bench.py --config cfg3 --steps 3 --warmup 1 > $O/bench_cfg3_full.json 2> $O/bench_cfg3_full.err
timeout 600 python bench.py --config cfg3-share --steps 2 --warmup 1 > $O/bench_cfg3_share.json 2> $O/bench_cfg3_share.err
timeout 300 python tools/fbp_bench.py > $O/fbp_bench.txt 2>&1
timeout 300 python tools/archive/probes/ir_methods_probe.py > $O/ir_methods.txt 2>&1
cat $O/pytest.log; for f in bench_cfg3_full bench_cfg3_share; do python -c "
import json;d=json.load(open('$O/$f.json'));print('$f', round(d['value'],4), round(d['ms_per_step'],1), {k:round(v['avg_ms'],2) for k,v in d['kernels'].items()}, d['roofline']['kernel'], d['roofline'].get('traffic'))"; done; grep -v amdgpu $O/fbp_bench.txt; grep -v amdgpu $O/ir_methods.txt
