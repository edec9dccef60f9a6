# round 5: GPU suite on the final sources + refresh of profiles/pmc_traffic.json (proj_kernels.hip changed after r5z: relay allocation policy)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5j; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $O/pytest.log; cat $O/pytest.log
bash tools/run_pmc_refresh.sh r5zb
