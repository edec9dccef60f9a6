# round 5: driver-level fuzz campaign (RecToolsIRCuPy.FISTA / ADMM / OSEM vs the oracle's loops, array_equal)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5v; mkdir -p $O
timeout 1500 python tools/fuzz_drivers.py --minutes ${FUZZ_MINUTES:-8} --seed0 ${FUZZ_SEED0:-0} ${FUZZ_ARGS:-} 2>&1 | grep -v "amdgpu.ids" > $O/fuzz_drivers${FUZZ_TAG:-}.txt
tail -25 $O/fuzz_drivers${FUZZ_TAG:-}.txt | cut -c1-700
