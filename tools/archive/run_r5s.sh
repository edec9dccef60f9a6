# round 5: fuzz campaign over seeds outside the suite (array_equal against the oracle)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5s; mkdir -p $O
timeout 1500 python tools/fuzz_campaign.py --minutes ${FUZZ_MINUTES:-10} --seed0 ${FUZZ_SEED0:-100} ${FUZZ_ARGS:-} 2>&1 | grep -v "amdgpu.ids" > $O/fuzz${FUZZ_TAG:-}.txt
tail -12 $O/fuzz${FUZZ_TAG:-}.txt
