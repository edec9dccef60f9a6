# round 6: fuzz on the final tree -- driver level (now with the ring / Huber / Student's-t / SWLS data terms in the mix) and kernel level
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6k; mkdir -p $O
timeout 800 python tools/fuzz_drivers.py --minutes 9 --seed0 20000 > $O/fuzz_drivers.txt 2>&1
timeout 500 python tools/fuzz_drivers.py --minutes 5 --seed0 40000 --scale 3 > $O/fuzz_drivers_scale3.txt 2>&1
timeout 900 python tools/fuzz_campaign.py --minutes 8 --seed0 5000 > $O/fuzz_campaign.txt 2>&1
tail -3 $O/fuzz_drivers.txt | cut -c1-400; tail -3 $O/fuzz_drivers_scale3.txt | cut -c1-400; tail -3 $O/fuzz_campaign.txt | cut -c1-400
