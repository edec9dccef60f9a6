# round 4, call H: compiler scheduling strategies for tv_kernels.hip, same-box A/B of the PD_TV launch (ab/lib_base.so = shipped
# flags; S1 = -amdgpu-sched-strategy=max-ilp, S2 = max-memory-clause, S3 = -amdgpu-schedule-metric-bias=0, S4 = -amdgpu-use-amdgpu-trackers)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
T=${1:-r4h}; O=gpurun_out/$T; mkdir -p $O
bash tools/run_ab.sh ${T}_ab python tools/archive/probes/pd_time.py 1024 3 > $O/pd_sched_ab.txt 2>&1
grep -v amdgpu $O/pd_sched_ab.txt
