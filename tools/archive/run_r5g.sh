# round 5: L2 behaviour of the back projector: quad residual + LDS-DMA staging (shipped) vs planar staging (bp variant 3)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5g; mkdir -p $O
rocprofv3 -L 2>/dev/null | grep -o "TCC_[A-Za-z0-9_]*" | sort -u | tr '\n' ' ' > $O/tcc_counters.txt
PMC_GROUPS="FETCH_SIZE;WRITE_SIZE;TCC_HIT_sum TCC_MISS_sum;TCC_REQ_sum TCC_READ_sum;TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum;SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" bash tools/pmc_run.sh r5g bpq bpp 2>&1 | grep -v native > $O/pmc_bp.txt
cat $O/pmc_bp.txt | cut -c1-700
cut -c1-1500 $O/tcc_counters.txt
