# round 5: what FETCH_SIZE counts -- fabric read requests by size (32 / 64 / 128 B) for the kernels whose traffic the bench reports
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5h; mkdir -p $O
rocprofv3 -L 2>/dev/null | grep -o "TCC_EA0_WR[A-Za-z0-9_]*" | sort -u | tr '\n' ' ' > $O/wr_counters.txt
PMC_PD_ITERS=9 PMC_GROUPS="FETCH_SIZE;WRITE_SIZE;TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum;TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" bash tools/pmc_run.sh r5h pdtv0 roftv bpq bpp bp0 fpq 2>&1 | grep -v native > $O/pmc_request_sizes.txt
cut -c1-420 $O/pmc_request_sizes.txt
cat $O/wr_counters.txt
