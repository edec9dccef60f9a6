# round 6: dword LDS-DMA probe (destination bases above 64 KiB, zero fill of out-of-range lanes), the residual-layout check test
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6i; mkdir -p $O
timeout 120 tools/probes/_build/lds_dma_dword_probe > $O/lds_dma_dword_probe.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "residual_buffer_must_match or quad_interleaved" 2>&1 | tail -4 > $O/pytest.log
cat $O/lds_dma_dword_probe.txt; cat $O/pytest.log
