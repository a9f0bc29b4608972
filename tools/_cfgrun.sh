for c in 0 1 2; do echo "DL_GEMM_BIGCFG=$c"; DL_GEMM_BIGCFG=$c timeout 100 python tools/epi_bench.py 2>&1 | grep "^("; done
DL_GEMM_BIGCFG=1 timeout 200 python -m pytest tests/test_kernels_gpu.py -x -q -k "big" 2>&1 | tail -2
DL_GEMM_BIGCFG=2 timeout 200 python -m pytest tests/test_kernels_gpu.py -x -q -k "big" 2>&1 | tail -2
