for d in 0 1 2 3; do echo "DL_GEMM_DBG=$d"; DL_GEMM_DBG=$d timeout 100 python tools/pitch_bench.py 2>&1 | grep "^(" | cut -c1-60; done
