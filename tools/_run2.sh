set -x
mkdir -p gpurun_out/r6b
timeout 900 python -m pytest "tests/test_kernels_gpu.py::test_large_tile_gemm_is_bitwise_equal_to_the_128_tile_path" -m gpu -q -x 2>&1 | tail -15 > gpurun_out/r6b/pytest_bitwise.txt
cat gpurun_out/r6b/pytest_bitwise.txt
timeout 900 python tools/trickle_bench.py > gpurun_out/r6b/trickle_bench.txt 2>&1
cat gpurun_out/r6b/trickle_bench.txt
