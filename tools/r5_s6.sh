#!/bin/bash
O=gpurun_out/r5; mkdir -p $O
python -m pytest tests/test_parity_gpu.py tests/test_model_gpu.py tests/test_kernels_gpu.py -q -m gpu -x 2>&1 | tail -8 > $O/s6_tests.log
python bench.py --no-cpu-baseline --no-kernel-timing --steps 30 > /dev/null 2>&1
python bench.py --no-cpu-baseline --steps 200 > $O/s6_bench_256.json 2> $O/s6_bench.err
python bench.py --no-cpu-baseline --no-kernel-timing --steps 300 --batch 32 > $O/s6_bench_32.json 2>> $O/s6_bench.err
python tools/gemm_shapes.py > $O/s6_gemm_shapes.txt 2>&1
