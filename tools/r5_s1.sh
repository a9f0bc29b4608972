#!/bin/bash
# round-5 session 1: new tests, bench with distinct batches (A/B against one static batch), full GPU suite
set -x
O=gpurun_out/r5; mkdir -p $O
python -m pytest tests/test_protein_cnn_compact_gpu.py tests/test_graph_step_gpu.py -x -q -m gpu 2>&1 | tail -15 > $O/s1_tests_new.log
python bench.py --no-cpu-baseline --no-kernel-timing --steps 30 > /dev/null 2>&1   # throwaway first process
python bench.py --no-cpu-baseline --steps 200 > $O/s1_bench_256.json 2> $O/s1_bench_256.err
python bench.py --no-cpu-baseline --no-kernel-timing --steps 200 --distinct-batches 1 > $O/s1_bench_256_static.json 2>> $O/s1_bench_256.err
python bench.py --no-cpu-baseline --no-kernel-timing --steps 300 --batch 32 > $O/s1_bench_32.json 2> $O/s1_bench_32.err
python bench.py --no-cpu-baseline --no-kernel-timing --steps 300 --batch 32 --distinct-batches 1 > $O/s1_bench_32_static.json 2>> $O/s1_bench_32.err
python bench.py --no-cpu-baseline --no-kernel-timing --steps 200 --batch 64 > $O/s1_bench_64.json 2> $O/s1_bench_64.err
python bench.py --no-cpu-baseline --no-kernel-timing --steps 200 --batch 128 > $O/s1_bench_128.json 2> $O/s1_bench_128.err
python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > $O/s1_tests_all.log
