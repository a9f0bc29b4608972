#!/bin/bash
O=gpurun_out/r5; mkdir -p $O
python -m pytest tests/test_kernels_gpu.py -k "cross_entropy" tests/test_bench_contract_gpu.py -q -m gpu 2>&1 | tail -6 > $O/s12_tests.log
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/s12_driver_cmd.json 2> $O/s12_driver_cmd.err
python bench.py --epoch 5 --steps 50 --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 > $O/s12_bench_ssl_256.json
python bench.py --model DrugLAMP2C2P --epoch 6 --steps 50 --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 > $O/s12_bench_2c2p_e6_256.json
python bench.py --model DrugLAMP2C2P --epoch 10 --steps 50 --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 > $O/s12_bench_2c2p_e10_256.json
