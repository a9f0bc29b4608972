#!/bin/bash
O=gpurun_out/r5; mkdir -p $O
python -m pytest tests/test_kernels_gpu.py -k "cross_entropy or batch_norm_relu or site_pool_kernel or gate_dpre" -q -m gpu 2>&1 | tail -12 > $O/s11_tests_k.log
python -m pytest tests/test_model_gpu.py tests/test_parity_gpu.py tests/test_graph_step_gpu.py -q -m gpu 2>&1 | tail -12 > $O/s11_tests_m.log
python bench.py --no-cpu-baseline --no-kernel-timing --steps 30 > /dev/null 2>&1
python bench.py --epoch 5 --steps 50 --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 > $O/s11_bench_ssl_256.json
python bench.py --epoch 5 --steps 100 --batch 32 --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 > $O/s11_bench_ssl_32.json
python bench.py --model DrugLAMP2C2P --epoch 10 --steps 50 --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 > $O/s11_bench_2c2p_e10_256.json
python tools/torch_glue_profile.py 256 5 > $O/s11_glue256_ssl.txt 2>&1
