#!/bin/bash
# round-5 session 3: bisect of the bn.hip no-SLP failure; new tests (key multiplicities); bench A/B of DL_KEY_COMPACT
O=gpurun_out/r5; mkdir -p $O
python tools/bn_bisect.py rec /tmp/bn_a.pt > $O/s3_bisect.log 2>&1
DL_USE_STUDY_LIB=libdruglamp_hip_noslpbn.so python tools/bn_bisect.py rec /tmp/bn_b.pt >> $O/s3_bisect.log 2>&1
python tools/bn_bisect.py cmp /tmp/bn_a.pt /tmp/bn_b.pt >> $O/s3_bisect.log 2>&1
python -m pytest tests/test_kernels_gpu.py -k "multiplicities" tests/test_parity_gpu.py tests/test_model_gpu.py -q -m gpu -x 2>&1 | tail -25 > $O/s3_tests_new.log
python bench.py --no-cpu-baseline --no-kernel-timing --steps 30 > /dev/null 2>&1
python bench.py --no-cpu-baseline --no-kernel-timing --steps 200 > $O/s3_bench_256.json 2> $O/s3_bench.err
DL_KEY_COMPACT=0 python bench.py --no-cpu-baseline --no-kernel-timing --steps 200 > $O/s3_bench_256_fullkeys.json 2>> $O/s3_bench.err
python bench.py --no-cpu-baseline --no-kernel-timing --steps 300 --batch 32 > $O/s3_bench_32.json 2>> $O/s3_bench.err
DL_KEY_COMPACT=0 python bench.py --no-cpu-baseline --no-kernel-timing --steps 300 --batch 32 > $O/s3_bench_32_fullkeys.json 2>> $O/s3_bench.err
python -m pytest tests -q -m gpu 2>&1 | tail -15 > $O/s3_tests_all.log
