#!/bin/bash
# kernel traces: replayed batch-32 steps; one-stream batch-256 steps; new-test run
ROOT=$(pwd); O=$ROOT/gpurun_out/r5; mkdir -p $O
python -m pytest tests/test_model_gpu.py -k "distinct_drug_rows" -q -m gpu 2>&1 | tail -3 > $O/s5_test.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/trace32 -o t -- python3 "$ROOT/bench.py" --batch 32 --steps 100 --no-cpu-baseline --no-kernel-timing > $O/s5_b32_under_rocprof.log 2>&1
T32=$(find /tmp/trace32 -name '*kernel_trace.csv' | head -1)
python3 "$ROOT/tools/prof_summary.py" "$T32" 0.2 150 > $O/s5_batch32_kernel_summary.txt 2>&1
export DL_BRANCH_STREAMS=0
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trace1 -o t -- python3 "$ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing > $O/s5_under_rocprof_one_stream.log 2>&1
S1=$(find /tmp/trace1 -name '*kernel_stats.csv' | head -1); cp "$S1" $O/s5_kernel_stats_one_stream.csv
