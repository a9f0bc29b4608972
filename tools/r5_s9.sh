#!/bin/bash
ROOT=$(pwd); O=$ROOT/gpurun_out/r5; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export DL_BRANCH_STREAMS=0
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trace1 -o t -- python3 "$ROOT/bench.py" --epoch 5 --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing > $O/s9_ssl_under_rocprof.log 2>&1
S1=$(find /tmp/trace1 -name '*kernel_stats.csv' | head -1); cp "$S1" $O/s9_ssl_kernel_stats_one_stream.csv
unset DL_BRANCH_STREAMS
python3 "$ROOT/bench.py" --epoch 5 --steps 50 --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 > $O/s9_bench_ssl_256.json
python3 "$ROOT/bench.py" --epoch 5 --steps 100 --batch 32 --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 > $O/s9_bench_ssl_32.json
