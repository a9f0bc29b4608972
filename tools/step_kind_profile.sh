#!/bin/bash
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r5i; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export DL_BRANCH_STREAMS=0
rocprofv3 --kernel-trace --output-format csv -d "$OUT/t1" -o t -- python3 "$ROOT/bench.py" --epoch 5 --steps 30 --warmup 3 --no-cpu-baseline --no-kernel-timing > "$OUT/ssl.log" 2>&1
T=$(find "$OUT/t1" -name '*kernel_trace.csv' | head -1); python3 "$ROOT/tools/prof_summary.py" "$T" 0.4 > "$OUT/ssl_step_kernel_summary.txt" 2>&1; rm -rf "$OUT/t1"
rocprofv3 --kernel-trace --output-format csv -d "$OUT/t2" -o t -- python3 "$ROOT/bench.py" --model DrugLAMP2C2P --epoch 6 --steps 30 --warmup 3 --no-cpu-baseline --no-kernel-timing > "$OUT/cm.log" 2>&1
T=$(find "$OUT/t2" -name '*kernel_trace.csv' | head -1); python3 "$ROOT/tools/prof_summary.py" "$T" 0.4 > "$OUT/cm_step_kernel_summary.txt" 2>&1; rm -rf "$OUT/t2"
head -70 "$OUT/ssl_step_kernel_summary.txt" | cut -c1-200
