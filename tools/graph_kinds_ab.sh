#!/bin/bash
# same-box A/B: SSL / CM step kinds at batch 256, eager (side streams) against hipGraph replay
ROOT=$(cd "$(dirname "$0")/.." && pwd); OUT=$ROOT/gpurun_out/r5j; mkdir -p "$OUT"; cd "$ROOT"
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "transpose or view" 2>&1 | tail -3 > "$OUT/tests.log"
python bench.py --steps 30 --no-cpu-baseline --no-kernel-timing > /dev/null 2>&1
for rep in 1 2; do
for g in off on; do
  python bench.py --epoch 5 --graph $g --steps 60 --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 > "$OUT/ssl_${g}_$rep.json"
  python bench.py --model DrugLAMP2C2P --epoch 6 --graph $g --steps 60 --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 > "$OUT/cm_${g}_$rep.json"
  python bench.py --model DrugLAMP2C2P --epoch 10 --graph $g --steps 60 --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 > "$OUT/sslcm_${g}_$rep.json"
done; done
cat "$OUT/tests.log"; for f in "$OUT"/*.json; do python -c "import json,sys; d=json.loads(open('$f').read()); print('$f'.split('/')[-1], d['ms_per_step'], d['config'].get('hip_graph'), d['config'].get('hip_graphs_live'), d['config'].get('hip_graph_captures_in_timed_region'))"; done
