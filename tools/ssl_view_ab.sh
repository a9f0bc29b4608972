#!/bin/bash
# same-box A/B: SSL-epoch step with the site_len-1 transpose kernel (product) against the general site-pooling kernel (variant notv)
ROOT=$(cd "$(dirname "$0")/.." && pwd); OUT=$ROOT/gpurun_out/r5h; mkdir -p "$OUT"; cd "$ROOT"
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_parity_gpu.py -q -x 2>&1 | tail -5 > "$OUT/tests.log"
python bench.py --steps 30 --no-cpu-baseline --no-kernel-timing > /dev/null 2>&1
for i in 1 2; do
  python bench.py --epoch 5 --steps 60 --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 > "$OUT/ssl_new_$i.json"
  DL_USE_STUDY_LIB=libdruglamp_hip_notv.so python bench.py --epoch 5 --steps 60 --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 > "$OUT/ssl_notv_$i.json"
done
python bench.py --model DrugLAMP2C2P --epoch 10 --steps 60 --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 > "$OUT/2c2p_e10_new.json"
python bench.py --epoch 5 --batch 32 --steps 200 --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 > "$OUT/ssl_b32_new.json"
cat "$OUT/tests.log"; for f in "$OUT"/*.json; do echo $f; python -c "import json,sys; d=json.loads(open('$f').read()); print(d['ms_per_step'])"; done
