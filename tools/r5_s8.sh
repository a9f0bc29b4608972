#!/bin/bash
O=gpurun_out/r5; mkdir -p $O
python -m pytest tests/test_kernels_gpu.py -k "gate_dpre" tests/test_parity_gpu.py -q -m gpu -x 2>&1 | tail -4 > $O/s8_tests.log
python bench.py --no-cpu-baseline --no-kernel-timing --steps 30 > /dev/null 2>&1
for i in 1 2; do
python bench.py --no-cpu-baseline --no-kernel-timing --steps 150 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('new ', d['ms_per_step'])" >> $O/s8_ab.log
DL_GATE_DPRE=0 python bench.py --no-cpu-baseline --no-kernel-timing --steps 150 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('old ', d['ms_per_step'])" >> $O/s8_ab.log
done
python bench.py --no-cpu-baseline --no-kernel-timing --steps 300 --batch 32 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('new32 ', d['ms_per_step'])" >> $O/s8_ab.log
DL_GATE_DPRE=0 python bench.py --no-cpu-baseline --no-kernel-timing --steps 300 --batch 32 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('old32 ', d['ms_per_step'])" >> $O/s8_ab.log
