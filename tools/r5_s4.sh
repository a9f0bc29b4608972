#!/bin/bash
O=gpurun_out/r5; mkdir -p $O
python tools/bn_bisect.py rec /tmp/bn_a.pt > $O/s4_bisect.log 2>&1
DL_USE_STUDY_LIB=libdruglamp_hip_noslpbn.so python tools/bn_bisect.py rec /tmp/bn_b.pt >> $O/s4_bisect.log 2>&1
python tools/bn_bisect.py cmp /tmp/bn_a.pt /tmp/bn_b.pt >> $O/s4_bisect.log 2>&1
python -m pytest tests/test_model_gpu.py -k "distinct_drug_rows" -q -m gpu -x 2>&1 | tail -60 > $O/s4_test.log
