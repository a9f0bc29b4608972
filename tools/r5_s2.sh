#!/bin/bash
# round-5 session 2: bn.hip SLP / no-SLP against the fp64 oracle, then the full GPU suite
O=gpurun_out/r5; mkdir -p $O
python -m pytest tests/test_model_gpu.py -k "gcn" -q -s 2>&1 | grep -v Warning | tail -12 > $O/s2_gcn_slp.log
DL_USE_STUDY_LIB=libdruglamp_hip_noslpbn.so python -m pytest tests/test_model_gpu.py -k "gcn" -q -s 2>&1 | grep -v Warning | tail -40 > $O/s2_gcn_noslp.log
python -m pytest tests -q -m gpu 2>&1 | tail -15 > $O/s2_tests_all.log
