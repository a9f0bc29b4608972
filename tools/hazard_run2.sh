# Round 6, hazard experiment 2: (a) a standalone reproducer of the shift -> v_pk sequence; (b) the real LayerNorm backward, SLP build, with a
# wait state behind every unpacking shift (libdruglamp_hip_slpnop.so) against the plain SLP build, 240 launches each under same-process load
mkdir -p gpurun_out/r6g
O=gpurun_out/r6g/pk_forward.txt
( cd tools/micro && timeout 600 ./pk_forward ) > $O 2>&1; cat $O
P=gpurun_out/r6g/ln_nop.txt; : > $P
for lib in libdruglamp_hip_slpnorm.so libdruglamp_hip_slpnop.so; do
  for load in thread process; do
    echo "== $lib, load: $load, 240 launches per shape" >> $P
    DL_USE_STUDY_LIB=$lib CR_ONLY_LN=1 CR_LOAD=$load timeout 500 python tools/contention_repeat.py 240 2>&1 | grep "mismatching" >> $P
  done
done
cat $P
