mkdir -p gpurun_out/r6j
P=gpurun_out/r6j/ln_nopk.txt; : > $P
for lib in libdruglamp_hip_slpnopk.so libdruglamp_hip_slpnorm.so; do
  echo "== $lib, load: process, 480 launches per shape" >> $P
  DL_USE_STUDY_LIB=$lib CR_ONLY_LN=1 CR_LOAD=process timeout 900 python tools/contention_repeat.py 480 2>&1 | grep "mismatching" >> $P
done
cat $P
