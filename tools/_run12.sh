mkdir -p gpurun_out/r6k
CR_LOAD=process timeout 1500 python tools/contention_repeat.py 200 2>&1 | grep -v amdgpu.ids > gpurun_out/r6k/contention_200.txt
cat gpurun_out/r6k/contention_200.txt
