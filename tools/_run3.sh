mkdir -p gpurun_out/r6c
timeout 1200 python tools/trickle_parts.py > gpurun_out/r6c/trickle_parts.txt 2>&1
cat gpurun_out/r6c/trickle_parts.txt
