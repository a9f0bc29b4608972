bash tools/same_box_vs_commit.sh run > gpurun_out/same_box_r5_vs_r6.txt 2>&1
cat gpurun_out/same_box_r5_vs_r6.txt
bash tools/same_box_vs_commit.sh run --batch 32 --steps 200 --no-cpu-baseline --no-kernel-timing > gpurun_out/same_box_r5_vs_r6_batch32.txt 2>&1
cat gpurun_out/same_box_r5_vs_r6_batch32.txt
( time bash tools/refresh_profiles.sh ) > gpurun_out/refresh.log 2>&1
tail -30 gpurun_out/refresh.log
