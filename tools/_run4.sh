set -x
mkdir -p gpurun_out/r6d
( time timeout 2400 python -m pytest tests -m gpu -q -x ) > gpurun_out/r6d/pytest_gpu.txt 2>&1
tail -5 gpurun_out/r6d/pytest_gpu.txt
# cm and ssl+cm steps at batch 256, >= 2000 UNSYNCHRONISED replays, 8 distinct batches (the trainer's own bound switched off)
for cfg in "DrugLAMP2C2P 6" "DrugLAMP2C2P 10"; do
  echo "== 2000 $cfg 256 8 (DL_GRAPH_SYNC_EVERY=0 SYNC_EVERY=0)" >> gpurun_out/r6d/soak.txt
  DL_GRAPH_SYNC_EVERY=0 NSTEPS=2000 SYNC_EVERY=0 GRAPH=1 timeout 600 python tools/graph_nosync_soak.py $cfg 256 8 2>&1 | tail -2 >> gpurun_out/r6d/soak.txt
done
echo "== 600 DrugLAMP2C2P 10 256 8 (defaults: one device sync per 256 replays)" >> gpurun_out/r6d/soak.txt
NSTEPS=600 SYNC_EVERY=0 GRAPH=1 timeout 600 python tools/graph_nosync_soak.py DrugLAMP2C2P 10 256 8 2>&1 | tail -2 >> gpurun_out/r6d/soak.txt
cat gpurun_out/r6d/soak.txt
bash tools/northstar_profile.sh > /dev/null 2>&1; cp gpurun_out/northstar/summary.txt gpurun_out/r6d/northstar_summary.txt
bash tools/attn_pmc.sh paired > gpurun_out/r6d/attn_paired_sq.txt 2>&1
tail -40 gpurun_out/r6d/attn_paired_sq.txt
python bench.py --steps 100 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r6d/bench_line.json
head -c 300 gpurun_out/r6d/bench_line.json
