mkdir -p gpurun_out/r6l
( time timeout 2400 python -m pytest tests -m gpu -q ) > gpurun_out/r6l/pytest_gpu.txt 2>&1
tail -6 gpurun_out/r6l/pytest_gpu.txt
python __graft_entry__.py --smoke 2>&1 | tail -2
python bench.py --steps 20 --warmup 3 2>/dev/null | tail -1 > gpurun_out/r6l/bench_driver_cmd.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r6l/bench_driver_cmd.json')); r=d['roofline']
print(d['ms_per_step'], d['value'], r['frac'], r['traffic'], r['traffic_passes'], r['traffic_source'][:80])
PY
