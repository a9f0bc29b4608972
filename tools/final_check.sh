mkdir -p gpurun_out/r6z
( time timeout 2400 python -m pytest tests -x -q -m gpu ) > gpurun_out/r6z/pytest_gpu.txt 2>&1
tail -5 gpurun_out/r6z/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -2
python bench.py --steps 30 --no-cpu-baseline --no-kernel-timing > /dev/null 2>&1
python bench.py 2>/dev/null | tail -1 > gpurun_out/r6z/bench_line.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r6z/bench_line.json')); r=d['roofline']
print(d['ms_per_step'], d['value'], r['frac'], r['traffic'], r['sub_families']['ffn']['ms_per_step'], d['projected_strong_scaling']['8'])
PY
