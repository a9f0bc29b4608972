mkdir -p gpurun_out/r6f
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x 2>&1 | tail -6 > gpurun_out/r6f/pytest_kernels.txt
cat gpurun_out/r6f/pytest_kernels.txt
timeout 600 python tools/epi_bench.py > gpurun_out/r6f/epi_bench.txt 2>&1; cat gpurun_out/r6f/epi_bench.txt
timeout 600 python bench.py --steps 100 --no-cpu-baseline --no-projection 2>/dev/null | tail -1 > gpurun_out/r6f/bench_line.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r6f/bench_line.json')); r=d['roofline']
print(d['ms_per_step'], r['frac'], {k:v.get('ms_per_step') for k,v in r['sub_families'].items()})
PY
