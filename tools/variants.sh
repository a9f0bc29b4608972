timeout 300 python tools/soak.py DrugLAMP2C2P 2>&1 | tail -5
for b in 64 128 512; do python bench.py --batch $b --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200; done
python bench.py --dtype f32 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200
python bench.py --model DrugLAMPwoLLM --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200
python bench.py --model DrugLAMP2C2P --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200
timeout 200 python tools/ssl_cm_step.py 256 DrugLAMP2C2P 2>&1 | tail -2
