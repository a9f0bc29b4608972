"""bench.py's cpu_baseline leg (the CPU oracle's training step on the host cores) at batch 16 / 32 / 64 — SURVEY D10 asks
for the reference-side timing at the strong-scaling batches as well.  One JSON object per line."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
for b in [int(a) for a in sys.argv[1:]] or [16, 32, 64]:
    r = bench.cpu_baseline(b, 3, budget_s=20.0)
    r["batch"] = b
    print(json.dumps(r), flush=True)
