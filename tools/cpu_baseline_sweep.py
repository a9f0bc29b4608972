"""bench.py's cpu_baseline leg (the CPU oracle's training step on the host cores) at batch 16 / 32 / 64 — SURVEY D10 asks
for the reference-side timing at the strong-scaling batches as well (since round 3 the same sweep is part of the bench
line itself).  One JSON object."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
print(json.dumps(bench.cpu_baseline([int(a) for a in sys.argv[1:]] or [16, 32, 64], 3, budget_s=60.0)), flush=True)
