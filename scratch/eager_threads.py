"""How the PyTorch-eager config-1 baseline scales with torch threads on the GPU box's host (bench.py: eager_config1)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import eager_config1
for th in (16, 32, 64, 128, os.cpu_count()):
    r = eager_config1(reps=2, warm=1, budget_s=30.0, threads=th)
    print(th, json.dumps({k: r.get(k) for k in ("median_s", "min_s", "threads", "reps")}), flush=True)
