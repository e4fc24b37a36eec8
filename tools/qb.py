"""one line of bench.py's result: python tools/qb.py [bench args]  (environment knobs apply)"""
import json
import os
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline", "--no-end-to-end"] + sys.argv[1:],
                     capture_output=True, text=True)
lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
if not lines:
    print("FAILED", out.stderr[-500:])
    sys.exit(1)
d = json.loads(lines[-1])
k = d.get("kernels", {}).get("ms_per_step")
print(os.environ.get("QB_TAG", ""), d["value"], d["ms_per_step"], "hash avg ms", d["roofline"]["avg_launch_ms"], "mx", d["config"]["minimizers"],
      "edges", d["config"]["edges"], k if k else "", flush=True)
