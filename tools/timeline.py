"""timeline of ONE step out of a rocprofv3 kernel_trace.csv of bench.py: python tools/timeline.py <b_kernel_trace.csv> [step-from-the-end]
prints every kernel of the step with its queue, start (us from the step's first kernel), duration and the idle time of its
queue before it; then the GPU-busy / both-queues-busy / idle totals."""
import csv
import sys
import re

rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    r["n"] = re.sub(r"^(void )?mxg::", "", r["Kernel_Name"]).split("(")[0][:40]
rows.sort(key=lambda r: r["s"])
# a step ends with k_edges
ends = [i for i, r in enumerate(rows) if r["n"].startswith("k_edges")]
lo, hi = ends[-back - 1] + 1, ends[-back] + 1
step = rows[lo:hi]
t0 = step[0]["s"]
last_end = {}
print(f"step of {len(step)} kernels, {(step[-1]['e'] - t0) / 1e3:.1f} us")
for r in step:
    q = r["Queue_Id"]
    gap = (r["s"] - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = r["e"]
    print(f"q{q} {(r['s'] - t0) / 1e3:9.1f} +{(r['e'] - r['s']) / 1e3:8.1f}  gap {gap:7.1f}  {r['n']}  grid {r['Grid_Size_X']} wg {r['Workgroup_Size_X']}")
ev = sorted([(r["s"], 1) for r in step] + [(r["e"], -1) for r in step])
busy = {0: 0, 1: 0, 2: 0}
depth, prev = 0, ev[0][0]
for t, d in ev:
    busy[min(depth, 2)] += t - prev
    depth += d
    prev = t
print({k: round(v / 1e3, 1) for k, v in busy.items()}, "us with 0 / 1 / >=2 kernels in flight")
