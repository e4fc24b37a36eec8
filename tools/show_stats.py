"""print a rocprofv3 kernel_stats.csv: python tools/show_stats.py gpurun_out/<name>/b_kernel_stats.csv [steps]"""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else None
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:28]:
    per = f" per_step_us={float(r['TotalDurationNs']) / 1e3 / steps:7.1f}" if steps else ""
    print(f"{r['Name'][:64]:64s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs']) / 1e3:8.1f} pct={float(r['Percentage']):5.1f}{per}")
if steps:
    print(f"sum of kernel time per step: {tot / 1e3 / steps:.1f} us")
