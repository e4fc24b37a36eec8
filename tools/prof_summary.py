"""print the top rows of a rocprofv3 --kernel-trace --stats kernel_stats.csv: python tools/prof_summary.py <csv> [n]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
for r in rows[:n]:
    print(f"{r['Name'][:64]:64s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs']) / 1e3:9.1f} "
          f"tot_ms={float(r['TotalDurationNs']) / 1e6:8.2f} pct={r['Percentage']}")
