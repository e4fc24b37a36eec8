#!/usr/bin/env python3
"""One step of bench.py out of a rocprofv3 kernel trace: tools/step_timeline.py b_kernel_trace.csv [out.txt]
(start offset, duration, queue and name of every kernel between two launches of k_hash_bs two apart, and how much of the span
had no kernel running / one / more than one)"""
import csv, sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mxg::", ""), r["Queue_Id"])
            for r in rows)
hs = [i for i, e in enumerate(ev) if e[2].startswith("k_hash_bs")]
i0, i1 = hs[len(hs) // 2 - 1 - (len(hs) // 2 - 1) % 2], hs[len(hs) // 2 + 1 - (len(hs) // 2 - 1) % 2]  # a whole step in the middle of the timed region: two filters and everything between them
sel = ev[i0:i1]
t0, t1 = sel[0][0], ev[i1][0]
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
print(f"# one step of `bench.py --steps 10 --warmup 3` under rocprofv3 --kernel-trace ({(t1 - t0) / 1e3:.0f} us from one reference filter to the next;", file=out)
print("# the profiler stretches the step: 3.9 ms without it).  Columns: start (us), duration (us), queue, kernel", file=out)
for s, e, n, q in sel:
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  q{q}  {n}", file=out)
pts = sorted([(s, 1) for s, e, *_ in sel] + [(min(e, t1), -1) for s, e, *_ in sel])
depth, last, acc = 0, t0, {0: 0, 1: 0, 2: 0}
for t, d in pts:
    acc[min(depth, 2)] += t - last
    last, depth = t, depth + d
acc[min(depth, 2)] += t1 - last
tot = float(t1 - t0)
print(f"# no kernel running {100 * acc[0] / tot:.1f} %, one {100 * acc[1] / tot:.1f} %, two or more {100 * acc[2] / tot:.1f} % of the span", file=out)
