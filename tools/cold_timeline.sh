#!/bin/bash
# kernel timeline of the first step of a fresh handle in a warm process (tools/first_step.py under rocprofv3): the launches from the
# 7th filter launch (handle 1, step 0) to the 9th (its step 1), beside the same for a steady-state step (9th to 11th)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_cold
timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_cold -o c -- python $GRAFT_REPO_ROOT/tools/first_step.py > /tmp/prof_cold.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/prof_cold/**/c_kernel_trace.csv', recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mxg::", ""), r["Queue_Id"], int(r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", 0) or 0), int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1)) for r in csv.DictReader(open(f))))
hs = [i for i, e in enumerate(rows) if e[2].startswith("k_hash_bs")]
for name, a, b in (("first step of handle 1", hs[6], hs[8]), ("its second step", hs[8], hs[10])):
    t0 = rows[a][0]
    print(f"# {name}: {(rows[b][0] - t0) / 1e3:.0f} us from its first filter launch to the next step's")
    for s, e, n, q, g, wg in rows[a:b]:
        print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  q{q}  {n}  blocks {g // max(wg, 1)}")
PY
