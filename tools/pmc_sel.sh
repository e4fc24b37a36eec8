#!/bin/bash
# SQ counters of the sketch-stage kernels on configs[2] (two rocprofv3 --pmc passes, kernels of the two streams kept apart):
#   tools/pmc_sel.sh name [bench args]   -> gpurun_out/<name>/pmc_by_kernel.json, the k_bs_select / k_emit / k_gap_fix rows printed
name=${1:-pmc_sel}; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/$name
mkdir -p $out
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE SQ_INSTS_SMEM" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_INSTS_VSKIPPED"; do
  i=$((i+1))
  rm -rf /tmp/pmc_${name}_$i
  MXG_ONE_STREAM=1 timeout -s KILL 400 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/pmc_${name}_$i -o p -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-end-to-end --no-kernels --steps 3 --warmup 1 "$@" > /tmp/pmc_${name}_$i.log 2>&1
  echo "group $i rc=$?"
  cp /tmp/pmc_${name}_$i/p_counter_collection.csv $out/counters_$i.csv 2>/dev/null
  cp /tmp/pmc_${name}_$i/p_kernel_trace.csv $out/trace_$i.csv 2>/dev/null
done
python3 - "$out" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in sorted(glob.glob(out + "/counters_*.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        a = acc[k][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
res = {k: {c: v[0] / v[1] for c, v in cs.items()} | {"launches": max(v[1] for v in cs.values())} for k, cs in acc.items()}
dur = collections.defaultdict(lambda: [0.0, 0])
for f in sorted(glob.glob(out + "/trace_*.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        dur[k][0] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"]); dur[k][1] += 1
for k in res:
    if k in dur: res[k]["avg_us"] = dur[k][0] / dur[k][1] / 1e3
json.dump(res, open(out + "/pmc_by_kernel.json", "w"), indent=1, sort_keys=True)
for k, c in res.items():
    if any(t in k for t in ("k_bs_select", "k_emit", "k_gap_fix", "k_bs_reorder", "k_resolve", "k_bs_count")):
        w = c.get("SQ_WAVES", 1)
        print(k, "launches", c["launches"], "avg_us", round(c.get("avg_us", 0), 1))
        for n in sorted(c):
            if n.startswith("SQ_") or n.startswith("GRBM"): print("   ", n, round(c[n]), "per wave", round(c[n] / w, 1))
PY
rm -f $out/counters_*.csv $out/trace_*.csv
