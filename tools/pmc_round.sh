#!/bin/bash
# PMC passes over bench.py at a workload (default configs[2]: 3 Gbp + 3 Gbp) on the GPU box: one rocprofv3 run per counter
# group (--pmc with --kernel-trace only), kernels of the two streams kept apart (MXG_ONE_STREAM=1) so that a kernel's counters are its
# own.  Per-kernel averages per launch (+ the launches' average duration) -> gpurun_out/<name>/pmc_by_kernel.json.
#   usage: tools/pmc_round.sh name [bench args]
name=${1:-pmc}; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/$name
mkdir -p $out
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE SQ_INSTS_SMEM" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rm -rf /tmp/pmc_${name}_$i
  MXG_ONE_STREAM=1 timeout -s KILL 400 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/pmc_${name}_$i -o p -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-end-to-end --no-kernels --no-repeats --steps 3 --warmup 1 "$@" > /tmp/pmc_${name}_$i.log 2>&1
  echo "group $i rc=$?"
  grep '^{' /tmp/pmc_${name}_$i.log | tail -1 > $out/bench_pass_$i.json
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
PY
rm -f $out/counters_*.csv $out/trace_*.csv
