#!/bin/bash
# HBM-side traffic of the sketch-stage kernels (FETCH_SIZE, WRITE_SIZE passes) on configs[2]: tools/pmc_fetch.sh name
name=${1:-pmc_fetch}; shift
cd /tmp && export TMPDIR=/tmp
for grp in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pf_${name}_$grp
  MXG_ONE_STREAM=1 timeout -s KILL 400 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/pf_${name}_$grp -o p -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-end-to-end --no-kernels --steps 3 --warmup 1 "$@" > /tmp/pf_${name}_$grp.log 2>&1
  python3 - /tmp/pf_${name}_$grp/p_counter_collection.csv $grp <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0]
    acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
for k, (v, n) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
    if "k_synth" in k: continue
    mult = 2 if sys.argv[2] == "FETCH_SIZE" else 1
    print(sys.argv[2], k[:40], "launches", n, "MB per step", round(v * 1024 * mult / 4 / 1e6, 1))
PY
done
