#!/bin/bash
# one PMC pass over bench.py: tools/pmc_one.sh <tag> "<counters>" [env assignments...]   (per-kernel averages printed)
tag=$1; ctrs=$2; shift 2
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc1_$tag
env "$@" MXG_ONE_STREAM=1 timeout -s KILL 400 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d /tmp/pmc1_$tag -o p -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-end-to-end --no-kernels --steps 2 --warmup 1 > /tmp/pmc1_$tag.log 2>&1
python3 - /tmp/pmc1_$tag/p_counter_collection.csv "$tag" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for r in csv.DictReader(open(sys.argv[1])):
    a = acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]]
    a[0] += float(r["Counter_Value"]); a[1] += 1
for k, cs in acc.items():
    if "k_hash_sparse" in k or "k_reorder" in k or "k_resolve" in k:
        print(sys.argv[2], k[:40], {c: round(v[0] / v[1], 1) for c, v in cs.items()}, flush=True)
PY
