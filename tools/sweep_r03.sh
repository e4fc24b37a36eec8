#!/bin/bash
# one line of bench.py per setting (environment knobs): tools/sweep_r03.sh "A=1 B=2" "C=3" ...
for kv in "$@"; do
  env $kv timeout 150 python bench.py --steps 10 --warmup 3 --no-end-to-end --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$kv', d['value'], d['ms_per_step'], d['config']['minimizers'], d['kernels']['ms_per_step'])"
done
