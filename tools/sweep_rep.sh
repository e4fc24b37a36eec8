#!/bin/bash
# the repeat-rich workload under environment knobs: tools/sweep_rep.sh "A=1" "B=2" ...
for kv in "$@"; do
  env $kv timeout 300 python bench.py --workload repeats --mbp 1000 --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$kv', d['value'], d['ms_per_step'], d['config']['minimizers'], d['fallbacks']['stretches_sketched_apart_per_step'], d['stage_ms_per_step'])"
done
