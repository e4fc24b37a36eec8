#!/bin/bash
# strip-length sweep of the sparse hash kernel: tools/sweep_S.sh <mbp> <steps> S1 S2 ...
mbp=$1; steps=$2; shift 2
for S in "$@"; do
  MXG_SPARSE_S=$S timeout 500 python bench.py --steps $steps --warmup 2 --no-cpu-baseline --mbp $mbp 2>&1 | tail -1 | S=$S python -c "
import json,sys,os
d=json.loads(sys.stdin.read()); print('mbp=$mbp S=%s'%os.environ['S'], round(d['value'],1), 'Gbp/s', d['ms_per_step'], 'ms/step  hash launch', d['roofline']['avg_launch_ms'], d['stage_ms_per_step'])"
done
