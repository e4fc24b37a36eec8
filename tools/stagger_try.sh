#!/bin/bash
# streams of single-batch assemblies: staggered (default), free (MXG_STAGGER=0), one stream -- step time and the filter's own time
cd "$(dirname "$0")/.."
for cfg in "MXG_STAGGER=1" "MXG_STAGGER=0" "MXG_ONE_STREAM=1" "MXG_STAGGER=1" "MXG_STAGGER=0"; do
  for wl in configs2 repeats; do
  env $cfg python bench.py --workload $wl --steps 20 --warmup 3 --no-cpu-baseline --no-end-to-end --no-kernels 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', '$wl', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['step_ms_min_max'])"
  done
done
