#!/bin/bash
# blocks of k_gap_fix (MXG_GAP_FIX_BLOCKS) on configs[2] and on the repeat-rich workload: tools/sweep_gap_blocks.sh
cd "$(dirname "$0")/.."
for b in 256 512 768 1024 2048; do
  for wl in configs2 repeats; do
    MXG_GAP_FIX_BLOCKS=$b python bench.py --workload $wl --steps 20 --warmup 3 --no-cpu-baseline --no-end-to-end 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$b', '$wl', d['value'], d['ms_per_step'], (d.get('kernels') or {}).get('ms_per_step'))"
  done
done
