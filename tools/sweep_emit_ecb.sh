#!/bin/bash
# slices per k_emit tile behind k_bs_select (MXG_EMIT_ECB = 16 | 32 | 64): tools/sweep_emit_ecb.sh
cd "$(dirname "$0")/.."
for e in "" 16 32 ""; do
  for wl in configs2 repeats; do
    env ${e:+MXG_EMIT_ECB=$e} python bench.py --workload $wl --steps 20 --warmup 3 --no-cpu-baseline --no-end-to-end 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$e', '$wl', d['value'], d['ms_per_step'], ((d.get('kernels') or {}).get('ms_per_step') or {}).get('k_emit'))"
  done
done
