#!/bin/bash
# strip length sweep of the k_bs_select route on configs[2]: tools/sweep_sel_S.sh [S ...]
cd "$(dirname "$0")/.."
for S in ${@:-320 352 384 416 448}; do
MXG_SPARSE_S=$S python bench.py --no-cpu-baseline --no-end-to-end --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d.get('kernels',{}).get('ms_per_step',{})
print('S=$S', d['ms_per_step'], {a.split(' ')[0]: b for a, b in k.items()}, d['config']['minimizers'])"
done
