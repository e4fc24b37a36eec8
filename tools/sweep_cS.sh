#!/bin/bash
# candidates per window x strip length on configs[2]: tools/sweep_cS.sh "c ..." "S ..."  (the slice kernel's own time beside the stretch kernels')
cd "$(dirname "$0")/.."
for c in ${1:-10 8 7 6}; do
for S in ${2:-320 416 480}; do
MXG_DEV_CAND=$c MXG_SPARSE_S=$S python bench.py --no-cpu-baseline --no-end-to-end --no-repeats --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d.get('kernels',{}).get('ms_per_step',{})
print('c=$c S=$S', d['ms_per_step'], {a.split(' ')[0]: b for a, b in k.items()}, d['config']['minimizers'], d.get('candidates_per_step'))"
done
done
