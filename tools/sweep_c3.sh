#!/bin/bash
# configs[3] (w = 500, four assemblies): strip length x candidates per window of the k_bs_select route.  tools/sweep_c3.sh "S..." "c..."
cd "$(dirname "$0")/.."
for S in ${1:-160 192 224 256 320}; do for c in ${2:-0}; do
MXG_SPARSE_S=$S python bench.py --workload configs3 --cand $c --no-cpu-baseline --no-end-to-end --steps 8 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d.get('kernels',{}).get('ms_per_step',{})
print('S=$S c=$c', d['value'], d['ms_per_step'], {a.split(' ')[0]: b for a, b in k.items()}, d['config']['minimizers'], d['fallbacks']['candidates_per_step'])"
done; done
