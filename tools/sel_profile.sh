#!/bin/bash
# per-phase cost of k_bs_select on configs[2]: the kernel with every slice stopped after phase n (MXG_SEL_ABLATE; results are
# wrong on purpose), read from bench.py's per-kernel pass.  Run on the GPU box:  tools/sel_profile.sh > gpurun_out/sel_profile.txt
cd "$(dirname "$0")/.."
run() {
  python bench.py --no-cpu-baseline --no-end-to-end --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d.get('kernels',{}).get('ms_per_step',{})
print('$1', 'ms/step', d['ms_per_step'], {a.split(' ')[0]: b for a, b in k.items()}, 'first', (d['fallbacks'].get('first_step_of_the_handle') or {}).get('ms'))
"
}
for a in 0 1 2 3; do MXG_SEL_ABLATE=$a run "ablate=$a"; done
for b in 536870912 1073741824; do MXG_SEL_BATCH_KMERS=$b run "batch_kmers=$b"; done
for s in 256 512; do MXG_SPARSE_S=$s run "S=$s"; done
MXG_BS_SELECT=0 run "old route"
