cd $GRAFT_REPO_ROOT
run() { python bench.py --no-cpu-baseline --no-end-to-end --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d.get('kernels',{}).get('ms_per_step',{})
print('$1', d['ms_per_step'], {a.split(' ')[0]: b for a, b in k.items()}, d['config']['minimizers'], d['config']['edges'], d['fallbacks']['batches_redone_per_step'], d['fallbacks']['assemblies_enqueued_twice_per_step'], d.get('one_shot'))"; }
run default
MXG_SEL_BATCH_KMERS=4200000000 MXG_GAP_BUDGET=1700 run onebatch
