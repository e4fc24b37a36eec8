cd $GRAFT_REPO_ROOT
for cfg in "" "MXG_GAP_BUDGET=1800 MXG_SEL_BATCH_KMERS=4000000000"; do
  env $cfg python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-end-to-end 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', d['value'], d['ms_per_step'], d['config']['minimizers'], d['config']['vertices'], d['config']['edges'], d['fallbacks'], (d.get('kernels') or {}).get('ms_per_step'))"
done
