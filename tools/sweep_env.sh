#!/bin/bash
# one environment knob over several values on configs[2]: tools/sweep_env.sh NAME "v1 v2 ..." [bench args]   (value "-" = unset)
cd "$(dirname "$0")/.."
name=$1; vals=$2; shift 2
for v in $vals; do
if [ "$v" = "-" ]; then unset $name; else export $name=$v; fi
python bench.py --no-cpu-baseline --no-end-to-end --no-repeats --steps 10 --warmup 3 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d.get('kernels',{}).get('ms_per_step',{})
print('$name=$v', d['ms_per_step'], {a.split(' ')[0]: b for a, b in k.items()}, d['config']['minimizers'])"
done
