#!/bin/bash
# A/B of environment settings on one GPU box, interleaved.  usage: tools/abenv.sh rounds "ENV=a" "ENV=b" ... [-- bench args]
rounds=$1; shift
envs=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do envs+=("$1"); shift; done
[ "$1" = "--" ] && shift
for i in $(seq $rounds); do
  for v in "${envs[@]}"; do
    env $v timeout 300 python bench.py --steps 20 --warmup 5 --no-end-to-end --no-cpu-baseline --no-repeats "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']['ms_per_step']; print('$v', d['value'], d['ms_per_step'], {x[:12]: k[x] for x in k})"
  done
done
