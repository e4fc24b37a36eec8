#!/bin/bash
# A/B of two builds of the library on one GPU box, interleaved: ab/base/libntjoin_mx.so against ntjoin_amd/lib/.  usage: tools/ab.sh [rounds] [bench args]
rounds=${1:-2}; shift
for i in $(seq $rounds); do
  for which in base new; do
    if [ $which = base ]; then export MXG_LIB_DIR=$PWD/ab/base; else unset MXG_LIB_DIR; fi
    timeout 300 python bench.py --steps 20 --warmup 5 --no-end-to-end --no-cpu-baseline --no-repeats "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$which', d['value'], d['ms_per_step'], d['kernels']['ms_per_step'])"
  done
done
