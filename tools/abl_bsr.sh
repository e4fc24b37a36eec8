#!/bin/bash
# k_bs_resolve stopped after phase n (MXG_BSR_ABLATE): rocprofv3 average per launch
cd /tmp && export TMPDIR=/tmp
for a in "$@"; do
  rm -rf /tmp/abl$a
  MXG_BSR_ABLATE=$a MXG_ONE_STREAM=1 timeout 60 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abl$a -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-end-to-end --no-cpu-baseline --no-kernels > /tmp/abl$a.log 2>&1
  f=$(find /tmp/abl$a -name "*kernel_stats.csv" | head -1)
  echo "ablate=$a $(grep k_bs_resolve $f | cut -d, -f2-4)"
done
