#!/bin/bash
# The filter's loads and stores as buffer instructions (gen/bs_gen.py reads BS_MUBUF=1) against the shipped global ones: same stream,
# same results, three alternating runs at 3 Gbp.   tools/bs_mubuf.sh   (GPU box)
cd "$(dirname "$0")/.."
mkdir -p /tmp/bs_mb
for v in 0 1; do
  BS_MUBUF=$v python ntjoin_amd/csrc/gen/bs_gen.py -o /tmp/bs_mb/hash_$v.inc >/dev/null 2>&1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I ntjoin_amd/csrc -I /tmp/bs_mb -DHASH_BS_INC_FILE="\"hash_$v.inc\"" tools/bs_bench.hip -o /tmp/bs_mb/bench_$v 2>/dev/null || echo "build failed"
done
for rep in 1 2 3; do for v in 0 1; do
  echo "== buffer instructions: $v (run $rep)"
  /tmp/bs_mb/bench_$v 3000 | grep "verify\| 256 \| 512 \|1024 " | cut -c1-100
done; done
