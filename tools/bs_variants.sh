#!/bin/bash
# Generator variants of the filter side by side: tools/bs_variants.sh "<gen args>" "<gen args>" ...   (GPU box; "" = the shipped stream)
# Each is generated, built into tools/bs_bench.hip, verified against the direct formula and timed at 3 Gbp, twice, interleaved.
cd "$(dirname "$0")/.."
mkdir -p /tmp/bs_var
i=0
for v in "$@"; do
  i=$((i+1))
  python ntjoin_amd/csrc/gen/bs_gen.py -o /tmp/bs_var/hash_$i.inc $v 2>&1 | sed "s|^|[$v] |"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I ntjoin_amd/csrc -I /tmp/bs_var -DHASH_BS_INC_FILE="\"hash_$i.inc\"" tools/bs_bench.hip -o /tmp/bs_var/bench_$i 2>/dev/null || echo "build of '$v' failed"
done
for rep in 1 2 3; do
  i=0
  for v in "$@"; do
    i=$((i+1))
    echo "== '$v' (run $rep)"
    /tmp/bs_var/bench_$i ${MBP:-3000} | grep "verify\| 256 \| 512 \|1024 " | cut -c1-100
  done
done
