#!/bin/bash
# The filter's loads and stores with other cache-policy bits (gen/bs_gen.py reads BS_LOAD_MOD / BS_STORE_MOD): same stream, same
# results, timing at 3 Gbp.   tools/bs_memmod.sh   (GPU box)
cd "$(dirname "$0")/.."
mkdir -p /tmp/bs_mm
i=0
for v in "|" " nt|" "| nt" " nt| nt" " sc1|" "| sc1" " sc0 sc1| sc0 sc1" " sc0|" "| sc0" "|"; do
  i=$((i+1))
  lm="${v%%|*}"; sm="${v##*|}"
  BS_LOAD_MOD="$lm" BS_STORE_MOD="$sm" python ntjoin_amd/csrc/gen/bs_gen.py -o /tmp/bs_mm/hash_$i.inc 2>/dev/null >/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I ntjoin_amd/csrc -I /tmp/bs_mm -DHASH_BS_INC_FILE="\"hash_$i.inc\"" tools/bs_bench.hip -o /tmp/bs_mm/bench_$i 2>/dev/null || { echo "build of '$v' failed"; continue; }
  echo "== loads '$lm' stores '$sm'"
  /tmp/bs_mm/bench_$i 3000 | grep "verify\| 512 \|1024 " | cut -c1-100
done
