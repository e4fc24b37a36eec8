#!/bin/bash
# Where the filter's time goes beyond its issue bound (k_hash_bs: 7840 full-rate VALU instructions per chunk, 2 cycles each at best):
# the same generated instruction stream with its vector loads removed, its stores removed, both removed -- timing variants of
# gen/bs_gen.py (--ablate; results wrong on purpose), each built into tools/bs_bench.hip and run at 3 Gbp with 1, 2, 3 and 4 blocks per CU.
#   tools/bs_ablate.sh [Mbp]   (GPU box)   -> stdout; profiles/ubench/bs_ablate_r05.txt is a committed run
cd "$(dirname "$0")/.."
mbp=${1:-3000}
mkdir -p tools/bin /tmp/bs_abl
for v in ${VARIANTS:-full loads stores loads,stores}; do
  name=$(echo $v | tr ',' '_')
  abl=$v; [ "$v" = full ] && abl=""
  python ntjoin_amd/csrc/gen/bs_gen.py -o /tmp/bs_abl/hash_$name.inc --ablate "$abl" 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I ntjoin_amd/csrc -I /tmp/bs_abl -DHASH_BS_INC_FILE="\"hash_$name.inc\"" tools/bs_bench.hip -o /tmp/bs_abl/bench_$name 2>/dev/null || { echo "build of $name failed"; continue; }
  echo "== variant: $( [ "$v" = full ] && echo 'the shipped kernel' || echo "without its $v" )"
  /tmp/bs_abl/bench_$name $mbp 164 $PROBE | grep -v MISMATCH
done
