#!/bin/bash
# Does the filter's time follow its CODE SIZE?  The same instruction stream with every two-source instruction in its 8-byte
# (VOP3) encoding instead of the 4-byte one (gen/bs_gen.py reads BS_E64): same instructions, same results, a larger loop.
#   tools/bs_codesize.sh   (GPU box)
cd "$(dirname "$0")/.."
mkdir -p /tmp/bs_cs
for v in 0 1 0 1; do
  BS_E64=$v python ntjoin_amd/csrc/gen/bs_gen.py -o /tmp/bs_cs/hash_$v.inc >/dev/null 2>&1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I ntjoin_amd/csrc -I /tmp/bs_cs -DHASH_BS_INC_FILE="\"hash_$v.inc\"" tools/bs_bench.hip -o /tmp/bs_cs/bench_$v 2>/dev/null || { echo "build failed"; continue; }
  /opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=/tmp/bs_cs/bench_$v --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=/tmp/bs_cs/dev_$v.co 2>/dev/null
  echo "== 8-byte encodings: $v"
  /tmp/bs_cs/bench_$v 3000 | grep "verify\| 256 \| 512 \|1024 " | cut -c1-100
done
