#!/bin/bash
# k_sel_stretch by parts (MXG_SST_ABLATE: 1 no rolls, 2 no first hash, 4 no window scans, 8 nothing per request, 32 no row update): its average launch under rocprofv3
cd "$(dirname "$0")/.."
for abl in ${@:-0 1 3 4 7 8 32}; do
MXG_SST_ABLATE=$abl MXG_DEV_CAND=${CAND:-8} tools/prof_bench.sh sst$abl --no-end-to-end --no-repeats --steps 6 --warmup 2 > /dev/null 2>&1
echo "ablate=$abl $(grep k_sel_stretch gpurun_out/sst$abl/b_kernel_stats.csv | cut -d, -f2-4)"
done
