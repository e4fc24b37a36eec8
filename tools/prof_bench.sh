#!/bin/bash
# kernel-trace profile of bench.py on the GPU box; CSVs are copied to gpurun_out/<name>/ (usage: tools/prof_bench.sh name [bench args])
name=${1:-prof}; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$name
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -o b -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline "$@" > /tmp/prof_$name.log 2>&1
echo "rocprofv3 rc=$?"
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/$name
cp /tmp/prof_$name/b_kernel_stats.csv /tmp/prof_$name/b_kernel_trace.csv $GRAFT_REPO_ROOT/gpurun_out/$name/ 2>/dev/null
grep '^{' /tmp/prof_$name.log > $GRAFT_REPO_ROOT/gpurun_out/$name/bench.json
