#!/bin/bash
# kernel-trace stats of an arbitrary command on the GPU box: tools/prof_cmd.sh <name> <command...>; CSV -> gpurun_out/<name>/
name=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$name
( cd $GRAFT_REPO_ROOT && timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -o b -- "$@" ) > /tmp/prof_$name.log 2>&1
echo "rocprofv3 rc=$?"; tail -3 /tmp/prof_$name.log
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/$name
cp /tmp/prof_$name/b_kernel_stats.csv $GRAFT_REPO_ROOT/gpurun_out/$name/ 2>/dev/null
