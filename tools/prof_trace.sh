#!/bin/bash
# usage: tools/prof_trace.sh <tag> <rows> -- <program and args>: kernel trace + the timeline summary of its steady-state half
set -e
tag=$1; rows=$2; shift; shift; shift
out=$GRAFT_REPO_ROOT/gpurun_out/trace_$tag
mkdir -p $out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- "$@" > $out/stdout.log 2>&1
f=$(find $out -name "*kernel_trace.csv" | head -1)
python3 tools/timeline.py $f $rows > gpurun_out/${tag}_timeline.txt
find $out -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/${tag}_kernel_stats.csv
cat gpurun_out/${tag}_timeline.txt
