#!/bin/bash
# usage: tools/pmc.sh <tag> "<counters...>" -- <program and args>
# One rocprofv3 --pmc pass (counters only; no tracing domains combined with it).
set -e
tag=$1; ctr=$2; shift; shift; shift
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --pmc $ctr --output-format csv -d $out -- "$@" > $out/stdout.log 2>&1 || { tail -20 $out/stdout.log; exit 1; }
f=$(find $out -name "*counter_collection.csv" | head -1)
cp $f $GRAFT_REPO_ROOT/gpurun_out/${tag}_counters.csv
