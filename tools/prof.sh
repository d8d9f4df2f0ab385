#!/bin/bash
# usage: tools/prof.sh <tag> -- <program and args>   (run on the GPU box from the repo root)
# kernel-trace + stats only (PMC counters are collected in separate runs, see tools/pmc.sh)
set -e
tag=$1; shift; shift
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- "$@" > $out/stdout.log 2>&1
find $out -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $GRAFT_REPO_ROOT/gpurun_out/${tag}_kernel_stats.csv
