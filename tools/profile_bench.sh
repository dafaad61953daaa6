#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel trace + stats of bench.py; summaries land in gpurun_out/prof_<tag>/
# usage: tools/profile_bench.sh <tag> <bench args...>
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o run -- python3 bench.py "$@" > $out/bench.json 2> $out/stderr.log
find $out -name "*kernel_stats*" | head
