#!/bin/bash
# Runs on the GPU box: kernel trace of the VFE leg at the 8-GPU shard size, per-stream timeline of the last evaluation
set -e
tag=$1; shift
tools/profile_bench.sh $tag --workload vfe --vfe-n 524288 --vfe-steps 4 --no-cpu-baseline "$@"
out=gpurun_out/prof_$tag
tr=$(find $out -name "*kernel_trace.csv" | head -1)
python3 tools/trace_timeline.py $tr -1 scale_inputs_kernel --full > $out/timeline_full.txt
rm -f $tr
