#!/bin/bash
# Runs on the GPU box: kernel trace of the FITC leg at the 8-GPU shard size, full per-stream timeline of the last evaluation
# usage: tools/profile_shard.sh <tag> [extra bench args]
set -e
tag=$1; shift
tools/profile_bench.sh $tag --workload fitc --fitc-n 131072 --sparse-steps 6 --no-cpu-baseline "$@"
out=gpurun_out/prof_$tag
tr=$(find $out -name "*kernel_trace.csv" | head -1)
python3 tools/trace_timeline.py $tr -1 scale_inputs_kernel --full > $out/timeline_full.txt
python3 tools/chain_from_trace.py $tr > $out/chains.txt
rm -f $tr
tail -3 $out/timeline_full.txt | head -1
