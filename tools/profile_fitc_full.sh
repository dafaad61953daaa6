#!/bin/bash
# Runs on the GPU box: kernel trace of the FITC leg at full size, per-stream timeline of the last evaluation
set -e
tag=$1; shift
tools/profile_bench.sh $tag --workload fitc --sparse-steps 3 --no-cpu-baseline "$@"
out=gpurun_out/prof_$tag
tr=$(find $out -name "*kernel_trace.csv" | head -1)
python3 tools/trace_timeline.py $tr -1 scale_inputs_kernel --full > $out/timeline_full.txt
python3 tools/chain_from_trace.py $tr > $out/chains.txt
rm -f $tr
