#!/bin/bash
# Runs on the GPU box: kernel trace of the exact n = 8192 leg, per-stream timeline of the last evaluation
set -e
tag=$1; shift
tools/profile_bench.sh $tag --workload exact8192 --no-cpu-baseline "$@"
out=gpurun_out/prof_$tag
tr=$(find $out -name "*kernel_trace.csv" | head -1)
python3 tools/trace_timeline.py $tr -1 scale_inputs_kernel --full > $out/timeline_full.txt
python3 tools/eval_breakdown.py $tr "trace_sym_mfma_kernel<3" 64 > $out/breakdown.txt
rm -f $tr
