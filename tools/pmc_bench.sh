#!/bin/bash
# HBM traffic of the bench kernels: separate --pmc passes (FETCH_SIZE; WRITE_SIZE), no tracing.
# usage: tools/pmc_bench.sh <tag> <evaluations the command runs> <bench.py args...>
tag=$1; shift
evals=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out -o fetch -- python3 bench.py "$@" > $out/fetch.log 2>&1 || { tail -3 $out/fetch.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out -o write -- python3 bench.py "$@" > $out/write.log 2>&1 || { tail -3 $out/write.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES --output-format csv -d $out -o mfma -- python3 bench.py "$@" > $out/mfma.log 2>&1 || { tail -3 $out/mfma.log; exit 1; }
python3 tools/pmc_traffic.py $out $evals
