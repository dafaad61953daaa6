#!/bin/bash
# FETCH_SIZE (L2-miss bytes) per launch of the GEMM shape probe.  usage: tools/probe/pmc_shape.sh <tag> <probe args...>
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out -o fetch -- tools/probe/gemm_shape_probe "$@" > $out/fetch.log 2>&1 || { tail -3 $out/fetch.log; exit 1; }
python3 - $out <<'PY'
import csv, glob, sys, os
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        if "gemm" in r["Kernel_Name"]:
            print(r["Kernel_Name"][:50], "grid", r.get("Grid_Size"), "fetch GB (x2 corrected)", 2 * float(r["Counter_Value"]) * 1024 / 1e9)
PY
