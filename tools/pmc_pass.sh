#!/bin/bash
# One rocprofv3 --pmc pass per counter group (no tracing combined), summarised per kernel.
# usage: tools/pmc_pass.sh <tag> <python script and args...>
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
i=0
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $group --output-format csv -d $out -o pass$i -- python3 "$@" > $out/pass$i.log 2>&1 || { echo "pass $i failed"; tail -3 $out/pass$i.log; exit 1; }
done <<GROUPS
SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU
FETCH_SIZE
WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
GROUPS
python3 tools/pmc_summary.py $out
