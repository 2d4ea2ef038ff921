#!/bin/bash
# PMC passes over the pairs form (each its own run; --kernel-trace only)
OUT=gpurun_out/pairs_pmc; mkdir -p $OUT; export TMPDIR=/tmp; ROOT=$(pwd); i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES" "VALUBusy VALUUtilization" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $ROOT/$OUT/p$i -- python3 $ROOT/tools/pairs_prof.py 6 > $OUT/p$i.log 2>&1
  echo "pass $i rc=$?"
done
