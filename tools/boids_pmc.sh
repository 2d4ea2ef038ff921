#!/bin/bash
# boids_pmc.sh [COUNT] -- kernel stats + PMC passes over a few boids steps: the whole 131 072-body set, or with COUNT one
# rank's share of it (the form the library picks for that size).  Output under gpurun_out/boids_pmc[_COUNT]/.
set -u
COUNT=${1:-0}
OUT=gpurun_out/boids_pmc
[ "$COUNT" != 0 ] && OUT=${OUT}_$COUNT
ROOT=$(pwd)
export TMPDIR=/tmp
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/stats" -- python3 "$ROOT/tools/boids_run.py" 131072 10 $COUNT > $OUT/stats.log 2>&1
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES" "VALUBusy VALUUtilization" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$ROOT/$OUT/boids_p$i" -- python3 "$ROOT/tools/boids_run.py" 131072 3 $COUNT > $OUT/p$i.log 2>&1
  echo "pass $i rc=$?"
done
