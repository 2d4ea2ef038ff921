#!/bin/bash
# profile_bench.sh OUTDIR -- the rocprofv3 evidence of one round, run on the GPU box from the repo root:
#   OUTDIR/stats   rocprofv3 --kernel-trace --stats of the bench command (kernel_stats: average duration per kernel)
#   OUTDIR/pmc/pN  one --pmc pass per counter set (never combined with a trace domain other than --kernel-trace),
#                  over a short bench run; tools/pmc_summary.py reduces them (and writes profiles/hbm_traffic.json)
# The program after `--` is python3 itself (no env / bash -c hop: the profiler has initialised the GPU by then).
set -u
OUT=${1:-gpurun_out/prof_r02}
ROOT=$(pwd)
mkdir -p "$OUT/stats" "$OUT/pmc"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/stats" -- python3 "$ROOT/bench.py" --steps 20 --warmup 3 --no-cpu-baseline > "$OUT/bench_under_rocprof.log" 2>&1
echo "stats rc=$?"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES" \
           "VALUBusy VALUUtilization" "TCC_HIT_sum TCC_MISS_sum" \
           "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS"; do
    i=$((i + 1))
    rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$ROOT/$OUT/pmc/p$i" -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/pmc/p$i.log" 2>&1
    echo "pmc pass $i ($set) rc=$?"
done
find "$OUT" -name "*.csv" | head -40
