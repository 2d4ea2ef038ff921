#!/bin/bash
# profile_bench.sh OUTDIR -- the rocprofv3 evidence of one round, run on the GPU box from the repo root:
#   OUTDIR/stats_{strict,fast}  rocprofv3 --kernel-trace --stats of `bench.py --mode M --no-secondary --steps 100`: ONE arithmetic per run, so
#                               a kernel's average duration is that of the bench's own timed launches (the 3-D leg of the full
#                               bench runs the same kernel template on slower data and would skew it)
#   OUTDIR/stats_all            the same of the full default bench command (boids, 3-D legs, the issue-ceiling streams)
#   OUTDIR/pmc/M_pN             one --pmc pass per counter set and arithmetic (never combined with a trace domain other than
#                               --kernel-trace), over a short bench run; tools/pmc_summary.py reduces them (and writes
#                               profiles/hbm_traffic.json)
# The program after `--` is python3 itself (no env / bash -c hop: the profiler has initialised the GPU by then).
set -u
OUT=${1:-gpurun_out/prof_r05}
ROOT=$(pwd)
mkdir -p "$OUT/pmc"
export TMPDIR=/tmp
for mode in strict fast; do
    mkdir -p "$OUT/stats_$mode"
    rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/stats_$mode" -- python3 "$ROOT/bench.py" --mode $mode --no-secondary --steps 100 --warmup 3 --no-cpu-baseline > "$OUT/bench_${mode}_under_rocprof.log" 2>&1
    echo "stats $mode rc=$?"
done
mkdir -p "$OUT/stats_all"
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/stats_all" -- python3 "$ROOT/bench.py" --steps 20 --warmup 3 --no-cpu-baseline > "$OUT/bench_all_under_rocprof.log" 2>&1
echo "stats all rc=$?"
for mode in strict fast; do
    i=0
    for set in "FETCH_SIZE" "WRITE_SIZE" \
               "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES" \
               "VALUBusy VALUUtilization" "TCC_HIT_sum TCC_MISS_sum" \
               "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS"; do
        i=$((i + 1))
        rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$ROOT/$OUT/pmc/${mode}_p$i" -- python3 "$ROOT/bench.py" --mode $mode --no-secondary --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/pmc/${mode}_p$i.log" 2>&1
        echo "pmc $mode pass $i ($set) rc=$?"
    done
done
# the pairs form on shards (tools/ring_times.py: rank 0 and rank 7 of eight at N = 131 072): kernel trace + the two HBM counters
mkdir -p "$OUT/stats_ring"
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/stats_ring" -- python3 "$ROOT/tools/ring_times.py" 131072 8 > "$OUT/ring_under_rocprof.log" 2>&1
echo "stats ring rc=$?"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES" "VALUBusy VALUUtilization"; do
    i=$((i + 1))
    rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$ROOT/$OUT/pmc_ring/ring_p$i" -- python3 "$ROOT/tools/ring_times.py" 131072 8 > "$OUT/pmc_ring_p$i.log" 2>&1
    echo "pmc ring pass $i ($set) rc=$?"
done
# one rank's share at 2 / 4 / 8 ranks: the two HBM counters of every form a rank may run -- STRICT, FAST as ordered pairs, FAST in
# the pairs form on shards (tools/shard_run.py, tools/ring_times.py) -- so that a multi-GPU bench line carries `roofline.traffic` too
for P in 2 4 8; do
    C=$((131072 / P))
    i=0
    for set in "FETCH_SIZE" "WRITE_SIZE"; do
        i=$((i + 1))
        rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$ROOT/$OUT/pmc_shard/c$C/strict_p$i" -- python3 "$ROOT/tools/shard_run.py" $C 5 > "$OUT/pmc_shard_c${C}_strict_p$i.log" 2>&1
        echo "pmc shard $C strict pass $i rc=$?"
        export NB_MODE=fast
        rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$ROOT/$OUT/pmc_shard/c$C/fast_p$i" -- python3 "$ROOT/tools/shard_run.py" $C 5 > "$OUT/pmc_shard_c${C}_fast_p$i.log" 2>&1
        echo "pmc shard $C fast pass $i rc=$?"
        unset NB_MODE
        rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$ROOT/$OUT/pmc_shard/c$C/ring_p$i" -- python3 "$ROOT/tools/ring_times.py" 131072 $P > "$OUT/pmc_shard_c${C}_ring_p$i.log" 2>&1
        echo "pmc shard $C ring pass $i rc=$?"
    done
done
# the smoke of the same sources on the same box, beside the profiles (collect_profiles.sh copies it: no stale smoke.log)
python3 -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1
echo "smoke rc=$?"
find "$OUT" -name "*.csv" | wc -l
