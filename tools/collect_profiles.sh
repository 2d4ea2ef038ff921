#!/bin/bash
# collect_profiles.sh SRC DST -- copy what tools/profile_bench.sh left under SRC (scratch, gpurun_out/...) into the tracked
# evidence directory DST (profiles/rNN) under stable names, then reduce the PMC passes and stamp profiles/hbm_traffic.json
# with the hash of the kernel sources (bench.py refuses a stale stamp).  Runs anywhere (no GPU).
set -eu
SRC=${1:-gpurun_out/prof_r03}
DST=${2:-profiles/r03}
mkdir -p "$DST/pmc"
for mode in strict fast all; do
    f=$(find "$SRC/stats_$mode" -name "*_kernel_stats.csv" | head -1)
    cp "$f" "$DST/kernel_stats_$mode.csv"
    [ -f "$SRC/bench_${mode}_under_rocprof.log" ] && cp "$SRC/bench_${mode}_under_rocprof.log" "$DST/"
done
for d in "$SRC"/pmc/*_p[0-9]*; do
    [ -d "$d" ] || continue
    f=$(find "$d" -name "*_counter_collection.csv" | head -1)
    cp "$f" "$DST/pmc/$(basename "$d").counter_collection.csv"
done
python tools/pmc_summary.py "$DST/pmc" --json profiles/hbm_traffic.json --n 131072 --count 131072 --source "$DST/pmc/" > "$DST/pmc_summary.txt"
tail -n 12 "$DST/pmc_summary.txt"
