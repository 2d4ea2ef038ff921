#!/bin/bash
# collect_profiles.sh SRC DST -- copy what tools/profile_bench.sh left under SRC (scratch, gpurun_out/...) into the tracked
# evidence directory DST (profiles/rNN) under stable names, then reduce the PMC passes and stamp profiles/hbm_traffic.json
# with the hash of the kernel sources (bench.py refuses a stale stamp).  Runs anywhere (no GPU).
set -eu
SRC=${1:-gpurun_out/prof_r04}
DST=${2:-profiles/r04}
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
# the pairs form on shards: its own kernel stats and counter passes (not part of hbm_traffic.json: that file is the one-GPU bench's)
if [ -d "$SRC/stats_ring" ]; then
    f=$(find "$SRC/stats_ring" -name "*_kernel_stats.csv" | head -1)
    cp "$f" "$DST/kernel_stats_ring.csv"
    cp "$SRC/ring_under_rocprof.log" "$DST/" 2>/dev/null || true
    mkdir -p "$DST/pmc_ring"
    for d in "$SRC"/pmc_ring/ring_p[0-9]*; do
        [ -d "$d" ] || continue
        f=$(find "$d" -name "*_counter_collection.csv" | head -1)
        cp "$f" "$DST/pmc_ring/$(basename "$d").counter_collection.csv"
    done
    python tools/pmc_summary.py "$DST/pmc_ring" > "$DST/pmc_ring_summary.txt"
fi
# the logs of the same GPU call that belong with the evidence (smoke, the GPU test suite), when the call left them beside the profiles
for f in smoke.log pytest_gpu.log; do
    [ -f "$SRC/$f" ] && cp "$SRC/$f" "$DST/$f"
done
python tools/pmc_summary.py "$DST/pmc" --json profiles/hbm_traffic.json --n 131072 --count 131072 --source "$DST/pmc/" > "$DST/pmc_summary.txt"
tail -n 12 "$DST/pmc_summary.txt"
# one rank's share at 2 / 4 / 8 ranks (STRICT, FAST ordered, the pairs form on shards): HBM counters only, added to the same file
for cdir in "$SRC"/pmc_shard/c*; do
    [ -d "$cdir" ] || continue
    C=$(basename "$cdir"); C=${C#c}
    mkdir -p "$DST/pmc_shard/c$C"
    for d in "$cdir"/*_p[0-9]*; do
        [ -d "$d" ] || continue
        f=$(find "$d" -name "*_counter_collection.csv" | head -1)
        [ -n "$f" ] && cp "$f" "$DST/pmc_shard/c$C/$(basename "$d").counter_collection.csv"
    done
    python tools/pmc_summary.py "$DST/pmc_shard/c$C" --json profiles/hbm_traffic.json --n 131072 --count "$C" --shard --source "$DST/pmc_shard/c$C/" > "$DST/pmc_shard/c${C}_summary.txt"
done
