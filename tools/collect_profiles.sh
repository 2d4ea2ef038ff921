#!/bin/bash
# collect_profiles.sh SRC DST -- reduce what tools/profile_bench.sh left under SRC (scratch, gpurun_out/...) into the tracked
# evidence directory DST (profiles/rNN): the rocprofv3 --stats tables and the bench lines under stable names, the --pmc passes as
# SUMMARIES (pmc_summary.txt, pmc_ring_summary.txt, pmc_shard/cN_summary.txt), and profiles/hbm_traffic.json stamped with the hash
# of the benchmarked kernels' device code (bench.py refuses a stale stamp).  The raw per-dispatch *_counter_collection.csv files
# stay in SRC: they were 13 MB of a 22 MB tree in round 4, and the summaries are the evidence.
# Runs anywhere (no GPU).
set -eu
SRC=${1:-gpurun_out/prof_r05}
DST=${2:-profiles/r05}
mkdir -p "$DST"
for mode in strict fast all; do
    f=$(find "$SRC/stats_$mode" -name "*_kernel_stats.csv" | head -1)
    cp "$f" "$DST/kernel_stats_$mode.csv"
    [ -f "$SRC/bench_${mode}_under_rocprof.log" ] && cp "$SRC/bench_${mode}_under_rocprof.log" "$DST/"
done
# the pairs form on shards: its own kernel stats and counter passes (not part of hbm_traffic.json: that file is the one-GPU bench's)
if [ -d "$SRC/stats_ring" ]; then
    f=$(find "$SRC/stats_ring" -name "*_kernel_stats.csv" | head -1)
    cp "$f" "$DST/kernel_stats_ring.csv"
    cp "$SRC/ring_under_rocprof.log" "$DST/" 2>/dev/null || true
    python tools/pmc_summary.py "$SRC/pmc_ring" > "$DST/pmc_ring_summary.txt"
fi
# the logs of the same GPU call that belong with the evidence (smoke, the GPU test suite), when the call left them beside the profiles
for f in smoke.log pytest_gpu.log; do
    [ -f "$SRC/$f" ] && cp "$SRC/$f" "$DST/$f"
done
python tools/pmc_summary.py "$SRC/pmc" --json profiles/hbm_traffic.json --n 131072 --count 131072 --source "$DST/pmc_summary.txt" > "$DST/pmc_summary.txt"
tail -n 12 "$DST/pmc_summary.txt"
# one rank's share at 2 / 4 / 8 ranks (STRICT, FAST ordered, the pairs form on shards): HBM counters only, added to the same file
for cdir in "$SRC"/pmc_shard/c*; do
    [ -d "$cdir" ] || continue
    C=$(basename "$cdir"); C=${C#c}
    mkdir -p "$DST/pmc_shard"
    python tools/pmc_summary.py "$cdir" --json profiles/hbm_traffic.json --n 131072 --count "$C" --shard --source "$DST/pmc_shard/c${C}_summary.txt" > "$DST/pmc_shard/c${C}_summary.txt"
done
