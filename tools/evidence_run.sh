#!/bin/bash
# evidence_run.sh OUTDIR -- the un-profiled measurements of a round, on the GPU box from the repo root (tools/profile_bench.sh holds
# the rocprofv3 runs): the bench line, one rank's share in every form (shard_times, ring_times), the step in phases (ring_phases),
# what the exchanges' call path costs on a one-rank communicator (step_overhead), the bench rehearsals over gloo.
OUT=${1:-gpurun_out/r05}
mkdir -p "$OUT"
python3 bench.py > "$OUT/bench.log" 2>&1; echo "bench rc=$?"
python3 -u tools/shard_times.py > "$OUT/shard_times.log" 2>&1; echo "shard_times rc=$?"
python3 -u tools/ring_times.py 131072 > "$OUT/ring_times.log" 2>&1; echo "ring_times rc=$?"
python3 -u tools/ring_times.py 1048576 >> "$OUT/ring_times.log" 2>&1; echo "ring_times 2^20 rc=$?"
python3 -u tools/ring_phases.py 131072 8,4,2 0 0 > "$OUT/ring_phases.log" 2>&1; echo "ring_phases rc=$?"
python3 -u tools/step_overhead.py > "$OUT/step_overhead.log" 2>&1; echo "step_overhead rc=$?"
NB_BENCH_BACKEND=gloo python3 bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/bench_rehearsal_gpus2_gloo.log" 2>&1; echo "rehearsal 2 rc=$?"
NB_BENCH_BACKEND=gloo python3 bench.py --gpus 4 --mode fast --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/bench_rehearsal_gpus4_fast_gloo.log" 2>&1; echo "rehearsal 4 fast rc=$?"
NB_BENCH_BACKEND=gloo python3 bench.py --gpus 4 --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/bench_rehearsal_gpus4_gloo.log" 2>&1; echo "rehearsal 4 rc=$?"
