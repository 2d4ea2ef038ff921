#!/bin/bash
# ab_pairs.sh OUT NAME... -- the pairs form's step time (tools/pairs_time.py: N = 131 072, one GPU) and an eight-rank ring share
# (tools/ring_times.py) with the product library and with each build/variants/NAME.so (tools/build_variant.sh), interleaved twice so
# that a clock drift of the device shows up as a spread instead of as a difference.  Run on the GPU box from the repo root.
OUT=$1; shift
mkdir -p "$(dirname "$OUT")"
: > "$OUT"
for round in 1 2; do
    for name in product "$@"; do
        if [ "$name" = product ]; then unset NENBODY_LIB; else export NENBODY_LIB="$PWD/build/variants/$name.so"; fi
        echo "== round $round: $name" >> "$OUT"
        python3 tools/pairs_time.py 131072 40 >> "$OUT" 2>&1
        python3 tools/ring_times.py 131072 8 2>&1 | grep "8 ranks" >> "$OUT"
    done
done
unset NENBODY_LIB
cat "$OUT"
