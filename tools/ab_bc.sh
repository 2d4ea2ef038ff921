#!/bin/bash
# ab_bc.sh OUT NAME... -- one rank's STRICT share (tools/shard_times.py strict) with the product library and each variant, twice interleaved
OUT=$1; shift
mkdir -p "$(dirname "$OUT")"; : > "$OUT"
for round in 1 2; do
  for name in product "$@"; do
    if [ "$name" = product ]; then unset NENBODY_LIB; else export NENBODY_LIB="$PWD/build/variants/$name.so"; fi
    echo "== round $round: $name" >> "$OUT"
    python3 tools/shard_times.py strict 2>&1 | grep -v amdgpu.ids >> "$OUT"
  done
done
unset NENBODY_LIB
cat "$OUT"
