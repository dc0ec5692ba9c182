#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
OUT=gpurun_out/prof_broker
rm -rf "$OUT"; mkdir -p "$OUT"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python tools/broker_bench.py ${N:-64} < /dev/null > "$OUT/stats.log" 2>&1
grep callers "$OUT/stats.log" | cut -c1-200
f=$(find "$OUT/stats" -name '*kernel_stats.csv' | head -1)
cut -c1-160 "$f" | head -14
