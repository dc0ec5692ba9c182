#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
for n in 96 128 192; do JD_BROKER_RESIDENT=1 JD_BROKER_TICK_FRAMES=512 JD_BENCH_PUSH_FRAMES=2000 timeout 600 python tools/broker_bench.py $n 2>&1 | grep "callers\|rror" | cut -c1-330; done
