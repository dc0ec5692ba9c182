#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
for rep in 1 2; do
for dp in 0 6; do
echo "== pipeline depth $dp"
timeout 600 python bench.py --no-cpu-baseline --no-extra-legs --steps 20 --warmup 5 --pipeline-depth $dp 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['ms_each_step'])"
done; done 2>&1 | tee gpurun_out/r4_pipe_vs_two.log
