#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 900 python bench.py --no-cpu-baseline --no-extra-legs --steps 20 --warmup 5 2> gpurun_out/r4_bench_pipe.err | tee gpurun_out/r4_bench_pipe.json | cut -c1-1500
tail -3 gpurun_out/r4_bench_pipe.err
