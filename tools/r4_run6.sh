#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
python -m pytest tests/test_gpu_lazy.py tests/test_gpu_compose.py -q -m gpu > gpurun_out/r4_tests_lazy.log 2>&1
tail -3 gpurun_out/r4_tests_lazy.log
python bench.py > gpurun_out/r4_bench_a.json 2> gpurun_out/r4_bench_a.err
tail -c 1500 gpurun_out/r4_bench_a.json; tail -3 gpurun_out/r4_bench_a.err
