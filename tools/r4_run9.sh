#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
python -m pytest tests/test_gpu_cells.py tests/test_gpu_parity.py tests/test_gpu_broker.py -q -m gpu -k "cells or streaming or two_batches or broker_threads or partial" 2>&1 | tail -3
python tools/broker_bench.py 1 4 16 32 64 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_broker_bench.log
JD_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --steps 4 --warmup 3 --no-cpu-baseline --no-extra-legs 2>gpurun_out/r4_bench_2ranks.err | tail -c 700
echo
JD_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --total-utts 128 --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs 2>>gpurun_out/r4_bench_2ranks.err | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['scaling'], d['config'].get('predicted_rank_ms'))"
tail -3 gpurun_out/r4_bench_2ranks.err
