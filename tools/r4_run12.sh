#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
python -m pytest tests -q -m gpu -x -k "gmm or streaming or stream or broker or partial" 2>&1 | tail -4
python tools/stream_latency.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_stream_latency.log
python tools/broker_bench.py 1 16 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_broker_bench3.log
