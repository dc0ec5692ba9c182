#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_broker.py -q -m gpu -x 2>&1 | grep -v amdgpu.ids | tail -30
JD_BROKER_RESIDENT=0 timeout 300 python -m pytest tests/test_gpu_broker.py -q -m gpu -x -k "threads_match or error_stays" 2>&1 | tail -3
timeout 600 python tools/broker_bench.py 4 16 32 48 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_broker_resident.log
