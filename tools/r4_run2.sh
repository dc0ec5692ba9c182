#!/bin/bash
# development: RCCL banner probe + broker tests
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
python tools/rccl_banner_probe.py > gpurun_out/r4_rccl_probe.log 2>&1
cat gpurun_out/r4_rccl_probe.log
python -m pytest tests/test_gpu_broker.py -x -q -m gpu -s > gpurun_out/r4_tests_broker.log 2>&1
tail -25 gpurun_out/r4_tests_broker.log
