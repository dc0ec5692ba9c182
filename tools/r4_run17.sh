#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
{
JD_BROKER_RESIDENT=1 timeout 600 python tools/broker_bench.py 40 48 56 64
} 2>&1 | grep -v "amdgpu.ids\|^one batch" | tee gpurun_out/r4_broker_resident3.log
