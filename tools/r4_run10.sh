#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
{
python tools/broker_bench.py 1 4 16 32 64
for tf in 96 ; do echo "tick $tf"; JD_BROKER_TICK_FRAMES=$tf python tools/broker_bench.py 16 32; done
} 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/r4_broker_bench2.log
