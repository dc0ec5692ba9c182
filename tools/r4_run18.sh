#!/bin/bash
# the drop-in seam's throughput: N serial callers through the broker, both workers (profiles/r04_broker_bench.log)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
{
echo "== resident search kernel (default)"
timeout 900 python tools/broker_bench.py 1 4 16 32 48 64
echo "== ticks (JD_BROKER_RESIDENT=0)"
JD_BROKER_RESIDENT=0 timeout 900 python tools/broker_bench.py 1 4 16 32 48 64
} 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/r04_broker_bench.log
