#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
{
for cw in 64 32 16 8 4 2 1; do echo "== one stream, cluster cap $cw"; JD_CW=$cw python tools/phase_trace.py --utts 1 2>&1 | grep -v amdgpu.ids; done
} | tee gpurun_out/r4_phase_single.log
