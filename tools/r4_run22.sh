#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
{
for ch in 64 128 256 100000; do echo "== chunk $ch"; JD_PIPE_CHUNK=$ch PIPE_AB_STEPS=24 timeout 600 python tools/pipe_ab.py 160 | grep resident; done
for ch in 128 256; do echo "== bench, chunk $ch"; JD_PIPE_CHUNK=$ch timeout 600 python bench.py --no-cpu-baseline --no-extra-legs --steps 20 --warmup 5 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['ms_each_step'])"; done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_pipe_ab2.log
