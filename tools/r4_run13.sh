#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
{
python tools/pf_ab.py c2 3 "two xch2=" "two xch3=JD_XCH:3" "two xch4=JD_XCH:4" "two xch8=JD_XCH:8" "two xch1=JD_XCH:1" 2>&1 | grep -v amdgpu.ids | tail -12
for x in 2 4 8; do echo "== one stream, 8 workgroups, JD_XCH=$x"; JD_XCH=$x JD_CW=8 python tools/phase_trace.py --utts 1 2>&1 | grep -E "phase X|wg wait X|barriers X|^sum"; done
} | tee gpurun_out/r4_xch.log
