#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
{
python tools/pf_ab.py c2 3 "ahead xch2=" "ahead xch1=JD_XCH:1" 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-60
for leg in north c512 c3 hyps; do for x in 2 1; do echo "== $leg JD_XCH=$x"; JD_XCH=$x python tools/run_leg.py $leg 3 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d.get('value'), d.get('search_ms'), d.get('ms_per_pass'))"; done; done
} | tee gpurun_out/r4_xch2.log
