#!/bin/bash
# development: the GPU suite + the phase accounting and the step time of configs[1]   usage: tools/r4_run.sh <tag> [pytest args]
cd "$(dirname "$0")/.." || exit 1
tag=${1:-x}; shift
mkdir -p gpurun_out
python -m pytest tests -q -m gpu "$@" > gpurun_out/r4_tests_$tag.log 2>&1
tail -4 gpurun_out/r4_tests_$tag.log
python tools/phase_trace.py > gpurun_out/r4_phase_$tag.log 2>&1
tail -12 gpurun_out/r4_phase_$tag.log | head -11
python tools/pf_ab.py c2 2 ahead= serial=JD_NOTHING:0 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['variant'], d['median_ms'], [(r['ms_per_step'], r['search_ms'], r['identical']) for r in d['runs']])
" | tee gpurun_out/r4_ab_$tag.log
