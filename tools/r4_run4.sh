#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
python tools/pf_ab.py c2 3 two= two_r40=JD_SCORE_RESERVE:40 two_r48=JD_SCORE_RESERVE:48 two_r56=JD_SCORE_RESERVE:56 two_w4=JD_BG_WEIGHT:0.4 two_w6=JD_BG_WEIGHT:0.6 two_bg3=JD_BG_CW:3 two_bg6=JD_BG_CW:6 two_fg10=JD_FG_CW:10 two_fg6=JD_FG_CW:6 two_a=JD_MODEL_A:20 two_b=JD_MODEL_B:240 ahead= 2>gpurun_out/r4_ab_pipe.err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['variant'], d['median_ms'], [(r['ms_per_step'], r['search_ms'], r['gmm_ms'], r['search_launches'], r['ahead_frames'], r['identical']) for r in d['runs']])
" | tee gpurun_out/r4_ab_pipe2.log
