#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "two_batches or scores_ahead" > gpurun_out/r4_tests_pipe.log 2>&1
tail -3 gpurun_out/r4_tests_pipe.log
python tools/pf_ab.py c2 2 two= two_r16=JD_SCORE_RESERVE:16 two_r32=JD_SCORE_RESERVE:32 two_r48=JD_SCORE_RESERVE:48 two_w3=JD_BG_WEIGHT:0.3,JD_SCORE_RESERVE:32 two_w7=JD_BG_WEIGHT:0.7,JD_SCORE_RESERVE:32 two_w10=JD_BG_WEIGHT:1.0,JD_SCORE_RESERVE:32 two_bg8=JD_BG_CW:8,JD_SCORE_RESERVE:32 two_fg12=JD_FG_CW:12,JD_SCORE_RESERVE:32 two_ab=JD_MODEL_A:24.6,JD_MODEL_B:61.4,JD_SCORE_RESERVE:32 ahead= 2>gpurun_out/r4_ab_pipe.err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['variant'], d['median_ms'], [(r['ms_per_step'], r['search_ms'], r['gmm_ms'], r['search_launches'], r['ahead_frames'], r['identical']) for r in d['runs']])
" | tee gpurun_out/r4_ab_pipe.log
JD_VERBOSE=1 JD_SCORE_RESERVE=32 python tools/pf_ab.py c2 1 two= 2>&1 | grep -E "k_search|cut short" | tail -3
