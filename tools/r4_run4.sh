#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "two_batches or scores_ahead or small_decode" > gpurun_out/r4_tests_pipe.log 2>&1
tail -3 gpurun_out/r4_tests_pipe.log
python tools/pf_ab.py c2 3 two= two_nochain=JD_BG_CHAIN:0 two_c3=JD_BG_CHAIN_FRAC:0.3 two_c7=JD_BG_CHAIN_FRAC:0.7 two_c9=JD_BG_CHAIN_FRAC:0.9 ahead= 2>gpurun_out/r4_ab_pipe.err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['variant'], d['median_ms'], [(r['ms_per_step'], r['search_ms'], r['gmm_ms'], r['search_launches'], r['ahead_frames'], r['identical']) for r in d['runs']])
" | tee gpurun_out/r4_ab_pipe3.log
python tools/phase_trace.py --two 2>&1 | tail -4 | head -2
python - <<'PY'
import numpy as np, torch
from juicer_amd import capi, synth
am, net, feats, _ = synth.config_c2(n_utts=64)
dec = capi.Decoder(capi.Network.from_synth(net), capi.Models.from_htk(am), main_beam=150.0, max_streams=64)
dec.decode_batch(feats)
dec.debug_cells(True)
dec.decode_batch(feats)
print("cells read / cells of the table:", dec.debug_cells(False))
PY
