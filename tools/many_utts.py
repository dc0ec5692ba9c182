#!/usr/bin/env python
"""More utterances than streams (configs[2]'s shape on one GPU): U utterances on S streams."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from juicer_amd import capi, synth  # noqa: E402

U = int(sys.argv[1]) if len(sys.argv) > 1 else 256
am, net, feats, _ = synth.config_c2(n_utts=U)
gnet, gam = capi.Network.from_synth(net), capi.Models.from_htk(am)
frames = sum(f.shape[0] for f in feats)
ref = None
for S in [int(x) for x in (sys.argv[2:] or ["64", "128", "256"])]:
    dec = capi.Decoder(gnet, gam, main_beam=150.0, max_streams=S)
    dec.decode_batch(feats[:S])
    best = None
    for _ in range(2):
        t0 = time.time(); h = dec.decode_batch(feats); dt = time.time() - t0
        best = dt if best is None else min(best, dt)
    tm = dec.last_timing()
    if ref is None:
        ref = h
    same = all(a.n == b.n and np.array_equal(a.label, b.label) and np.array_equal(a.score.view(np.uint32), b.score.view(np.uint32)) for a, b in zip(ref, h))
    print("%d utterances (%d frames) on %d streams: %.1f ms host-inclusive = %.0f frames/s (search %.1f ms, gmm %.1f ms, %d launches); identical to the first run: %s"
          % (U, frames, S, best * 1e3, frames / best, tm["search_ms"], tm["gmm_ms"], tm["search_launches"], same))
    dec.close()
