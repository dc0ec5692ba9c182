#!/usr/bin/env python
"""Streaming API (IDecoder protocol) on ONE stream: time per push of n frames, frames/s, x real time."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from juicer_amd import capi, synth  # noqa: E402

am, net, feats, _ = synth.config_c2(n_utts=4)
dec = capi.Decoder(capi.Network.from_synth(net), capi.Models.from_htk(am), main_beam=150.0, max_streams=1)
x = np.concatenate(feats)
for n in (1, 8, 32, 64, 128, 512):
    for rep in range(2):
        dec.stream_init(0)
        t0 = time.time()
        lat = []
        for pos in range(0, x.shape[0], n):
            t1 = time.time()
            dec.stream_push(0, x[pos:pos + n])
            lat.append(time.time() - t1)
        h = dec.stream_finish(0)
        dt = time.time() - t0
    lat = np.array(lat[1:]) * 1e3
    print("push of %3d frames: median %.3f ms, p99 %.3f ms per push; %.0f frames/s = %.1fx real time (%d words)"
          % (n, np.median(lat), np.percentile(lat, 99), x.shape[0] / dt, x.shape[0] / dt / 100.0, h.n))
