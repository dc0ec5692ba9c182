#!/usr/bin/env python
"""Throughput of N serial IDecoder callers (threads) on one decoder through the broker, at configs[1] (GPU box):
python tools/broker_bench.py [callers ...]   ->  frames/s per caller count against one batch of 64"""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401
from juicer_amd import capi, synth  # noqa: E402

counts = [int(a) for a in sys.argv[1:]] or [1, 4, 16, 32, 64]
CHUNK = int(os.environ.get("JD_BENCH_PUSH_FRAMES", "64"))
am, net, feats, _ = synth.config_c2(n_utts=64)
gnet, gam = capi.Network.from_synth(net), capi.Models.from_htk(am)
frames = sum(f.shape[0] for f in feats)
bd = capi.Decoder(gnet, gam, main_beam=150.0, max_streams=64)
bd.decode_batch(feats)
t0 = time.perf_counter()
bd.decode_batch(feats)
t_batch = time.perf_counter() - t0
bd.close()
print("one batch of 64 (jd_decode_batch, host features): %.0f frames/s" % (frames / t_batch))


def drive(broker, utts, chunk):
    c = broker.open()
    for x in utts:
        broker.init(c)
        for i in range(0, x.shape[0], chunk):
            broker.push(c, x[i:i + chunk])
        broker.finish(c)
    broker.close_client(c)


for n in counts:
    dec = capi.Decoder(gnet, gam, main_beam=150.0, max_streams=n)
    broker = capi.Broker(dec)
    best = None
    for rep in range(1):
        # every caller decodes the whole list, each from another starting point: equal work per caller, so that the
        # rate is the steady state's and not the tail of the caller that drew the longest utterances
        th = [threading.Thread(target=drive, args=(broker, feats[(t * 64) // n:] + feats[:(t * 64) // n], CHUNK)) for t in range(n)]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    st = broker.stats()
    tk = max(st["ticks"], 1)
    if st["resident"]:
        print("%3d callers: %8.0f frames/s = %.2f of the batch rate; resident kernel, clusters of %d: %.0f frames per chunk, %.0f us from its post to its report "
              "(%.1f us per frame; the cluster's own clock: %.1f), %.0f us until the next chunk of the utterance is posted; staging call %.0f us, "
              "result fetch %.0f us per utterance"
              % (n, n * frames / best, (n * frames / best) / (frames / t_batch), st["resident"], st["frames"] / tk, st["us_search"] / tk,
                 st["us_search"] / max(st["frames"], 1), st["us_coalesce"] / max(st["frames"], 1), st["us_idle"] / tk, st["us_push"] / tk,
                 st["us_finish"] / (64.0 * n)))
        broker.close()
        dec.close()
        continue
    print("%3d callers: %8.0f frames/s = %.2f of the batch rate; %.1f streams and %.0f frames per tick; worker us per tick: idle %.0f, coalesce %.0f, "
          "init %.0f, push %.0f (search kernel %.0f), finish %.0f"
          % (n, n * frames / best, (n * frames / best) / (frames / t_batch), st["stream_ticks"] / tk, st["frames"] / tk, st["us_idle"] / tk,
             st["us_coalesce"] / tk, st["us_init"] / tk, st["us_push"] / tk, st["us_search"] / tk, st["us_finish"] / tk))
    broker.close()
    dec.close()
