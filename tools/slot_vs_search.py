#!/usr/bin/env python
"""The slot kernel as a plain launch (JD_SLOT_BATCH=1) against k_search on a graph with long rows and epsilon closures (8 utterances of
the north-star workload): words, times, scores bit for bit and the reference's statistics must be the same.  (Arcs visited and Path
records are build counters that depend on the order arrivals are expanded in: tests/helpers.py LE_KEYS.)  GPU box."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from helpers import STAT_KEYS  # noqa: E402
os.environ["JD_DEV"] = "1"
from juicer_amd import capi, synth
am, net, feats, _ = synth.config_c4(seed=0, n_utts=8, n_words=10000, n_tri_hist=100_000)
models, network = capi.Models.from_htk(am), capi.Network.from_synth(net)
outs = []
for sb in ("0", "1"):
    os.environ["JD_SLOT_BATCH"] = sb
    os.environ["JD_CW"] = "1" if sb == "1" else "64"                   # (one workgroup per stream: what the slot kernel serves)
    dec = capi.Decoder(network, models, main_beam=200.0, max_streams=8)
    out = dec.decode_batch(feats)
    print("JD_SLOT_BATCH", sb, "slot launches", dec.last_timing()["slot_launches"], "search ms", dec.last_timing()["search_ms"])
    assert (dec.last_timing()["slot_launches"] > 0) == (sb == "1")
    outs.append(out); dec.close()
f32 = lambda a: np.asarray(a, np.float32).view(np.uint32)
bad = 0
for a, b in zip(*outs):
    ok = a.n == b.n and list(a.label) == list(b.label) and list(a.time) == list(b.time) and np.array_equal(f32(a.score), f32(b.score)) and all(a.stats[k] == b.stats[k] for k in STAT_KEYS)
    bad += not ok
    if not ok: print("differs", a.stats, b.stats)
print("identical incl. the reference's statistics: %d of %d" % (len(outs[0]) - bad, len(outs[0])))
sys.exit(1 if bad else 0)
