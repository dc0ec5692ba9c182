#!/usr/bin/env python
"""Development: the kernels' own counters (jd_stats tot_recs_read ..) per stream-frame for one small workload through k_search and through the slot kernel."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["JD_DEV"] = "1"
from juicer_amd import capi, synth  # noqa: E402

am, net, feats, _ = synth.config_c2(seed=0, n_utts=6, target_arcs=60_000, n_gmm=300, n_hmm=800, n_mix=8, n_words=500)
for name, env in (("k_search", {}), ("slot", {"JD_CW": "1", "JD_SLOT_BATCH": "1"})):
    for k in ("JD_CW", "JD_SLOT_BATCH"):
        os.environ.pop(k, None)
    os.environ.update(env)
    gd = capi.Decoder(capi.Network.from_synth(net), capi.Models.from_htk(am), max_streams=len(feats), main_beam=150.0)
    hs = gd.decode_batch(feats)
    fr = sum(h.stats["n_frames"] for h in hs)
    tm = gd.last_timing()
    print(name, "slot_launches", tm["slot_launches"], "of", tm["search_launches"],
          json.dumps({k: round(sum(h.stats[k] for h in hs) / fr, 1) for k in hs[0].stats if k.startswith("tot_")}))
    gd.close()
