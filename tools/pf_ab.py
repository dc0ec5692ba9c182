#!/usr/bin/env python
"""Scoring one batch ahead (jd_dec_prefetch_scores) against the serial order, on one box, alternating:
python tools/pf_ab.py [c2|north] [rounds]   ->  ms per step of every variant, search / scoring spans, identical results."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from juicer_amd import capi, synth  # noqa: E402

leg = sys.argv[1] if len(sys.argv) > 1 else "c2"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda", 0)
if leg == "c2":
    am, net, feats, _ = synth.config_c2(seed=0, n_utts=64, target_arcs=1_000_000)
    beam = 150.0
else:
    am, net, feats, _ = synth.config_c4(seed=0, n_utts=64, n_words=10000, n_tri_hist=100_000)
    beam = 200.0
gnet, gam = capi.Network.from_synth(net), capi.Models.from_htk(am)
offs = np.zeros(len(feats) + 1, dtype=np.int64)
offs[1:] = np.cumsum([f.shape[0] for f in feats])
d_feats = torch.from_numpy(np.concatenate(feats)).to(dev)
torch.cuda.synchronize()
# extra variants from the command line: name=ENV1:val,ENV2:val (scored ahead)
extra = {}
for a in sys.argv[3:]:
    nm, _, kv = a.partition("=")
    extra[nm] = (not nm.startswith("serial"), dict(x.split(":") for x in kv.split(",") if x))
KNOBS = ("JD_PF_REBALANCE", "JD_REBALANCE", "JD_MODEL_A", "JD_MODEL_B", "JD_PLAN", "JD_MODEL2_A", "JD_MODEL2_B", "JD_PF_GMM_WEIGHT", "JD_PLAN_MIN_CW", "JD_GMM_WAVES", "JD_REBALANCE_FRAC", "JD_XL_SLACK", "JD_CW", "JD_XCH", "JD_EXP", "JD_PIPELINE", "JD_SCORE_RESERVE", "JD_FG_CW", "JD_BG_CW", "JD_BG_REBALANCE", "JD_BG_WAIT_US", "JD_BG_WEIGHT")
variants = extra if extra else {"serial": (False, {}), "ahead": (True, {}), "ahead, re-plan held": (True, {"JD_PF_REBALANCE": "0"}),
            "ahead, re-planned at will": (True, {"JD_PF_REBALANCE": "1"}), "serial, never re-planned": (False, {"JD_REBALANCE": "0"})}
# a variant named two... : two batches in flight (streams for two batches, announcements two batches ahead)


def make(env, two=False):
    for k in KNOBS:
        os.environ.pop(k, None)
    os.environ.update(env)
    return capi.Decoder(gnet, gam, main_beam=beam, device=0, max_streams=(2 if two else 1) * len(feats))


want = None
res = {k: [] for k in variants}
for r in range(rounds + 1):
    for name, (pf, env) in variants.items():
        two = name.startswith("two")
        dec = make(env, two)                                           # (one decoder at a time: each sizes its arenas from the free HBM)
        steps = 4
        if two and pf:
            dec.prefetch_scores(d_feats.data_ptr(), offs, 0)
        for _ in range(3 if two else 2):                               # warm-up: load learnt, the first table announced
            if pf:
                dec.prefetch_scores(d_feats.data_ptr(), offs, 0)
            hy = dec.decode_batch_device(d_feats.data_ptr(), offs, 0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        acc = {"search_ms": 0.0, "gmm_ms": 0.0, "gmm_wait_ms": 0.0, "search_launches": 0, "prefetched": 0, "ahead_frames": 0}
        for _ in range(steps):
            if pf:
                dec.prefetch_scores(d_feats.data_ptr(), offs, 0)
            hy = dec.decode_batch_device(d_feats.data_ptr(), offs, 0)
            tm = dec.last_timing()
            for k in acc:
                acc[k] += tm[k]
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps * 1e3
        sig = [(h.n, h.label.tobytes(), h.time.tobytes(), np.asarray(h.score, np.float32).tobytes()) for h in hy]
        if want is None:
            want = sig
        same = sum(int(a == b) for a, b in zip(sig, want))
        dec.close()
        if r > 0:
            res[name].append({"ms_per_step": round(dt, 3), **{k: round(v / steps, 3) for k, v in acc.items()}, "identical": same})
for name in variants:
    ms = sorted(x["ms_per_step"] for x in res[name])
    print(json.dumps({"variant": name, "median_ms": ms[len(ms) // 2], "runs": res[name]}))
