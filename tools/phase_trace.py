#!/usr/bin/env python
"""In-kernel cycle accounting of k_search on the bench workload (or --arcs/--beam/--utts):
per workgroup {phase A, barrier wait, phase X, barrier wait} in 100 MHz ticks, averaged."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from juicer_amd import capi, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--utts", type=int, default=64)
ap.add_argument("--arcs", type=int, default=1_000_000)
ap.add_argument("--beam", type=float, default=150.0)
ap.add_argument("--max-hyps", type=int, default=0)
ap.add_argument("--trim", type=int, default=0, help="cut every utterance to this many frames (0 = keep)")
ap.add_argument("--c4", type=float, default=0.0, help="trigram-shaped graph of config_c4 at this scale (1.0 = configs[3]) instead of config_c2")
ap.add_argument("--clg", action="store_true", help="the configs[4] pair (lexicon tree o back-off trigram) composed on the device; with --lazy: by the search")
ap.add_argument("--lazy", action="store_true")
ap.add_argument("--two", action="store_true", help="two batches in flight (streams for two batches, announcements two ahead)")
args = ap.parse_args()
gnet = None
if args.clg:
    am = synth.make_models(0, n_gmm=3000, n_hmm=2000, n_mix=16, n_tm=8, sep=0.6, with_tee=True)
    cl, g = synth.make_cl_g(0, am, n_words=20000, n_succ=40, n_tri=200000, n_succ3=8, with_sp=True)
    ncl, ng = capi.Network.from_synth(cl, 1.0, 0.0), capi.Network.from_synth(g, 10.0, 0.0)
    gnet = (capi.Network.lazy(ncl, ng, capi.Models.from_htk(am), max_states=1 << 22, max_arcs=1 << 23) if args.lazy
            else capi.Network.compose(ncl, ng, max_states=1 << 26, max_arcs=1 << 27))
    feats = [synth.sample_utterance(100 + u, g, am, 8)[0] for u in range(args.utts)]
elif args.c4 > 0:
    am, net, feats, _ = synth.config_c4(n_utts=args.utts, n_words=int(20000 * args.c4 ** 0.5), n_tri_hist=int(400000 * args.c4))
else:
    am, net, feats, _ = synth.config_c2(n_utts=args.utts, target_arcs=args.arcs)
if args.trim:
    feats = [f[:args.trim] for f in feats]
dec = capi.Decoder(gnet if gnet is not None else capi.Network.from_synth(net), capi.Models.from_htk(am), main_beam=args.beam, max_hyps=args.max_hyps,
                   max_streams=(2 if args.two else 1) * args.utts)
if args.two:
    offs = np.zeros(len(feats) + 1, dtype=np.int64)
    offs[1:] = np.cumsum([f.shape[0] for f in feats])
    d_feats = torch.from_numpy(np.concatenate(feats)).to("cuda:0")
    torch.cuda.synchronize()
    dec.prefetch_scores(d_feats.data_ptr(), offs, 0)
    for _ in range(4):
        dec.prefetch_scores(d_feats.data_ptr(), offs, 0)
        dec.decode_batch_device(d_feats.data_ptr(), offs, 0)
    dec.debug_trace(0)
    dec.prefetch_scores(d_feats.data_ptr(), offs, 0)
    h = dec.decode_batch_device(d_feats.data_ptr(), offs, 0)
else:
    dec.decode_batch(feats)
    dec.debug_trace(0)
    h = dec.decode_batch(feats)
tm = dec.last_timing()
buf = dec.debug_trace(0, fetch=True)
used = buf[buf[:, 8] > 0]
fr = used[:, 8].astype(np.float64)
names = ["lists A", "phase A", "wg wait A", "barrier 1", "lists X", "phase X", "wg wait X", "barriers X"]
print("workgroups that ran frames: %d, stream-frames per workgroup: mean %.0f" % (len(used), fr.mean()))
tot = 0.0
for k, n in enumerate(names):
    us = used[:, k] / 100.0 / fr
    tot += us.mean()
    print("%-11s mean %7.2f us/frame   min %7.2f   max %7.2f" % (n, us.mean(), us.min(), us.max()))
print("sum %.2f us/frame;  search_ms %.2f for %d frames, %d launches, last cluster size %d"
      % (tot, tm["search_ms"], tm["search_frames"], tm["search_launches"], tm["cluster_wgs"]))
busy = used[:, :8].sum(axis=1) / 100.0 / 1e3                          # ms of accounted time per workgroup
print("workgroup-time accounted: %.1f ms of %d x %.2f ms = %.1f (%.0f %%); per workgroup: min %.1f, median %.1f, max %.1f ms; frames searched ahead %d"
      % (busy.sum(), len(used), tm["search_ms"], len(used) * tm["search_ms"], 100.0 * busy.sum() / max(len(used) * tm["search_ms"], 1e-9),
         busy.min(), float(np.median(busy)), busy.max(), tm.get("ahead_frames", 0)))
if used[:, 9:16].any():                                               # a -DJD_FINE build: hops inside a phase
    if used[:, 14].any():                                             # JD_FINE=2: phase X (slots 5, 6 are counts)
        names = ["X items", "X rows + key + Path", "X winners", "X prefix + first arcs", "X arc passes"]
        print("  batches per frame %.2f, arc passes per frame %.2f (thread 0's wave)" % ((used[:, 14] / fr).mean(), (used[:, 15] / fr).mean()))
    else:
        names = ["A wait for stage K", "A item + next record", "A arithmetic", "A stage K + stores issue"]
        print("  passes per frame %.2f (thread 0's wave)" % (used[:, 13] / fr).mean())
    for k, n in enumerate(names):
        us = used[:, 9 + k] / 100.0 / fr
        print("  %-22s mean %7.2f us/frame (thread 0's wave, memory counters drained after every hop)" % (n, us.mean()))
st = {k: sum(x.stats[k] for x in h) for k in h[0].stats}
print({k: round(v / tm["search_frames"], 1) for k, v in st.items() if k.startswith("tot")})
