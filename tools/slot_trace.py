#!/usr/bin/env python
"""The batch pipeline (JD_FLOW_RESIDENT) at configs[1] with the in-kernel cycle accounting on: what a slot's frame costs,
phase by phase, on its own clock; how busy the slots are; frames/s.  tools/slot_trace.py --slots 320 [--depth 6] [--steps 30]
JD_DEV=1 JD_RES_SLOT=0: k_resident's one-workgroup clusters instead of the slot kernel."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from juicer_amd import capi, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--slots", type=int, default=320)
ap.add_argument("--depth", type=int, default=6)
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--utts", type=int, default=64)
ap.add_argument("--beam", type=float, default=150.0)
ap.add_argument("--max-hyps", type=int, default=0)
ap.add_argument("--no-trace", action="store_true")
ap.add_argument("--scoring", choices=("exact", "fast"), default="exact")
args = ap.parse_args()
am, net, feats, _ = synth.config_c2(n_utts=args.utts)
dec = capi.Decoder(capi.Network.from_synth(net), capi.Models.from_htk(am), main_beam=args.beam, max_hyps=args.max_hyps, max_streams=args.slots)
if args.scoring == "fast":
    dec.set_scoring(capi.SCORE_FAST)
dec.set_pipeline(capi.FLOW_RESIDENT, args.depth + 1, args.slots)
offs = np.zeros(len(feats) + 1, dtype=np.int64)
offs[1:] = np.cumsum([f.shape[0] for f in feats])
d_feats = torch.from_numpy(np.concatenate(feats)).to("cuda:0")
torch.cuda.synchronize()
if not args.no_trace:
    dec.debug_trace(0)
for _ in range(args.depth):
    dec.prefetch_scores(d_feats.data_ptr(), offs, 0)
for _ in range(args.depth + 4):
    dec.prefetch_scores(d_feats.data_ptr(), offs, 0)
    dec.decode_batch_device(d_feats.data_ptr(), offs, 0)
dec.quiesce(); torch.cuda.synchronize()
if not args.no_trace:
    dec.debug_trace(0)                                                  # (clears the sums)
p0 = dec.pipeline_stats()
t0 = time.perf_counter()
for _ in range(args.steps):
    dec.prefetch_scores(d_feats.data_ptr(), offs, 0)
    dec.decode_batch_device(d_feats.data_ptr(), offs, 0)
dec.quiesce(); torch.cuda.synchronize()
dt = time.perf_counter() - t0
p1 = dec.pipeline_stats()
fr = p1["frames_searched"] - p0["frames_searched"]
busy = (p1["slot_busy_us"] - p0["slot_busy_us"]) / max(1.0, (p1["on_us"] - p0["on_us"]) * args.slots)
print("slots %d depth %d: %.3f M frames/s (%.2f ms per batch of %d frames); slots busy %.1f %% of the kernel's time; %.1f us per frame on a slot's clock"
      % (args.slots, args.depth, fr / dt / 1e6, dt * 1e3 * int(offs[-1]) / fr, int(offs[-1]), 100.0 * busy,
         (p1["slot_busy_us"] - p0["slot_busy_us"]) / max(1, fr)))
if not args.no_trace:
    buf = dec.debug_trace(0, fetch=True)
    used = buf[buf[:, 8] > 0]
    f = used[:, 8].astype(np.float64)
    names = ["frame start", "phase A", "wait A", "-", "lists X", "phase X", "wait X", "-"]
    tot = 0.0
    for k, n in enumerate(names):
        if n == "-":
            continue
        us = used[:, k] / 100.0 / f
        tot += us.mean()
        print("  %-11s mean %7.2f us/frame   min %7.2f   max %7.2f" % (n, us.mean(), us.min(), us.max()))
    print("  sum %.2f us/frame over %d slots that ran frames" % (tot, len(used)))
    if used[:, 15].sum() > 0:                                          # a SLOT_FINE build: wave 0's passes over record chunks, every stage waited for
        n = used[:, 15].astype(np.float64).sum()
        for k, nm in enumerate(["record trip", "key trip", "item trip", "arithmetic", "store issue", "store drain"]):
            print("  fine: %-12s %6.3f us per pass" % (nm, used[:, 9 + k].sum() / 100.0 / n))
        print("  fine: %.1f record passes of wave 0 per frame" % (n / f.sum()))
