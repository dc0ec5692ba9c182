#!/usr/bin/env python
"""jd_net_compose at size: separate C.L (lexicon tree) and G (back-off n-gram) -> composed graph on the
device; prints sizes and wall time, then decodes a few utterances through the static path."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from juicer_amd import capi, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--words", type=int, default=20000)
ap.add_argument("--succ", type=int, default=40)
ap.add_argument("--tri", type=int, default=200000)
ap.add_argument("--utts", type=int, default=8)
args = ap.parse_args()
am = synth.make_models(0, n_gmm=3000, n_hmm=2000, n_mix=4, n_tm=8, sep=0.6, with_tee=True)
t0 = time.time()
cl, g = synth.make_cl_g(0, am, n_words=args.words, n_succ=args.succ, n_tri=args.tri, n_succ3=8, with_sp=True)
print("generated C.L %d states / %d arcs, G %d states / %d arcs in %.1f s" % (cl.n_states, cl.n_arcs, g.n_states, g.n_arcs, time.time() - t0))
ncl, ng = capi.Network.from_synth(cl, 1.0, 0.0), capi.Network.from_synth(g, 10.0, 0.0)
for rep in range(2):
    t0 = time.time()
    net = capi.Network.compose(ncl, ng, max_states=1 << 27, max_arcs=1 << 28)
    dt = time.time() - t0
    print("jd_net_compose: %d states, %d arcs in %.3f s (%.1f M arcs/s incl. H2D/D2H)" % (net.n_states, net.n_arcs, dt, net.n_arcs / dt / 1e6))
feats = [synth.sample_utterance(100 + u, g, am, 8)[0] for u in range(args.utts)]
dec = capi.Decoder(net, capi.Models.from_htk(am), main_beam=200.0, max_streams=args.utts)
t0 = time.time()
hyps = dec.decode_batch(feats)
print("decoded %d utterances (%d frames) in %.3f s; words: %s" % (len(feats), sum(f.shape[0] for f in feats), time.time() - t0,
                                                                [int(h.n) for h in hyps]))
