#!/usr/bin/env python
"""Randomised sweep (GPU box) of search-driven composition against composing first: random lexicon / grammar
pairs, with / without the tee model and weight pushing, random pruning, several streams expanding one shared
network at once, a second batch on the grown network.  Hypotheses must be bit-identical.

    python tests/manual/fuzz_lazy.py [n_cases] [first_seed]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from juicer_amd import capi, synth                                      # noqa: E402

BEAMS = [dict(main_beam=150.0), dict(main_beam=300.0), dict(main_beam=100.0, end_beam=70.0, word_beam=50.0),
         dict(main_beam=200.0, max_hyps=300), dict(main_beam=0.0, max_hyps=2000), dict(main_beam=250.0, end_beam=200.0)]


def same(a, b):
    if a.n != b.n or not (np.array_equal(a.label, b.label) and np.array_equal(a.time, b.time)):
        return False
    return all(np.array_equal(np.asarray(getattr(a, k), np.float32).view(np.uint32), np.asarray(getattr(b, k), np.float32).view(np.uint32))
               for k in ("score", "ac", "lm", "tot_score", "tot_lm"))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
    rng = np.random.default_rng(s0)
    t0 = time.time()
    bad = 0
    utts = 0
    for case in range(n):
        seed = s0 + case
        with_sp = bool(rng.integers(2))
        pushing = bool(rng.integers(2))
        n_words = int(rng.integers(8, 120))
        am = synth.make_models(seed, n_gmm=int(rng.integers(30, 200)), n_hmm=int(rng.integers(20, 60)), n_mix=int(rng.integers(1, 4)),
                               n_tm=8, sep=float(rng.uniform(0.4, 1.0)), with_tee=with_sp)
        cl, g = synth.make_cl_g(seed, am, n_words=n_words, n_succ=int(rng.integers(2, 8)), n_tri=int(rng.integers(0, 60)), with_sp=with_sp)
        ncl, ng = capi.Network.from_synth(cl, 1.0, 0.0), capi.Network.from_synth(g, float(rng.uniform(1.0, 12.0)), 0.0)
        models = capi.Models.from_htk(am)
        kw = dict(BEAMS[int(rng.integers(len(BEAMS)))])
        ns = int(rng.integers(1, 6))
        feats = [synth.sample_utterance(seed * 100 + u, g, am, int(rng.integers(2, 10)))[0] for u in range(2 * ns)]
        static = capi.Network.compose(ncl, ng, pushing=pushing)
        lazy = capi.Network.lazy(ncl, ng, models, max_states=1 << 18, max_arcs=1 << 20, pushing=pushing)
        want = capi.Decoder(static, models, max_streams=ns, **kw).decode_batch(feats)
        dec = capi.Decoder(lazy, models, max_streams=ns, **kw)
        got = dec.decode_batch(feats[:ns]) + dec.decode_batch(feats[ns:])
        ok = all(same(a, b) for a, b in zip(got, want)) and lazy.lazy_size()[0] <= static.n_states
        utts += len(feats)
        if not ok:
            bad += 1
            print("MISMATCH seed %d: words %d sp %d pushing %d %s streams %d" % (seed, n_words, with_sp, pushing, kw, ns), flush=True)
    print("%d cases, %d utterances, %d mismatching cases, %.0f s" % (n, utts, bad, time.time() - t0))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
