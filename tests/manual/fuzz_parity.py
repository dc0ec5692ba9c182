#!/usr/bin/env python
"""Randomised parity sweep (GPU box): many small seeded problems - both hub shapes, with / without
the tee model, 5-state and mixed-topology HMMs, random pruning settings - each decoded in one
batch and compared with the CPU oracle.  Not part of the test suite; prints a summary.

    python tests/manual/fuzz_parity.py [n_cases] [first_seed]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import assert_hyp_matches, bit_exact                      # noqa: E402
from juicer_amd import capi, synth                                      # noqa: E402
from oracle.oracle import OracleAM, OracleDecoder, OracleNet            # noqa: E402

BEAMS = [dict(main_beam=150.0), dict(main_beam=200.0), dict(main_beam=100.0, end_beam=70.0, word_beam=50.0),
         dict(main_beam=150.0, max_hyps=200), dict(max_hyps=400), dict(main_beam=120.0, start_beam=100.0, max_hyps=150),
         dict(main_beam=250.0, end_beam=200.0)]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    rng = np.random.default_rng(s0)
    t0 = time.time()
    checked = exact = ties = 0
    for case in range(n):
        seed = s0 + case
        hub = "tree" if rng.random() < 0.6 else "flat"
        with_sp = bool(rng.random() < 0.7)
        if rng.random() < 0.35:
            am = synth.make_models_mixed(seed, n_gmm=260, n_hmm=40, n_mix=int(rng.integers(1, 5)), with_tee=with_sp, sep=0.7)
            kind = "mixed"
        else:
            am = synth.make_models(seed, n_gmm=100, n_hmm=45, n_mix=int(rng.integers(1, 6)), n_tm=8, sep=0.6, with_tee=with_sp)
            kind = "5state"
        net = synth.make_wfst(seed + 100, am, n_words=int(rng.integers(20, 90)), n_succ=int(rng.integers(2, 9)),
                              with_sp=with_sp, hub=hub, eps_word_frac=float(rng.choice([0.0, 0.05, 0.3])))
        feats = [synth.sample_utterance(seed + 1000 + u, net, am, int(rng.integers(3, 12)))[0] for u in range(int(rng.integers(1, 7)))]
        kw = dict(BEAMS[int(rng.integers(0, len(BEAMS)))])
        lm = float(rng.choice([1.0, 7.5])); pen = float(rng.choice([0.0, -2.0]))
        gnet = capi.Network.from_synth(net, lm, pen)
        gd = capi.Decoder(gnet, capi.Models.from_htk(am), max_streams=len(feats), **kw)
        gs = gd.decode_batch(feats)
        od = OracleDecoder(OracleNet(net, lm, pen), OracleAM(am), **kw)
        for u, x in enumerate(feats):
            o = od.decode_certified(x)            # fails (never skips) if a case depended on tie order
            what = "case %d (%s hub=%s sp=%s %s lm=%g pen=%g) utt %d" % (seed, kind, hub, with_sp, kw, lm, pen, u)
            ties += int(o.stats["ties"] > 0)
            assert_hyp_matches(gs[u], o, what)
            checked += 1
            exact += bit_exact(gs[u], o)
    print("fuzz: %d cases, %d utterances checked, %d bit-exact incl. scores, %d with order-dependent ties (certified), %.1f s"
          % (n, checked, exact, ties, time.time() - t0))


if __name__ == "__main__":
    main()
