#!/usr/bin/env python
"""Search-driven composition (jd_net_create_lazy) next to composing first (jd_net_compose) at configs[4]
size: time to the first hypothesis, throughput while the graph is still growing and once it has grown,
how much of the full composition the search ever asked for, and that the hypotheses are identical."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from juicer_amd import capi, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--words", type=int, default=20000)
ap.add_argument("--succ", type=int, default=40)
ap.add_argument("--tri", type=int, default=200000)
ap.add_argument("--utts", type=int, default=64)
ap.add_argument("--mix", type=int, default=16)
ap.add_argument("--beam", type=float, default=200.0)
ap.add_argument("--sets", type=int, default=3)
args = ap.parse_args()
am = synth.make_models(0, n_gmm=3000, n_hmm=2000, n_mix=args.mix, n_tm=8, sep=0.6, with_tee=True)
cl, g = synth.make_cl_g(0, am, n_words=args.words, n_succ=args.succ, n_tri=args.tri, n_succ3=8, with_sp=True)
ncl, ng = capi.Network.from_synth(cl, 1.0, 0.0), capi.Network.from_synth(g, 10.0, 0.0)
models = capi.Models.from_htk(am)
sets = [[synth.sample_utterance(100 + 1000 * k + u, g, am, 8)[0] for u in range(args.utts)] for k in range(args.sets)]
frames = [sum(f.shape[0] for f in s) for s in sets]
out = {"cl_arcs": int(cl.n_arcs), "g_arcs": int(g.n_arcs), "utts_per_set": args.utts, "frames_per_set": frames, "beam": args.beam}


def timed(dec, feats):
    t0 = time.perf_counter()
    h = dec.decode_batch(feats)
    return h, time.perf_counter() - t0


# compose first, then search
t0 = time.perf_counter()
static = capi.Network.compose(ncl, ng, max_states=1 << 26, max_arcs=1 << 27)
t_comp = time.perf_counter() - t0
dec = capi.Decoder(static, models, main_beam=args.beam, max_streams=args.utts)
want, st_t = [], []
for k, s in enumerate(sets):
    h, dt = timed(dec, s)
    want.append(h); st_t.append(dt)
h, dt = timed(dec, sets[0])
out["static"] = {"compose_s": round(t_comp, 3), "states": static.n_states, "arcs": static.n_arcs,
                 "decode_s": [round(x, 4) for x in st_t], "again_set0_s": round(dt, 4), "frames_per_s_warm": round(frames[0] / dt, 1)}
dec.close()
del dec, static

# search-driven
t0 = time.perf_counter()
lazy = capi.Network.lazy(ncl, ng, models, max_states=1 << 22, max_arcs=1 << 23)
t_create = time.perf_counter() - t0
dec = capi.Decoder(lazy, models, main_beam=args.beam, max_streams=args.utts)
lz_t, sizes, same, lz_search = [], [lazy.lazy_size()], 0, []
for k, s in enumerate(sets):
    h, dt = timed(dec, s)
    lz_search.append(round(dec.last_timing()["search_ms"] / 1e3, 4))
    lz_t.append(dt); sizes.append(lazy.lazy_size())
    same += sum(int(a.n == b.n and np.array_equal(a.label, b.label) and np.array_equal(a.time, b.time)
                    and np.array_equal(np.asarray(a.score, np.float32).view(np.uint32), np.asarray(b.score, np.float32).view(np.uint32)))
                for a, b in zip(h, want[k]))
h, dt = timed(dec, sets[0])
out["lazy"] = {"create_s": round(t_create, 3), "decode_s": [round(x, 4) for x in lz_t], "search_s": lz_search, "again_set0_s": round(dt, 4),
               "frames_per_s_warm": round(frames[0] / dt, 1), "states_arcs_after_each_set": sizes,
               "identical_hyps": same, "of": args.sets * args.utts,
               "fraction_of_full_composition": round(sizes[-1][0] / out["static"]["states"], 4)}
print(json.dumps(out))
