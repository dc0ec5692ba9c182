#!/usr/bin/env python
"""BASELINE.json configs[3] on one MI355X: ~50M-arc trigram-shaped C.L.G, 5000 tied states x 16
mixtures, mainBeam 300 (wide-beam stress).  Decodes --utts utterances in lock-step on the GPU,
checks the first --oracle-utts of them against the CPU oracle, prints one JSON line.

    python tests/manual/run_c4.py [--scale 1.0] [--utts 8] [--oracle-utts 2] [--beam 300]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=1.0, help="shrinks the graph (1.0 = ~48M arcs)")
    ap.add_argument("--utts", type=int, default=8)
    ap.add_argument("--oracle-utts", type=int, default=2)
    ap.add_argument("--beam", type=float, default=300.0)
    ap.add_argument("--max-hyps", type=int, default=0)
    ap.add_argument("--words", type=int, nargs=2, default=(9, 28))
    ap.add_argument("--passes", type=int, default=2)
    args = ap.parse_args()

    from juicer_amd import build as jbuild, capi, synth
    from helpers import assert_hyp_matches, bit_exact
    jbuild.build()
    t0 = time.time()
    am, net, feats, words = synth.config_c4(
        n_utts=args.utts, n_words=max(200, int(20000 * args.scale ** 0.5)),
        n_tri_hist=max(100, int(400_000 * args.scale)), utt_words=tuple(args.words))
    t_gen = time.time() - t0
    deg = np.bincount(net.src, minlength=net.n_states)
    gnet = capi.Network.from_synth(net)
    gam = capi.Models.from_htk(am)
    dec = capi.Decoder(gnet, gam, main_beam=args.beam, max_hyps=args.max_hyps, max_streams=args.utts)
    frames = int(sum(f.shape[0] for f in feats))
    best = None
    for _ in range(args.passes):
        t0 = time.time()
        hyps = dec.decode_batch(feats)
        dt = time.time() - t0
        tm = dec.last_timing()
        best = dt if best is None else min(best, dt)
    st = {k: int(sum(h.stats[k] for h in hyps)) for k in hyps[0].stats}
    out = {"workload": "configs[3]: trigram-shaped C.L.G, %d arcs, %d states, %d tied states x %d mix, "
                       "%d utterances, mainBeam %g, maxHyps %d" % (net.n_arcs, net.n_states, am.n_gmm,
                                                                 am.max_mix, args.utts, args.beam, args.max_hyps),
           "max_out_degree": int(deg.max()), "states_with_degree_ge_1000": int((deg >= 1000).sum()),
           "eps_arc_fraction": round(float((net.ilab == 0).mean()), 4),
           "gen_seconds": round(t_gen, 1), "frames": frames,
           "gpu_seconds_per_pass_host_inclusive": round(best, 3), "frames_per_sec": round(frames / best, 1),
           "search_ms": round(tm["search_ms"], 2), "gmm_ms": round(tm["gmm_ms"], 2),
           "search_launches": int(tm["search_launches"]), "relaunches": int(tm["relaunches"]),
           "per_frame": {k: round(st[k] / frames, 1) for k in ("tot_insts_in", "tot_proc_emit_hyps", "tot_proc_end_hyps",
                                                               "tot_arcs_visited", "tot_paths")},
           "n_hyp_words": [int(h.n) for h in hyps],
           "generating_words_recovered": int(sum(int(np.array_equal(h.label[::-1], w)) for h, w in zip(hyps, words)))}
    if args.oracle_utts > 0:
        from oracle.oracle import OracleAM, OracleDecoder, OracleNet
        od = OracleDecoder(OracleNet(net), OracleAM(am), main_beam=args.beam, max_hyps=args.max_hyps)
        secs, fr, same, exact, ties = 0.0, 0, 0, 0, 0
        for u in range(min(args.oracle_utts, args.utts)):
            o = od.decode(feats[u])
            secs += o.cpu_seconds; fr += feats[u].shape[0]
            ties += int(o.stats["ties"])
            ok = hyps[u].n == o.n and np.array_equal(hyps[u].label, o.label) and np.array_equal(hyps[u].time, o.time)
            same += int(ok)
            if o.stats["ties"] == 0:
                assert_hyp_matches(hyps[u], o, "c4 utt %d" % u)
            exact += int(bit_exact(hyps[u], o))
        out["oracle"] = {"utts": min(args.oracle_utts, args.utts), "frames": fr, "cpu_frames_per_sec": round(fr / secs, 2),
                         "identical_1best": same, "bit_exact_scores": exact, "ties": ties}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
