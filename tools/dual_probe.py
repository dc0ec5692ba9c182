#!/usr/bin/env python
"""Experiment: does splitting the lock-step batch into G independent groups (each with its own
HIP streams, driven by its own host thread) overlap the per-kernel latency chains?
Runs configs[1] with one 64-stream decoder, then with G decoders of 64/G streams in threads."""
import os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from juicer_amd import capi, synth

def main():
    U = 64
    am, net, feats, _ = synth.config_c2(n_utts=U)
    gnet = capi.Network.from_synth(net); gam = capi.Models.from_htk(am)
    frames = sum(f.shape[0] for f in feats)
    for G in (1, 2, 4):
        # length-balanced round-robin split so every group has the same longest utterance mix
        order = np.argsort([-f.shape[0] for f in feats])
        groups = [[feats[i] for i in order[g::G]] for g in range(G)]
        decs = [capi.Decoder(gnet, gam, main_beam=150.0, max_streams=U // G) for _ in range(G)]
        def run(g):
            decs[g].decode_batch(groups[g])
        for rep in range(3):
            ths = [threading.Thread(target=run, args=(g,)) for g in range(G)]
            t0 = time.perf_counter()
            for t in ths: t.start()
            for t in ths: t.join()
            dt = time.perf_counter() - t0
            print("G=%d rep %d: %.1f ms  %.0f frames/s (host-inclusive)" % (G, rep, dt * 1e3, frames / dt), flush=True)
        del decs

if __name__ == "__main__":
    main()
