"""Timing of the configs[1] workload on the GPU box (not a test)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from juicer_amd import synth, capi
U = int(sys.argv[1]) if len(sys.argv) > 1 else 64
beam = float(sys.argv[2]) if len(sys.argv) > 2 else 150.0
am, net, feats, _ = synth.config_c2(n_utts=U)
dec = capi.Decoder(capi.Network.from_synth(net), capi.Models.from_htk(am), main_beam=beam, max_streams=U)
for it in range(int(sys.argv[3]) if len(sys.argv) > 3 else 3):
    t0 = time.time(); hyps = dec.decode_batch(feats); t1 = time.time()
    print("iter", it, "wall %.3fs" % (t1 - t0), dec.last_timing())
nf = sum(h.stats["n_frames"] for h in hyps)
print("frames", nf, "ok", sum(h.n > 0 for h in hyps), "of", U)
st = {k: sum(h.stats[k] for h in hyps) for k in hyps[0].stats}
print("per stream-frame:", {k: round(v / nf, 1) for k, v in st.items() if k != "n_frames"})
