import os, sys
sys.path.insert(0, "/root/repo")
os.environ["JD_VERBOSE"] = "1"
from juicer_amd import capi, synth
import bench, torch
for name, mk in (("c2", lambda: synth.config_c2(seed=0, n_utts=2)), ("north", lambda: synth.config_c4(seed=0, n_utts=2, n_words=10000, n_tri_hist=100_000)),
                 ("c3", lambda: synth.config_c4(seed=0, n_utts=2))):
    am, net, feats, _ = mk()
    print(name, flush=True)
    dec = capi.Decoder(capi.Network.from_synth(net), capi.Models.from_htk(am), main_beam=150.0, max_streams=1)
    dec.close()
