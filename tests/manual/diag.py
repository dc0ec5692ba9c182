"""GPU-box diagnosis: print GPU vs oracle results side by side (not a test)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from juicer_amd import synth, capi
from oracle.oracle import OracleNet, OracleAM, OracleDecoder

def show(tag, g, o):
    print(tag, "GPU n=%d tot=%r" % (g.n, (g.tot_score, g.tot_ac, g.tot_lm)))
    print(tag, "ORA n=%d tot=%r" % (o.n, (o.tot_score, o.tot_ac, o.tot_lm)))
    print("   GPU labels", g.label[::-1], "times", g.time[::-1])
    print("   ORA labels", o.label[::-1], "times", o.time[::-1])
    print("   GPU stats", g.stats)
    print("   ORA stats", o.stats)

for name, cfg in (("toy", synth.config_toy()), ("small", synth.config_small())):
    am, net, feats, words = cfg
    gnet, gam = capi.Network.from_synth(net), capi.Models.from_htk(am)
    onet, oam = OracleNet(net), OracleAM(am)
    x = np.concatenate(feats)[:500]
    g = gam.score_frames(x); o = oam.score_frames(x)
    d = g.view(np.uint32) != o.view(np.uint32)
    print(name, "gmm mismatches", int(d.sum()), "of", d.size, "maxabs", float(np.abs(g - o).max()))
    for kw in (dict(), dict(main_beam=150.0), dict(main_beam=150.0, max_hyps=200)):
        gd = capi.Decoder(gnet, gam, max_streams=len(feats), **kw)
        od = OracleDecoder(onet, oam, **kw)
        t0 = time.time(); gs = gd.decode_batch(feats); t1 = time.time()
        print(name, kw, "gpu wall %.3fs" % (t1 - t0), gd.last_timing())
        for u in range(len(feats)):
            show("%s utt%d %s" % (name, u, kw), gs[u], od.decode(feats[u]))
