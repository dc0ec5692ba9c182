"""Per-block timeline of k_phase_a / k_expand<0> at one lock-step frame (GPU box diagnostic)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from juicer_amd import synth, capi
U, frame = 64, int(sys.argv[1]) if len(sys.argv) > 1 else 250
am, net, feats, _ = synth.config_c2(n_utts=U)
dec = capi.Decoder(capi.Network.from_synth(net), capi.Models.from_htk(am), main_beam=150.0, max_streams=U)
dec.decode_batch(feats)
dec.debug_trace(frame)
dec.decode_batch(feats)
buf = dec.debug_trace(frame, fetch=True)
for name, b in (("k_phase_a", buf[0]), ("k_expand<0>", buf[1])):
    rec = b[b[:, 0] != 0]
    if len(rec) == 0:
        print(name, "no records"); continue
    t0 = rec[:, 0].min()
    work = rec[rec[:, 3] > 0]
    empty = rec[rec[:, 3] == -1]
    us = lambda x: x / 100.0
    print("%s: %d blocks recorded (%d working, %d empty)" % (name, len(rec), len(work), len(empty)))
    print("   block start (us after first): working p50 %.1f p90 %.1f max %.1f | empty p50 %.1f max %.1f" % (
        us(np.median(work[:, 0] - t0)), us(np.percentile(work[:, 0] - t0, 90)), us((work[:, 0] - t0).max()),
        us(np.median(empty[:, 0] - t0)) if len(empty) else -1, us((empty[:, 0] - t0).max()) if len(empty) else -1))
    print("   setup (ctl loads) us: p50 %.2f p90 %.2f max %.2f" % tuple(us(np.percentile(work[:, 1] - work[:, 0], q)) for q in (50, 90, 100)))
    print("   work us:              p50 %.2f p90 %.2f max %.2f" % tuple(us(np.percentile(work[:, 2] - work[:, 1], q)) for q in (50, 90, 100)))
    print("   epilogue us:          p50 %.2f p90 %.2f max %.2f" % tuple(us(np.percentile(work[:, 3] - work[:, 2], q)) for q in (50, 90, 100)))
    print("   last block end: %.1f us after first start" % us(work[:, 3].max() - t0))
    # concurrency profile: how many working blocks are alive at 10 sample times
    end = work[:, 3].max()
    for frac in (0.1, 0.3, 0.5, 0.7, 0.9):
        t = t0 + (end - t0) * frac
        print("      t=%5.1f us: %4d working blocks alive" % (us(t - t0), int(((work[:, 0] <= t) & (work[:, 3] >= t)).sum())))

# finer phase-A stamps (first unit of each block): loads+compute | to barrier | barrier+atomic | stores
a, f = buf[0], buf[2]
m = (a[:, 3] > 0) & (f[:, 0] != 0)
if m.any():
    us = lambda x: x / 100.0
    t_setup = a[m, 1]; t_c = f[m, 0]; t_b = f[m, 1]; t_a = f[m, 2]; t_end = a[m, 2]
    for name, d in (("record/token/ll loads + state update", t_c - t_setup), ("exit shuffles, ballots", t_b - t_c),
                    ("barrier + packed atomic + barrier", t_a - t_b), ("stores + (further units) ", t_end - t_a)):
        print("   phase A %-38s p50 %.2f p90 %.2f us" % (name, us(np.median(d)), us(np.percentile(d, 90))))

# finer expansion stamps (thread 0's group, first unit of each block)
a, f = buf[1], buf[3]
m = (a[:, 3] > 0) & (f[:, 3] != 0)
if m.any():
    us = lambda x: x / 100.0
    for name, d in (("item info/token load", f[m, 0] - a[m, 1]), ("winner check + row bounds", f[m, 1] - f[m, 0]),
                    ("path record / final state", f[m, 2] - f[m, 1]), ("arc walk (thread 0's wave)", f[m, 3] - f[m, 2]),
                    ("stage flush + stats", a[m, 3] - f[m, 3])):
        print("   expand<0> %-30s p50 %.2f p90 %.2f max %.2f us" % (name, us(np.median(d)), us(np.percentile(d, 90)), us(d.max())))
