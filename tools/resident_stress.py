#!/usr/bin/env python
"""Many decoders one after the other in one process, each starting the resident kernel (batch pipeline and broker): no start may
end up with its side stream or the null stream queued behind the kernel (jd_res_start probes for it and changes streams)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from juicer_amd import capi, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda", 0)
am, net, feats, _ = synth.config_small(n_utts=8)
gnet, gam = capi.Network.from_synth(net), capi.Models.from_htk(am)
offs = np.zeros(len(feats) + 1, dtype=np.int64)
offs[1:] = np.cumsum([f.shape[0] for f in feats])
d_feats = torch.from_numpy(np.concatenate(feats)).to(dev)
torch.cuda.synchronize()
want = capi.Decoder(gnet, gam, main_beam=150.0, max_streams=8).decode_batch(feats)
worst = 0.0
extra = []
for k in range(n):
    extra.append(torch.cuda.Stream(device=dev, priority=-(k % 2)))    # (the process keeps making streams, as applications do)
    with torch.cuda.stream(extra[-1]):
        torch.ones(16, device=dev).sum().item()
    t0 = time.perf_counter()
    os.environ["JD_PIPELINE"] = "3"; os.environ["JD_PIPE_DEPTH"] = "4"
    dec = capi.Decoder(gnet, gam, main_beam=150.0, max_streams=5)
    os.environ.pop("JD_PIPELINE"); os.environ.pop("JD_PIPE_DEPTH")
    for _ in range(3):
        dec.prefetch_scores(d_feats.data_ptr(), offs, 0)
    for step in range(3):
        got = dec.decode_batch_device(d_feats.data_ptr(), offs, 0)
        assert all(a.n == b.n and np.array_equal(a.label, b.label) for a, b in zip(got, want))
    dec.close()
    dec = capi.Decoder(gnet, gam, main_beam=150.0, max_streams=2)
    br = capi.Broker(dec)
    c = br.open()
    for u in range(2):
        br.init(c); br.push(c, feats[u]); h = br.finish(c)
        assert h.n == want[u].n and np.array_equal(h.label, want[u].label)
    br.close(); dec.close()
    dt = time.perf_counter() - t0
    worst = max(worst, dt)
    if dt > 2.0:
        print("round %d took %.1f s" % (k, dt))
print("%d rounds, the slowest %.2f s" % (n, worst))
