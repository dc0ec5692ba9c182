#!/usr/bin/env python
"""Do kernels on OTHER streams start beside the batch pipeline's resident kernel?  (HIP maps streams onto a few hardware queues;
a kernel queued behind one that never leaves would wait for it.)  Small kernels on many fresh torch streams while batches
are announced; the time each takes to come back."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from juicer_amd import capi, synth  # noqa: E402

dev = torch.device("cuda", 0)
am, net, feats, _ = synth.config_c2(seed=0, n_utts=64)
gnet, gam = capi.Network.from_synth(net), capi.Models.from_htk(am)
offs = np.zeros(len(feats) + 1, dtype=np.int64)
offs[1:] = np.cumsum([f.shape[0] for f in feats])
d_feats = torch.from_numpy(np.concatenate(feats)).to(dev)
torch.cuda.synchronize()
os.environ["JD_PIPELINE"] = "3"; os.environ["JD_PIPE_DEPTH"] = "7"
dec = capi.Decoder(gnet, gam, main_beam=150.0, device=0, max_streams=160)
for _ in range(6):
    dec.prefetch_scores(d_feats.data_ptr(), offs, 0)
dec.decode_batch_device(d_feats.data_ptr(), offs, 0)                  # the pipeline is running, five batches queued
x = torch.ones(1 << 20, device=dev)
worst = 0.0
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 24):
    s = torch.cuda.Stream(device=dev, priority=-1 if k % 2 else 0)
    t0 = time.perf_counter()
    with torch.cuda.stream(s):
        y = (x * 2.0).sum()
    ev = torch.cuda.Event(); ev.record(s)
    while not ev.query():
        if time.perf_counter() - t0 > 8.0:
            break
        time.sleep(0.0005)
    dt = time.perf_counter() - t0
    worst = max(worst, dt)
    print("stream %2d (priority %d): %.1f ms%s" % (k, -1 if k % 2 else 0, dt * 1e3, "  <-- waited for the resident kernel" if dt > 1.0 else ""))
    dec.prefetch_scores(d_feats.data_ptr(), offs, 0)
    dec.decode_batch_device(d_feats.data_ptr(), offs, 0)              # (keeps the kernel busy)
print("worst: %.1f ms" % (worst * 1e3))
dec.prefetch_scores(0, None)
dec.close()
