#!/usr/bin/env python
"""Where a configs[1] step's host time goes (scored ahead): announce / decode call / inside the call."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from juicer_amd import capi, synth  # noqa: E402

am, net, feats, _ = synth.config_c2(seed=0, n_utts=64, target_arcs=1_000_000)
dec = capi.Decoder(capi.Network.from_synth(net), capi.Models.from_htk(am), main_beam=150.0, device=0, max_streams=64)
offs = np.zeros(65, dtype=np.int64)
offs[1:] = np.cumsum([f.shape[0] for f in feats])
d_feats = torch.from_numpy(np.concatenate(feats)).to("cuda:0")
torch.cuda.synchronize()
for _ in range(3):
    dec.prefetch_scores(d_feats.data_ptr(), offs, 0)
    dec.decode_batch_device(d_feats.data_ptr(), offs, 0)
acc = np.zeros(5)
N = 10
for _ in range(N):
    t0 = time.perf_counter()
    dec.prefetch_scores(d_feats.data_ptr(), offs, 0)
    t1 = time.perf_counter()
    raw = dec.decode_batch_device(d_feats.data_ptr(), offs, 0, raw=True)
    t2 = time.perf_counter()
    hyps = [capi._hyp_from_c(raw[i]) for i in range(64)]
    t3 = time.perf_counter()
    tm = dec.last_timing()
    acc += [t1 - t0, t2 - t1, t3 - t2, tm["total_ms"] * 1e-3, tm["search_ms"] * 1e-3]
acc *= 1e3 / N
print("announce %.3f ms, decode call %.3f ms (inside: wave %.3f ms of which search kernels %.3f ms), python hyps %.3f ms" % (acc[0], acc[1], acc[3], acc[4], acc[2]))
