#!/usr/bin/env python
"""Batches through the resident kernel utterance by utterance (JD_PIPELINE=3) against two batches in flight, at configs[1]:
python tools/pipe_ab.py [slots ...]   ->  ms per step, frames/s, identical results (GPU box)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from juicer_amd import capi, synth  # noqa: E402

slots = [int(a) for a in sys.argv[1:]] or [176]
DEPTH = int(os.environ.get("PIPE_AB_DEPTH", "6"))
STEPS = int(os.environ.get("PIPE_AB_STEPS", "12"))
dev = torch.device("cuda", 0)
am, net, feats, _ = synth.config_c2(seed=0, n_utts=64, target_arcs=1_000_000)
gnet, gam = capi.Network.from_synth(net), capi.Models.from_htk(am)
offs = np.zeros(len(feats) + 1, dtype=np.int64)
offs[1:] = np.cumsum([f.shape[0] for f in feats])
# (the pipeline tells batches apart by their feature pointer: DEPTH + 1 copies of the batch)
bufs = [torch.from_numpy(np.concatenate(feats)).to(dev) for _ in range(DEPTH + 2)]
torch.cuda.synchronize()
frames = int(offs[-1])


def bit_same(a, b):
    return (a.n == b.n and np.array_equal(a.label, b.label) and np.array_equal(a.time, b.time)
            and all(np.array_equal(np.asarray(getattr(a, k), np.float32).view(np.uint32), np.asarray(getattr(b, k), np.float32).view(np.uint32))
                    for k in ("score", "ac", "lm")))


def run(env, max_streams, ahead):
    for k in ("JD_PIPELINE", "JD_PIPE_DEPTH"):
        os.environ.pop(k, None)
    os.environ.update(env)
    dec = capi.Decoder(gnet, gam, main_beam=150.0, device=0, max_streams=max_streams)
    nb = len(bufs)
    for k in range(ahead):
        dec.prefetch_scores(bufs[k % nb].data_ptr(), offs, 0)
    hy = None
    t_steps = []
    for step in range(STEPS + 4):
        if step == 4:
            t0 = time.perf_counter()
        dec.prefetch_scores(bufs[(step + ahead) % nb].data_ptr(), offs, 0)
        hy = dec.decode_batch_device(bufs[step % nb].data_ptr(), offs, 0)
        t_steps.append(time.perf_counter())
    dt = (time.perf_counter() - t0) / STEPS
    # drain what is announced behind the last decode
    for step in range(STEPS + 4, STEPS + 4 + ahead):
        hy2 = dec.decode_batch_device(bufs[step % nb].data_ptr(), offs, 0)
    tm = dec.last_timing()
    dec.close()
    return dt, hy, hy2


base_dt, base, base2 = run({}, 128, 2)
print("two batches in flight (128 streams): %.2f ms per step = %.0f frames/s" % (base_dt * 1e3, frames / base_dt))
assert all(bit_same(a, b) for a, b in zip(base, base2))
for s in slots:
    dt, hy, hy2 = run({"JD_PIPELINE": "3", "JD_PIPE_DEPTH": str(DEPTH + 1)}, s, DEPTH)
    same = sum(bit_same(a, b) for a, b in zip(hy, base)), sum(bit_same(a, b) for a, b in zip(hy2, base))
    print("resident pipeline, %3d slots, %d batches ahead: %.2f ms per step = %.0f frames/s; identical to the other path: %d/64, %d/64"
          % (s, DEPTH, dt * 1e3, frames / dt, same[0], same[1]))
