"""Development: one random-topology batch set through the slot pipeline (tests/test_gpu_random_topology.py's pipeline case) with knobs:
python tools/rt_pipe_diag.py seed streams n_utts max_parts [chunk]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
seed, streams, n_utts, max_parts = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
if len(sys.argv) > 5: os.environ["JD_DEV"] = "1"; os.environ["JD_PIPE_CHUNK"] = sys.argv[5]
import torch
from juicer_amd import capi
import random_topology as rt
import test_gpu_random_topology as trt
am, net, _, kw, lm, pen = trt._big_case(seed) if seed >= 8000 else (trt._case(seed)[:2] + (None, trt._case(seed)[3], 1.0, 0.0))
kw = {k: v for k, v in kw.items() if k != "max_hyps"}
fs = [np.concatenate([rt.random_walk_features(seed + 10 * u + j, net, am, n_arcs=10 + 3 * u) for j in range(1 + u % max_parts)]) for u in range(n_utts)]
dev = torch.device("cuda", 0)
offs = np.zeros(len(fs) + 1, dtype=np.int64); offs[1:] = np.cumsum([x.shape[0] for x in fs])
buf = torch.from_numpy(np.concatenate(fs)).to(dev)
gd = capi.Decoder(capi.Network.from_synth(net, lm, pen), capi.Models.from_htk(am), max_streams=streams, **kw)
ref = None
if not os.environ.get("NOPLAIN"):
    ref = gd.decode_batch(fs)
    print("plain decode ok:", [h.n for h in ref], "frames", [x.shape[0] for x in fs], flush=True)
gd.set_pipeline(capi.FLOW_RESIDENT, 3)
NB = int(os.environ.get("NB", "1"))
sets = [(fs, buf, offs, ref)]
for b in range(1, NB):                                                 # further batches: other walks on the same graph
    f2 = [np.concatenate([rt.random_walk_features(seed + 100 * b + 10 * u + j, net, am, n_arcs=10 + 3 * u) for j in range(1 + u % max_parts)]) for u in range(n_utts)]
    o2 = np.zeros(len(f2) + 1, dtype=np.int64); o2[1:] = np.cumsum([x.shape[0] for x in f2])
    gd.set_pipeline(capi.FLOW_DEFAULT if hasattr(capi, "FLOW_DEFAULT") else 0, 0) if False else None
    sets.append((f2, torch.from_numpy(np.concatenate(f2)).to(dev), o2, None))
t0 = time.time()
try:
    order = list(range(NB)) + list(range(NB - 1, -1, -1)) if NB > 1 else [0]
    ahead = min(2, len(order))
    for n in order[:ahead]: gd.prefetch_scores(sets[n][1].data_ptr(), sets[n][2], 0)
    for i, n in enumerate(order):
        if i + ahead < len(order): gd.prefetch_scores(sets[order[i + ahead]][1].data_ptr(), sets[order[i + ahead]][2], 0)
        t1 = time.time()
        gs = gd.decode_batch_device(sets[n][1].data_ptr(), sets[n][2], 0)
        r = sets[n][3]
        same = None if r is None else all(a.n == b.n and a.label.tolist() == b.label.tolist() and np.array_equal(a.score, b.score) for a, b in zip(gs, r))
        if r is None: sets[n] = sets[n][:3] + (gs,)
        print("step %d batch %d back in %.2f s, same as before: %s" % (i, n, time.time() - t1, same), flush=True)
    print("pipeline ok in %.2f s" % (time.time() - t0), gd.pipeline_stats(), flush=True)
except Exception as e:
    print("pipeline FAILED after %.1f s:" % (time.time() - t0), e, gd.pipeline_stats(), flush=True)
os._exit(0)
