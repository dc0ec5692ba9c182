#!/usr/bin/env python
"""One of bench.py's workloads on its own (GPU box): python tools/run_leg.py c2|north|c3|hyps|c512|clg|mixed|hypspipe|c512slot|... [passes [utterances of the c3 leg]]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from juicer_amd import synth  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "north"
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 2
n_utts = int(sys.argv[3]) if len(sys.argv) > 3 else 0
dev = torch.device("cuda:0")
if which == "north":
    a, n, f, _ = synth.config_c4(seed=0, n_utts=64, n_words=10000, n_tri_hist=100_000)
    out = bench.run_leg("north_star target (trigram-shaped)", a, n, f, 200.0, 0, dev, passes=passes, pmc_leg="north")
elif which == "northpipe":                                            # (the north-star workload through the slot pipeline: LEG_DEPTH batches ahead, LEG_SLOTS slots)
    a, n, f, _ = synth.config_c4(seed=0, n_utts=64, n_words=10000, n_tri_hist=100_000)
    out = bench.run_leg("north_star target (trigram-shaped), through the slot pipeline", a, n, f, 200.0, 0, dev, passes=passes,
                        pipe=(int(os.environ.get("LEG_DEPTH", "9")), int(os.environ.get("LEG_SLOTS", "448"))),
                        caps=[int(x) for x in os.environ.get("LEG_CAPS", "1048576,4194304,1048576").split(",")])
elif which == "northslot":                                            # (the north-star workload, LEG_UTTS utterances in ONE call on as many streams: one launch of
    nu = int(os.environ.get("LEG_UTTS", "320"))                       # the slot kernel per pass, a workgroup per utterance - jd_slot.h: k_slot_batch)
    a, n, f, _ = synth.config_c4(seed=0, n_utts=nu, n_words=10000, n_tri_hist=100_000)
    out = bench.run_leg("north_star target (trigram-shaped), %d utterances on %d streams" % (nu, nu), a, n, f, 200.0, 0, dev, passes=passes, max_streams=nu,
                        caps=[int(x) for x in os.environ.get("LEG_CAPS", "1048576,2097152,1048576").split(",")])
elif which == "clg":
    out = bench.compose_leg(0, dev)
elif which == "c3":
    a, n, f, _ = synth.config_c4(seed=0, n_utts=n_utts or 8)
    out = bench.run_leg("configs[3]", a, n, f, 300.0, 0, dev, passes=passes, pmc_leg="c3" if not n_utts or n_utts == 8 else None)
elif which == "c2pipe":                                               # (as the headline runs it: through the resident kernel, six batches ahead - not
    a, n, f, _ = synth.config_c2(seed=0, n_utts=64)                   # under --pmc: the profiler runs kernels one after the other, and this one waits for others)
    out = bench.run_leg("configs[1]", a, n, f, 150.0, 0, dev, passes=passes, pipe=(6, 160))
elif which == "mixed":                                                # (configs[1]'s graph with HMMs of 1-6 emitting states, through the slot pipeline)
    a, n, f, _ = synth.config_c2_mixed(seed=0, n_utts=64)
    out = bench.run_leg("configs[1]'s graph with HMMs of 1-6 emitting states", a, n, f, 150.0, 0, dev, passes=max(passes, 8), pipe=(9, 256))
elif which == "hypspipe":
    a, n, f, _ = synth.config_c2(seed=0, n_utts=64)
    out = bench.run_leg("configs[1] + histogram pruning, through the slot pipeline", a, n, f, 150.0, 6000, dev, passes=max(passes, 8), pipe=(9, 256))
elif which in ("c2", "c2two"):                                        # (... with two batches in flight, one launch per step: what the counters can see)
    a, n, f, _ = synth.config_c2(seed=0, n_utts=64)
    out = bench.run_leg("configs[1], two batches in flight", a, n, f, 150.0, 0, dev, passes=passes, pmc_leg="c2", two=True)
elif which == "c512pipe":                                             # (the 512-utterance batch through the resident kernel's slots)
    a, n, f, _ = synth.config_c2(seed=0, n_utts=512)
    out = bench.run_leg("configs[2]'s 512-utterance batch on one GPU, through the resident kernel", a, n, f, 150.0, 0, dev, passes=passes,
                        pipe=(int(os.environ.get("LEG_DEPTH", "1")), int(os.environ.get("LEG_SLOTS", "160"))))
elif which == "c512slot":                                             # (one launch of the slot kernel, a workgroup per utterance: what the counters see of jd_slot.h)
    a, n, f, _ = synth.config_c2(seed=0, n_utts=512)
    out = bench.run_leg("configs[2]'s 512-utterance batch on one GPU, 512 streams: k_slot_batch", a, n, f, 150.0, 0, dev, passes=passes, max_streams=512)
elif which == "c512slotfast":                                         # (... with the scoring option: what the dense table's cost is worth where scoring and search alternate)
    a, n, f, _ = synth.config_c2(seed=0, n_utts=512)
    out = bench.run_leg("configs[2]'s 512-utterance batch on one GPU, 512 streams: k_slot_batch, JD_SCORE_FAST", a, n, f, 150.0, 0, dev, passes=passes, max_streams=512,
                        scoring="fast")
elif which == "c2pipefast":
    a, n, f, _ = synth.config_c2(seed=0, n_utts=64)
    out = bench.run_leg("configs[1], JD_SCORE_FAST", a, n, f, 150.0, 0, dev, passes=max(passes, 24), pipe=(9, 256), scoring="fast")
elif which == "c512":
    a, n, f, _ = synth.config_c2(seed=0, n_utts=512)
    out = bench.run_leg("configs[2]'s 512-utterance batch on one GPU, 128 streams", a, n, f, 150.0, 0, dev, passes=passes, pmc_leg="c512", max_streams=128)
else:
    a, n, f, _ = synth.config_c2(seed=0, n_utts=64)
    out = bench.run_leg("configs[1] + histogram pruning", a, n, f, 150.0, 6000, dev, passes=passes, pmc_leg="hyps", two=True)
print(json.dumps(out))
