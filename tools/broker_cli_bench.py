#!/usr/bin/env python
"""The C++ side of the seam at configs[1]: jd_batch_test -threads N (N harness threads, the reference's per-utterance loop each, one
GpuWFSTPooledDecoder per thread over one GpuDecoderPool) - frames/s per thread count, without Python between the callers and the
broker (GPU box):  python tools/broker_cli_bench.py [threads ...]"""
import os
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from juicer_amd import build as jbuild, io as jio, synth  # noqa: E402

counts = [int(a) for a in sys.argv[1:]] or [16, 32, 64]
jbuild.build()
am, net, feats, _ = synth.config_c2(n_utts=64)
with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
    jio.write_fsm(os.path.join(tmp, "g.fsm"), net)
    jio.write_jdam(os.path.join(tmp, "m.jdam"), am)
    for u, x in enumerate(feats):
        jio.write_jdf(os.path.join(tmp, "u%d.jdf" % u), x)
    for n in counts:
        # thread t takes list entries t, t + n, ...: every thread decodes the whole list, each from another starting point
        with open(os.path.join(tmp, "list.txt"), "w") as f:
            for k in range(64):
                for t in range(n):
                    f.write("%s\n" % os.path.join(tmp, "u%d.jdf" % (((t * 64) // n + k) % 64)))
        for env in ({}, {"JD_BROKER_RESIDENT": "0"}):
            r = subprocess.run([jbuild.BATCH_TEST, "-fsmFName", os.path.join(tmp, "g.fsm"), "-modelsFName", os.path.join(tmp, "m.jdam"),
                                "-inputFName", os.path.join(tmp, "list.txt"), "-mainBeam", "150", "-threads", str(n), "-outputFormat", "ref"],
                               capture_output=True, text=True, env=dict(os.environ, **env))
            line = [ln for ln in r.stderr.splitlines() if "harness threads" in ln]
            print("%s: %s" % ("ticks" if env else "resident kernel", line[-1] if line else ("rc %d: %s" % (r.returncode, r.stderr[-300:]))))
