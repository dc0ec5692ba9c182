"""Regenerates tests/golden/refbase_random_golden.json: outputs of the REFERENCE's OWN WFSTDecoderLite (tools/refbase, the stand-in
build of make_refbase_golden.py) on the random graphs of ARBITRARY shape of tests/random_topology.py - any in- and out-degree,
parallel arcs, self loops, epsilon and tee arcs anywhere, labels and final weights anywhere, arcs into the initial state; 1-3
utterances and one pruning set per graph (tests/test_gpu_random_topology.py: _case).  Build container only; what is committed is
data: the inputs are regenerated from the seeds (digests stored), the expected outputs are the reference driver's.

    python tests/golden/make_refbase_random_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "tools", "refbase"), HERE, os.path.dirname(HERE)):
    sys.path.insert(0, p)
import refdiff                                                   # noqa: E402
from make_golden import input_digest                            # noqa: E402
import test_gpu_random_topology as trt                           # noqa: E402

SEEDS = list(range(7000, 7024))
f32hex = lambda v: np.float32(v).tobytes().hex()


def main():
    assert refdiff.available(), "the build container only"
    out = {"_note": "outputs of the reference's own WFSTDecoderLite on random graphs of arbitrary shape (stand-in build): see make_refbase_random_golden.py",
           "cases": []}
    for seed in SEEDS:
        am, net, feats, kw = trt._case(seed)
        rows, (rc, err, _) = refdiff.run_reference(am, net, feats, kw, loader=("fsm" if seed % 2 else "jwnt"))
        assert rc == 0 and len(rows) == len(feats), (seed, kw, rc, err)
        utts = []
        for r in rows:
            u = {"n": r["n"], "stats": {k: int(v) for k, v in r["stats"].items()}}
            if r["n"] > 0:
                u.update(label=r["label"], time=r["time"], score_hex=[f32hex(v) for v in r["score"]], ac_hex=[f32hex(v) for v in r["ac"]],
                         lm_hex=[f32hex(v) for v in r["lm"]], tot=[f32hex(v) for v in r["tot"]])
            utts.append(u)
        out["cases"].append({"seed": seed, "input_sha256": input_digest(am, net, feats), "n_states": int(net.n_states), "n_arcs": int(net.n_arcs),
                             "beams": kw, "utts": utts})
    path = os.path.join(HERE, "refbase_random_golden.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path, os.path.getsize(path), "bytes;", sum(len(c["utts"]) for c in out["cases"]), "utterances,",
          sum(u["n"] > 0 for c in out["cases"] for u in c["utts"]), "with a hypothesis")


if __name__ == "__main__":
    main()
