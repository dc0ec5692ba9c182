#!/usr/bin/env python
"""What would scoring ON DEMAND score?  (verdict r5 item 1.)

The reference evaluates a tied state only when a token that passed the emit threshold asks for it (WFSTDecoderLite.cpp:409-411) and
computes five frames at a time (HTKFlatModels::calcGMMOutput's block cache, HTKFlatModels.cpp:226-262); this build scores every
state of every frame with a dense kernel.  This tool takes the cells the reference algorithm really asks for - the CPU oracle with a
mark per calcGMMOutput call (jo_set_cells; the GPU path reads exactly the same cells: tests/test_gpu_cells.py) - on utterances of
configs[1] and counts, for block lengths B = 1 .. 128, the cells a demand-driven scorer with that block would COMPUTE: a block
(g, t .. t+B-1) is computed when g is asked for at t and not covered by the block before.  B = 128 is the dense kernel's own tile
(128 frames x one state): what skipping whole (state, tile) pairs nobody asks for would save.

    python tools/demand_stats.py [--utts 6] [--out profiles/r06_demand_stats.json]
"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--utts", type=int, default=6)
    ap.add_argument("--beam", type=float, default=150.0)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r06_demand_stats.json"))
    args = ap.parse_args()
    from juicer_amd import synth
    from oracle import oracle as orc
    from oracle.oracle import OracleAM, OracleDecoder, OracleNet
    orc.build(force=True)
    am, net, feats, _ = synth.config_c2(seed=0, n_utts=args.utts)
    od = OracleDecoder(OracleNet(net), OracleAM(am), main_beam=args.beam)
    L = orc.lib()
    G = am.n_gmm
    Bs = [1, 2, 3, 4, 5, 8, 16, 32, 64, 128]
    tot = {"frames": 0, "cells": 0, "read": 0, "new_per_frame": 0, "computed": {B: 0 for B in Bs}, "tiles_touched": 0, "tiles": 0}
    for x in feats:
        T = x.shape[0]
        cells = np.zeros((T, G), np.uint8)
        L.jo_set_cells(od.h, cells.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_int32(T))
        o = od.decode(x)
        L.jo_set_cells(od.h, None, C.c_int32(0))
        assert o.n > 0
        tot["frames"] += T; tot["cells"] += T * G; tot["read"] += int(cells.sum())
        # states asked for at t that were not asked for at t - 1 (what a frame adds to the working set)
        tot["new_per_frame"] += int((cells[1:] & (1 - cells[:-1])).sum() + cells[0].sum())
        for B in Bs:
            covered_until = np.full(G, -1, np.int64)                  # last frame the state's current block covers
            comp = 0
            for t in range(T):
                need = np.nonzero(cells[t])[0]
                miss = need[covered_until[need] < t]
                comp += int(np.minimum(B, T - t) * miss.shape[0])
                covered_until[miss] = t + B - 1
            tot["computed"][B] += comp
        nt = (T + 127) // 128
        touched = sum(int(cells[k * 128:(k + 1) * 128].any(axis=0).sum()) for k in range(nt))
        tot["tiles_touched"] += touched; tot["tiles"] += nt * G
    out = {"what": "cells of the likelihood table the reference algorithm asks for on configs[1] (CPU oracle with jo_set_cells, %d utterances, %d frames, "
                   "beam %g) and what a demand-driven scorer computing blocks of B frames would compute (DESIGN.md 3.7)" % (len(feats), tot["frames"], args.beam),
           "tied_states": G, "frames": tot["frames"],
           "cells_read_frac": round(tot["read"] / tot["cells"], 4),
           "states_read_per_frame": round(tot["read"] / tot["frames"], 1),
           "states_newly_asked_per_frame": round(tot["new_per_frame"] / tot["frames"], 1),
           "by_block_length": {str(B): {"cells_computed_frac": round(tot["computed"][B] / tot["cells"], 4),
                                        "computed_over_read": round(tot["computed"][B] / tot["read"], 3),
                                        "blocks_per_frame": round(tot["computed"][B] / B / tot["frames"], 1)} for B in Bs},
           "state_x_128_frame_tiles_touched_frac": round(tot["tiles_touched"] / tot["tiles"], 4)}
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1); f.write("\n")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
