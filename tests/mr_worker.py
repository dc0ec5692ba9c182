"""Worker of tests/test_gpu_multirank.py: one rank of a world-size-N run on ONE GPU (gloo).
Every rank decodes its contiguous shard of the batch with the HIP path, the 1-best records are
exchanged with the one all_gather of the path, and rank 0 checks the gathered records against a
single-rank decode of the whole batch."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch.distributed as dist  # noqa: E402
from juicer_amd import capi, parallel, synth  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    am, net, feats, _ = synth.config_small(n_utts=7)            # ragged: 4 + 3 on two ranks
    lo, hi = parallel.shard_range(len(feats), rank, world)
    per_rank = (len(feats) + world - 1) // world
    kw = dict(main_beam=150.0, max_hyps=200)
    gnet, gam = capi.Network.from_synth(net), capi.Models.from_htk(am)
    mine = capi.Decoder(gnet, gam, device=0, max_streams=per_rank, **kw).decode_batch(feats[lo:hi])
    allh = parallel.gather_hyps(mine, per_rank)                   # the one collective
    ok = True
    if rank == 0:
        ref = capi.Decoder(gnet, gam, device=0, max_streams=len(feats), **kw).decode_batch(feats)
        ok = len(allh) == len(feats)
        for g, r in zip(allh, ref):
            ok = ok and g["n"] == r.n and r.n > 0 and np.array_equal(g["label"], r.label) and np.array_equal(g["time"], r.time)
            ok = ok and np.array_equal(g["score"].view(np.uint32), r.score.view(np.uint32))
            ok = ok and np.float32(g["tot_ac"]).view(np.uint32) == np.float32(r.tot_ac).view(np.uint32)
    # the same batch dealt by length (BASELINE.json configs[2]'s sharding), short gather records (a longer hypothesis
    # makes the gather ask again)
    shards = parallel.shard_lpt([f.shape[0] for f in feats], world)
    mine = capi.Decoder(gnet, gam, device=0, max_streams=max(1, len(shards[rank])), **kw).decode_batch([feats[u] for u in shards[rank]])
    bal = parallel.gather_hyps(mine, max(len(x) for x in shards), max_words=2, index=shards[rank])
    if rank == 0:
        ok = ok and len(bal) == len(feats)
        for g, r in zip(bal, ref):
            ok = ok and g["n"] == r.n and np.array_equal(g["label"], r.label) and np.array_equal(g["time"], r.time)
            ok = ok and np.array_equal(g["score"].view(np.uint32), r.score.view(np.uint32))
        print("multirank: %d records gathered from %d ranks, identical to the single-rank decode: %s" % (len(allh), world, ok))
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
