"""world_size-2 gloo test of the multi-GPU layer: utterance sharding + the single gather of
padded 1-best records (RCCL on GPUs, gloo here)."""
import os
import socket

import numpy as np
import pytest


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, n_utts, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from juicer_amd import parallel, synth
    from oracle.oracle import OracleAM, OracleDecoder, OracleNet
    am, net, feats, _ = synth.config_small(n_utts=n_utts)
    od = OracleDecoder(OracleNet(net), OracleAM(am), main_beam=150.0)   # stand-in producer of hyps on CPU
    ser = lambda hs: [(h["n"], h["label"].tolist(), h["time"].tolist(), float(h["tot_score"])) for h in hs]
    # contiguous shards (weak scaling: every rank brings its own utterances)
    lo, hi = parallel.shard_range(n_utts, rank, world)
    mine = [od.decode(feats[u]) for u in range(lo, hi)]
    per_rank = (n_utts + world - 1) // world
    allh = parallel.gather_hyps(mine, per_rank, max_words=64)
    # length-balanced shards of one fixed batch (BASELINE.json configs[2]); records of 3 words, so that the longer
    # hypotheses force the gather to be repeated with longer records
    shards = parallel.shard_lpt([f.shape[0] for f in feats], world)
    mine = [od.decode(feats[u]) for u in shards[rank]]
    balanced = parallel.gather_hyps(mine, max(len(x) for x in shards), max_words=3, index=shards[rank])
    # the records of SEVERAL steps in one collective (what bench.py does with ranks: the K timed steps' records travel in one
    # all_gather behind jd_dec_quiesce): three steps - the contiguous shards, the same reversed, the balanced shards with their
    # index - and records of 3 words again, so that the one collective has to be repeated with longer records
    contig = [od.decode(feats[u]) for u in range(lo, hi)]
    steps = parallel.gather_hyps_steps([(contig, list(range(lo, hi))), (contig[::-1], list(range(lo, hi))[::-1]), (mine, shards[rank])],
                                       max(per_rank, max(len(x) for x in shards)), max_words=3)
    if rank == 0:
        q.put((ser(allh), ser(balanced), [ser(x) for x in steps]))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_covers_everything():
    from juicer_amd import parallel
    for n in (0, 1, 5, 64, 513):
        for w in (1, 2, 3, 8):
            spans = [parallel.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def test_pack_unpack_roundtrip(built):
    from juicer_amd import parallel, synth
    from oracle.oracle import OracleAM, OracleDecoder, OracleNet
    am, net, feats, _ = synth.config_toy()
    h = OracleDecoder(OracleNet(net), OracleAM(am)).decode(feats[0])
    none = OracleDecoder(OracleNet(net), OracleAM(am)).decode(feats[0][:1])
    ints, flts = parallel.pack_hyps([h, none], 16)
    back = parallel.unpack_hyps(ints, flts, 16)
    assert back[0]["n"] == h.n and back[0]["label"].tolist() == h.label.tolist()
    assert np.array_equal(back[0]["score"], h.score) and back[1]["n"] == -1
    with pytest.raises(ValueError):
        parallel.pack_hyps([h], 2)


@pytest.mark.timeout(300)
def test_gather_world_size_2(built):
    import torch.multiprocessing as mp
    from juicer_amd import synth
    from oracle.oracle import OracleAM, OracleDecoder, OracleNet
    n_utts = 5                                             # ragged: shards of 3 and 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_utts, q)) for r in range(2)]
    for p in procs:
        p.start()
    got, balanced, steps = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    am, net, feats, _ = synth.config_small(n_utts=n_utts)
    od = OracleDecoder(OracleNet(net), OracleAM(am), main_beam=150.0)
    want = [od.decode(f) for f in feats]
    assert max(w.n for w in want) > 3                                # (the balanced gather had to ask twice)
    assert len(steps) == 3
    for res in [got, balanced] + steps:                              # (every step of the multi-step gather: global utterance order)
        assert len(res) == n_utts
        for g, w in zip(res, want):
            assert g[0] == w.n and g[1] == w.label.tolist() and g[2] == w.time.tolist()
            assert g[3] == pytest.approx(w.tot_score, rel=1e-6)


def test_length_balanced_shards():
    """shard_lpt: every utterance exactly once, and the heaviest rank carries at most the lightest one's frames plus
    one utterance (the longest-processing-time-first bound)."""
    from juicer_amd import parallel
    rng = np.random.default_rng(3)
    for n, w in ((0, 2), (1, 4), (7, 3), (64, 8), (512, 8)):
        T = rng.integers(300, 1001, size=n).tolist()
        sh = parallel.shard_lpt(T, w)
        assert sorted(u for s in sh for u in s) == list(range(n)) and len(sh) == w
        load = [sum(T[u] for u in s) for s in sh]
        if n >= w:
            assert max(load) - min(load) <= max(T)
    # contiguous shards of the same 512 utterances are far worse balanced than the dealt ones
    T = rng.integers(300, 1001, size=512).tolist()
    lpt = [sum(T[u] for u in s) for s in parallel.shard_lpt(T, 8)]
    cont = [sum(T[slice(*parallel.shard_range(512, r, 8))]) for r in range(8)]
    assert max(lpt) - min(lpt) < max(cont) - min(cont)


def test_gather_refuses_inconsistent_steps(built):
    """a step with more records than `per_rank`, or steps that disagree about carrying utterance indices, would shift
    or misattribute every later step's rows - refused before anything travels (world size 1: no process group needed)."""
    from juicer_amd import parallel, synth
    from oracle.oracle import OracleAM, OracleDecoder, OracleNet
    am, net, feats, _ = synth.config_toy()
    h = OracleDecoder(OracleNet(net), OracleAM(am)).decode(feats[0])
    with pytest.raises(ValueError, match="sized for 1 per rank"):
        parallel.gather_hyps_steps([([h, h], None)], per_rank=1)
    with pytest.raises(ValueError, match="with and without"):
        parallel.gather_hyps_steps([([h], [0]), ([h], None)], per_rank=1)
    with pytest.raises(ValueError, match="utterance indices"):
        parallel.gather_hyps_steps([([h], [0, 1])], per_rank=2)
    ok = parallel.gather_hyps_steps([([h], [0]), ([h], [0])], per_rank=2)       # (short shards are padded, not refused)
    assert [len(x) for x in ok] == [1, 1] and ok[1][0]["n"] == h.n
