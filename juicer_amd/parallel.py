"""Multi-GPU sharding of a decoding batch (one process per GPU).

Utterances are independent (the reference decodes them serially,
DecoderBatchTest.cpp:738-771), so the batch is sharded across ranks with no
data-path collective; the only exchange is ONE gather of fixed-size padded
1-best records at the end (RCCL over xGMI on GPUs, gloo in the CPU tests).
"""
from __future__ import annotations

from typing import List, Sequence

import numpy as np


def shard_range(n_utts: int, rank: int, world: int):
    """Contiguous shard [lo, hi) of utterance indices owned by `rank`."""
    base, rem = divmod(n_utts, world)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def pack_hyps(hyps: Sequence, max_words: int):
    """Fixed-size records: ints [n, max_words*2 + 2], floats [n, max_words*3 + 3]."""
    n = len(hyps)
    ints = np.zeros((n, 2 + 2 * max_words), dtype=np.int32)
    flts = np.zeros((n, 3 + 3 * max_words), dtype=np.float32)
    for i, h in enumerate(hyps):
        k = max(int(h.n), 0)
        if k > max_words:
            raise ValueError("hypothesis with %d words exceeds the gather record (%d)" % (k, max_words))
        ints[i, 0] = h.n
        ints[i, 1] = h.stats.get("n_frames", 0) if isinstance(h.stats, dict) else 0
        ints[i, 2:2 + k] = h.label
        ints[i, 2 + max_words:2 + max_words + k] = h.time
        flts[i, 0:3] = (h.tot_score, h.tot_ac, h.tot_lm)
        flts[i, 3:3 + k] = h.score
        flts[i, 3 + max_words:3 + max_words + k] = h.ac
        flts[i, 3 + 2 * max_words:3 + 2 * max_words + k] = h.lm
    return ints, flts


def unpack_hyps(ints: np.ndarray, flts: np.ndarray, max_words: int) -> List[dict]:
    out = []
    for i in range(ints.shape[0]):
        n = int(ints[i, 0])
        k = max(n, 0)
        out.append(dict(n=n, n_frames=int(ints[i, 1]),
                        label=ints[i, 2:2 + k].copy(), time=ints[i, 2 + max_words:2 + max_words + k].copy(),
                        tot_score=float(flts[i, 0]), tot_ac=float(flts[i, 1]), tot_lm=float(flts[i, 2]),
                        score=flts[i, 3:3 + k].copy(), ac=flts[i, 3 + max_words:3 + max_words + k].copy(),
                        lm=flts[i, 3 + 2 * max_words:3 + 2 * max_words + k].copy()))
    return out


def gather_hyps(hyps: Sequence, per_rank: int, max_words: int = 256, device=None) -> List[dict]:
    """all_gather the 1-best records of every rank; returns them in global
    utterance order.  Every rank contributes exactly `per_rank` records (pad
    with n=-2 records when a shard is short)."""
    import torch
    import torch.distributed as dist
    ints, flts = pack_hyps(hyps, max_words)
    if ints.shape[0] < per_rank:
        pad = per_rank - ints.shape[0]
        ints = np.concatenate([ints, np.full((pad, ints.shape[1]), 0, np.int32)])
        ints[-pad:, 0] = -2
        flts = np.concatenate([flts, np.zeros((pad, flts.shape[1]), np.float32)])
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        keep = ints[:, 0] != -2
        return unpack_hyps(ints[keep], flts[keep], max_words)
    world = dist.get_world_size()
    ti = torch.from_numpy(ints)
    tf = torch.from_numpy(flts)
    if device is not None:
        ti, tf = ti.to(device), tf.to(device)
    li = [torch.empty_like(ti) for _ in range(world)]
    lf = [torch.empty_like(tf) for _ in range(world)]
    dist.all_gather(li, ti)          # the one collective of the whole path (RCCL over xGMI on GPUs)
    dist.all_gather(lf, tf)
    gi = torch.cat(li).cpu().numpy()
    gf = torch.cat(lf).cpu().numpy()
    keep = gi[:, 0] != -2
    return unpack_hyps(gi[keep], gf[keep], max_words)
