"""Multi-GPU sharding of a decoding batch (one process per GPU).

Utterances are independent (the reference decodes them serially,
DecoderBatchTest.cpp:738-771), so the batch is sharded across ranks with no
data-path collective - contiguously (weak scaling: every rank generates its own
utterances) or balanced by length (shard_lpt: a fixed batch over N ranks); the
only exchange is ONE gather of padded 1-best records at the end (RCCL over xGMI
on GPUs, gloo in the CPU tests).
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


def shard_lpt(n_frames: Sequence[int], world: int) -> List[List[int]]:
    """Length-balanced shards: utterances by decreasing length, each to the rank with the fewest
    frames so far (longest-processing-time-first; ties go to the rank with fewer utterances, then the lower rank).
    Returns one list of utterance indices per rank, the same on every rank that calls it with the same lengths
    (csrc/jd_multi.cpp deals a node's devices the same way).  A step lasts as long as its slowest rank:
    contiguous shards differ by whatever their longest utterances differ."""
    order = sorted(range(len(n_frames)), key=lambda u: (-int(n_frames[u]), u))
    load = [0] * world
    out: List[List[int]] = [[] for _ in range(world)]
    for u in order:
        r = min(range(world), key=lambda k: (load[k], len(out[k]), k))
        out[r].append(u)
        load[r] += max(int(n_frames[u]), 0) + 1
    return out


def pack_hyps(hyps: Sequence, max_words: int, truncate: bool = False):
    """Fixed-size records: ints [n, max_words*2 + 2], floats [n, max_words*3 + 3].  truncate: a longer hypothesis
    keeps its word count in the record and loses the words beyond max_words (gather_hyps then asks again)."""
    n = len(hyps)
    ints = np.zeros((n, 2 + 2 * max_words), dtype=np.int32)
    flts = np.zeros((n, 3 + 3 * max_words), dtype=np.float32)
    for i, h in enumerate(hyps):
        k = max(int(h.n), 0)
        if k > max_words:
            if not truncate:
                raise ValueError("hypothesis with %d words exceeds the gather record (%d)" % (k, max_words))
            k = max_words
        ints[i, 0] = h.n
        ints[i, 1] = h.stats.get("n_frames", 0) if isinstance(h.stats, dict) else 0
        ints[i, 2:2 + k] = h.label[:k]
        ints[i, 2 + max_words:2 + max_words + k] = h.time[:k]
        flts[i, 0:3] = (h.tot_score, h.tot_ac, h.tot_lm)
        flts[i, 3:3 + k] = h.score[:k]
        flts[i, 3 + max_words:3 + max_words + k] = h.ac[:k]
        flts[i, 3 + 2 * max_words:3 + 2 * max_words + k] = h.lm[:k]
    return ints, flts


def _unpack_one(ints: np.ndarray, flts: np.ndarray, max_words: int, i: int) -> dict:
    n = int(ints[i, 0])
    k = max(n, 0)
    return dict(n=n, n_frames=int(ints[i, 1]),
                label=ints[i, 2:2 + k].copy(), time=ints[i, 2 + max_words:2 + max_words + k].copy(),
                tot_score=float(flts[i, 0]), tot_ac=float(flts[i, 1]), tot_lm=float(flts[i, 2]),
                score=flts[i, 3:3 + k].copy(), ac=flts[i, 3 + max_words:3 + max_words + k].copy(),
                lm=flts[i, 3 + 2 * max_words:3 + 2 * max_words + k].copy())


def unpack_hyps(ints: np.ndarray, flts: np.ndarray, max_words: int) -> List[dict]:
    return [_unpack_one(ints, flts, max_words, i) for i in range(ints.shape[0])]


class GatheredHyps(Sequence):
    """The gathered 1-best records of all ranks, in utterance order: a sequence of dicts {n, n_frames, label, time,
    score, ac, lm, tot_*} that are made when they are asked for - every rank holds every record after the gather,
    and turning 512 of them into Python objects takes longer than the collective (a step of the 8-GPU bench would
    spend a tenth of its time on it)."""

    def __init__(self, ints: np.ndarray, flts: np.ndarray, max_words: int, rows: np.ndarray):
        self._i, self._f, self._mw, self._rows = ints, flts, max_words, rows

    def __len__(self):
        return int(self._rows.shape[0])

    def __getitem__(self, k):
        if isinstance(k, slice):
            return [self[j] for j in range(*k.indices(len(self)))]
        if k < 0:
            k += len(self)
        if not 0 <= k < len(self):
            raise IndexError(k)
        return _unpack_one(self._i, self._f, self._mw, int(self._rows[k]))


def _padded_records(hyps, per_rank, max_words, index):
    """One step's records of this rank: exactly per_rank of them (short shards are padded with n = -2 records)."""
    if len(hyps) > per_rank:                            # (a longer shard would shift every later step's rows on this rank)
        raise ValueError("a step holds %d records, the gather was sized for %d per rank" % (len(hyps), per_rank))
    if index is not None and len(index) != len(hyps):
        raise ValueError("%d utterance indices for %d hypotheses" % (len(index), len(hyps)))
    ints, flts = pack_hyps(hyps, max_words, truncate=True)
    if index is not None:                               # (the index rides in a column of its own, behind the record)
        ints = np.concatenate([ints, np.asarray(index, np.int32).reshape(-1, 1)], axis=1)
    if ints.shape[0] < per_rank:
        pad = per_rank - ints.shape[0]
        ints = np.concatenate([ints, np.full((pad, ints.shape[1]), 0, np.int32)])
        ints[-pad:, 0] = -2
        flts = np.concatenate([flts, np.zeros((pad, flts.shape[1]), np.float32)])
    return ints, flts


def _all_gather_records(ints, flts, device):
    """ONE all_gather of this rank's records (int words, the float words riding behind them); rank after rank."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return ints, flts
    world = dist.get_world_size()
    ti = torch.from_numpy(ints)
    tf = torch.from_numpy(flts).view(torch.int32)
    t = torch.cat([ti, tf], dim=1).contiguous()
    if device is not None:
        t = t.to(device)
    lt = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(lt, t)                                     # RCCL over xGMI on GPUs
    g = torch.cat(lt).cpu()
    gi = g[:, :ints.shape[1]].numpy()
    gf = g[:, ints.shape[1]:].contiguous().view(torch.float32).numpy()
    return gi, gf


def _ordered_rows(gi, rows, indexed):
    rows = rows[gi[rows, 0] != -2]
    if indexed:                                         # global utterance order: record of utterance u at rows[u]
        u = gi[rows, -1].astype(np.int64)
        if not np.array_equal(np.sort(u), np.arange(u.shape[0])):
            raise ValueError("gathered records do not cover utterances 0..%d" % (u.shape[0] - 1))
        ordered = np.empty_like(rows)
        ordered[u] = rows
        rows = ordered
    return rows


def gather_hyps(hyps: Sequence, per_rank: int, max_words: int = 256, device=None, index: Sequence[int] = None) -> "GatheredHyps":
    """all_gather the 1-best records of every rank - the ONE collective of a step.  Every rank contributes
    exactly `per_rank` records (pad with n=-2 records when a shard is short).  A record holds max_words
    words; a hypothesis that is longer travels truncated with its true word count, every rank sees that in
    the gathered records and the gather is repeated once with records of that length (no hypothesis is ever
    refused; an ordinary step is one collective).  index: the global utterance index of each of this rank's
    hypotheses (shard_lpt) - the result is then in global utterance order; without it the result is rank
    after rank (= global order for contiguous shards)."""
    return gather_hyps_steps([(hyps, index)], per_rank, max_words, device)[0]


def gather_hyps_steps(steps: Sequence, per_rank: int, max_words: int = 256, device=None) -> List["GatheredHyps"]:
    """The 1-best records of SEVERAL steps in ONE all_gather (fewer, larger collectives): steps = [(hyps, index or
    None), ...], every rank with the same number of steps and `per_rank` records per step.  What a rank needs of a
    step right away it has - its own hypotheses; what the others need of it travels when the caller says so: once
    per timed region in bench.py, behind jd_dec_quiesce, because a collective's kernels must not be queued on a
    device whose search kernel STAYS (HIP maps streams onto a few hardware queues).  Returns one
    GatheredHyps per step, as gather_hyps would have."""
    max_words = max(1, int(max_words))
    S = len(steps)
    if S == 0:
        return []
    indexed = steps[0][1] is not None
    if any((ix is not None) != indexed for _, ix in steps):        # (one decision for the whole exchange: the index column is there or not)
        raise ValueError("gather_hyps_steps: steps with and without utterance indices in one exchange")
    for attempt in range(2):
        packed = [_padded_records(h, per_rank, max_words, ix) for h, ix in steps]
        ints = np.concatenate([p[0] for p in packed]); flts = np.concatenate([p[1] for p in packed])
        gi, gf = _all_gather_records(ints, flts, device)
        longest = int(gi[:, 0].max()) if gi.shape[0] else 0
        if longest <= max_words:
            break
        max_words = longest                                        # (the same decision on every rank: all see the same records)
    world = gi.shape[0] // max(1, S * per_rank)
    if gi.shape[0] != max(1, world) * S * per_rank:
        raise ValueError("gathered %d records, expected a multiple of %d steps x %d per rank" % (gi.shape[0], S, per_rank))
    out = []
    for k in range(S):                                             # step k: rank after rank, per_rank records each
        rows = np.concatenate([np.arange(per_rank, dtype=np.int64) + (r * S + k) * per_rank for r in range(max(1, world))])
        out.append(GatheredHyps(gi, gf, max_words, _ordered_rows(gi, rows, indexed)))
    return out
