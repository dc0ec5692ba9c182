#!/usr/bin/env python
"""bench.py - frames/sec decoded on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path (dense GMM scoring + token-passing search) over one batch of synthetic utterances per
GPU: BASELINE.json configs[1] (~1M-arc composed WFST, 3000 tied states x 16 mixtures, 64 utterances per GPU, mainBeam 150).
Features are resident in HBM before the timed region.

How the batches share the chip (`jd_dec_set_pipeline`): by default through the RESIDENT SLOT PIPELINE - a search
kernel that stays on the device (csrc/jd_slot.h: 256 one-workgroup slots, ONE per CU - eight waves at 128 VGPRs, half a CU), the
scoring kernel's workgroups on the other half of the same CUs, a likelihood table per announced batch; announcements run nine
batches ahead (`jd_dec_prefetch_scores`), a slot takes the next queued utterance the moment its own is through, and every step
hands back ITS batch's 64 results, decoded in full.  The announced
batch is the same synthetic batch again, scored from its features every time: K timed steps hold K scorings and K batches' worth of
search.  `value` = the stream-frames the slots REALLY advanced between the two brackets (`jd_dec_pipeline_stats`) / the bracketed
time - a batch handed back inside the region was partly searched before it, batches announced inside it are partly searched behind
it.  What ONE batch costs a caller that does not announce that far ahead is printed beside it (`single_batch`: nothing announced /
one batch announced), measured behind the timed region.  --pipeline-depth 0: two batches in flight, one k_search launch per step.

N > 1: one rank per GPU, utterances sharded across ranks, no data-path collective, every rank on the same path as N = 1.  A
collective's kernels must not be queued on a device whose search kernel stays, so the 1-best records of the K timed steps travel in
ONE RCCL all_gather at the end of the timed region (inside it), behind jd_dec_quiesce; a rank has its own results the moment its
step returns.  --gather-every 1: one all_gather per step, two batches in flight instead of the pipeline.  `python bench.py --gpus N`
spawns its own ranks (torch.distributed.run) when it was not started by torchrun; under torchrun it reads RANK / LOCAL_RANK /
WORLD_SIZE.  Scaling is weak by default (64 utterances per GPU); `--total-utts 512` is BASELINE.json configs[2]: ONE fixed batch
dealt over the ranks by length (strong scaling).

The timed region is bracketed by jd_dec_quiesce (the resident kernel lets its commands run out and leaves: a device-wide
synchronisation waits for every kernel), barrier and torch.cuda.synchronize() on both sides; the kernel comes back with the first
timed step, inside the brackets.

After the timed region rank 0 of a 1-GPU run also times the other single-GPU workloads of BASELINE.json and reports them under
"legs" (each with its own roofline and a CPU-oracle sample): configs[1] + maxHyps 6000, configs[1] with two batches in flight,
configs[2]'s 512-utterance batch on one GPU (one launch of the slot kernel per pass), configs[1]'s graph with HMMs of 1-6 emitting
states, the north_star target (14 M-arc graph, beam 200), configs[3] (~48M-arc trigram-shaped graph, beam 300), configs[4] (C.L and G
composed on the device).  --no-extra-legs skips them.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def search_bytes(st, max_n):
    """Algorithmic bytes of the search for the work in `st` (batch totals of jd_stats): SURVEY.md
    8(d) / DESIGN.md 4.  Token read + write (16-B tokens) and arc->hmm lookup per instance,
    likelihood gather per emitting hypothesis, exit token + CSR bounds per end hypothesis,
    arc record + hook + entry-token read-modify-write per visited arc, one Path record per word end."""
    return ((32.0 * max_n + 4.0) * st["tot_insts_in"] + 4.0 * st["tot_proc_emit_hyps"]
            + 24.0 * st["tot_proc_end_hyps"] + 52.0 * st["tot_arcs_visited"] + 20.0 * st["tot_paths"])


def roofline_of(st, max_n, tm, traffic=None, design=None):
    """Roofline of a search launch (k_search unless the caller names another kernel): achieved = algorithmic
    bytes per launch / average launch duration (HIP events around each launch on the decoder's
    search stream, jd_dec_last_timing).  frac = achieved / peak prices the launch by SURVEY.md 8(d)'s
    algorithmic bytes; frac_measured by the HBM bytes the PMC passes counted (`traffic`), when there are any
    for the code that is running."""
    launches = max(1, tm["search_launches"])
    per_launch = search_bytes(st, max_n) / launches
    avg_us = 1e3 * tm["search_ms"] / launches
    achieved = per_launch / (avg_us * 1e-6) / 1e9 if avg_us > 0 else 0.0
    measured = traffic / (avg_us * 1e-6) / 1e9 if (traffic and avg_us > 0) else None
    out = {"bound": "hbm", "kernel": "k_search", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS,
           "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic,
           "frac_measured": round(measured / HBM_PEAK_GBS, 6) if measured else None,
           "algorithmic_bytes_per_launch": round(per_launch, 1), "avg_launch_us": round(avg_us, 3),
           "launches_per_step": launches, "workgroups_per_stream": tm["cluster_wgs"]}
    if design is not None:                                         # the bytes the design requests, by the kernels' own counters
        dpl = design / launches
        out["design_bytes_per_launch"] = round(dpl, 1)
        out["frac_design"] = round(dpl / (avg_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 6) if avg_us > 0 else None
    return out


def design_bytes(st, max_n, G, frames, row_in_lds):
    """Bytes the DESIGN requests for the work in `st` (batch totals of jd_stats, the kernels' own counters tot_recs_read ..
    tot_closure_items): what `search_bytes` prices is the REFERENCE's work - 11.7 k instances per configs[1]
    frame at 164 B each, where the GPU keeps ~5 k records and the rest are candidates that never become one, and arcs a prefix
    walk accounts for without reading them.  Sectors as requested (16-byte halves of a 32-byte StateRec, 32-byte items), not the
    64 / 128-byte transactions the memory system makes of them - that is `traffic`.
      phase A  record read R + the source state's arrival key 8; a newly entered arc: list entry 8 + arc 16 + template 16 (32) +
               key 8; the winning item's token 16 per entry token pulled; record written R; exit token: item 32, + a bid 8 where one
               is placed (tot_bids_placed: not by the exit tokens of an arc that is alone into its state, REC_SOLE);
               likelihoods: the frame's row into LDS (G x 4, the slot kernel) or 4 per emitting hypothesis (k_search)
      phase X  item taken up: item 32 + state record XState 64 + CSR bounds 8 + the row's instance flags 32, + the bid keys 16
               where a bid was placed;
               arrival: 8 (atomic max); arc walked: record 16 + flag 1; closure item: item 32 + atomic 8 + destination key 8;
               Path record 32"""
    R = 80.0 if max_n <= 5 else 144.0
    tmpl = 16.0 if max_n <= 5 else 32.0
    a = (st["tot_recs_read"] * (R + 8.0) + st["tot_new_attached"] * (8.0 + 16.0 + tmpl + 8.0) + st["tot_entry_items"] * 16.0
         + st["tot_recs_written"] * R + st["tot_active_end_hyps"] * 32.0 + st["tot_bids_placed"] * 8.0
         + (frames * G * 4.0 if row_in_lds else st["tot_proc_emit_hyps"] * 4.0))
    x = (st["tot_items_expanded"] * (32.0 + 64.0 + 8.0 + 32.0) + st["tot_bids_placed"] * 16.0 + st["tot_proc_end_hyps"] * 8.0 + st["tot_arcs_walked"] * 17.0
         + st["tot_closure_items"] * 48.0 + st["tot_paths"] * 32.0)
    return a + x


def reference_cpu():
    """BASELINE.md B1 / B2 as quoted constants: the reference's own WFSTDecoderLite / WFSTDecoderLiteThreading on configs[1],
    measured in the BUILD container (tools/refbase -> profiles/cpu_reference_baseline.json; a build against stand-ins: a
    timing and differential aid, not a reference build).  The GPU box has no /root/reference: nothing is run here."""
    try:
        r = json.load(open(os.path.join(ROOT, "profiles", "cpu_reference_baseline.json")))
        b1, b2 = r.get("WFSTDecoderLite", {}), r.get("WFSTDecoderLiteThreading", {})
        d = r.get("differential", {})
        return {"B1_WFSTDecoderLite_fps": b1.get("frames_per_s"), "B1_cores": 1,
                "B2_WFSTDecoderLiteThreading_fps": b2.get("frames_per_s"), "B2_cores": 2,
                "B1_identical_to_oracle": "%s/%s" % (b1.get("identical_to_oracle"), b1.get("utterances")),
                "B2_identical_to_oracle": "%s/%s" % (b2.get("identical_to_oracle"), b2.get("utterances")),
                "differential_cases_identical": "%s/%s" % (d.get("cases_identical"), d.get("cases_total")) if d else None,
                "oracle_port_same_host_fps": r.get("oracle_port", {}).get("frames_per_s"),
                "host": "%s, %s cores (build container)" % (r.get("host", {}).get("model"), r.get("host", {}).get("cores")),
                "source": "profiles/cpu_reference_baseline.json (tools/refbase; stand-in build: pins nothing)"}
    except Exception as e:
        return {"error": repr(e)}


def word_errors(hyp, ref):
    """Levenshtein distance between two label sequences (substitutions + deletions + insertions, unit costs)."""
    hyp, ref = list(hyp), list(ref)
    prev = list(range(len(ref) + 1))
    for i, h in enumerate(hyp, 1):
        cur = [i] + [0] * len(ref)
        for j, r in enumerate(ref, 1):
            cur[j] = min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (h != r))
        prev = cur
    return prev[-1]


def wer_vs_oracle(net, am, feats, hyps, beam, max_hyps, first_done=None, workers=None):
    """BASELINE.json's metric names "1-best WER vs ref": the GPU's 1-best of EVERY utterance of the step against the CPU
    oracle's (the reference algorithm restated; oracle/ is the checker here, outside the timed region): word errors / the
    oracle's words, and how many hypotheses are identical in words, times and - bit for bit - scores.  One oracle decoder per
    host thread (the C library keeps its state in the decoder; ctypes releases the GIL)."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle.oracle import OracleAM, OracleDecoder, OracleNet
    onet, oam = OracleNet(net), OracleAM(am)
    n = len(feats)
    workers = max(1, min(workers or (os.cpu_count() or 1), 32, n))
    t0 = time.perf_counter()

    def chunk(k):
        od = OracleDecoder(onet, oam, main_beam=beam, max_hyps=max_hyps)
        return [(u, od.decode(feats[u])) for u in range(k, n, workers)]
    with ThreadPoolExecutor(workers) as ex:
        res = dict(x for part in ex.map(chunk, range(workers)) for x in part)
    errs = words = same = bits = found = 0
    f32 = lambda a: np.asarray(a, np.float32).view(np.uint32)
    for u in range(n):
        o, g = res[u], hyps[u]
        ow = list(o.label[::-1]) if o.n > 0 else []                # (chain order is newest first)
        gw = list(g.label[::-1]) if g.n > 0 else []
        errs += word_errors(gw, ow); words += len(ow); found += int(o.n > 0)
        eq = g.n == o.n and np.array_equal(g.label, o.label) and np.array_equal(g.time, o.time)
        same += int(eq)
        bits += int(eq and (o.n <= 0 or (np.array_equal(f32(g.score), f32(o.score)) and np.array_equal(f32(g.ac), f32(o.ac))
                                        and np.array_equal(f32(g.lm), f32(o.lm)))))
    return {"wer": round(errs / max(words, 1), 6), "word_errors": int(errs), "ref_words": int(words), "utterances": n,
            "identical_1best": same, "identical_scores_bitwise": bits, "oracle_hyps_found": found,
            "oracle_threads": workers, "oracle_wall_s": round(time.perf_counter() - t0, 1),
            "ref": "CPU oracle (restated reference algorithm, oracle/juicer_oracle.c), every utterance of rank 0's step"}


def kernel_source_hash():
    """What the committed PMC passes were taken on: a hash of the search kernel's sources.  `traffic` is a
    measurement of ONE build; it is reported only while the sources are the ones it was taken on."""
    import hashlib
    h = hashlib.sha256()
    for f in ("jd_search.h", "jd_lazy.h", "jd_gmm.h", "jd_gc.h", "jd_resident.h", "jd_slot.h", "jd_host_resident.h", "jd_host_scoring.h",
              "jd_host_launch.h", "jd_host_stream.h", "jd_device.hip"):
        h.update(open(os.path.join(ROOT, "juicer_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def calibrated_traffic(fetch_kib, write_kib, wide_read_bytes):
    """HBM bytes from rocprofv3's FETCH_SIZE / WRITE_SIZE (KiB), calibrated on known access counts in this
    kernel's own access shapes (tools/traffic_probe.hip, profiles/README.md): the counters tally 64 B per read
    request and 32 / 64 B per write request - exact for scattered 4 .. 32-byte gathers (one 64-byte request
    each), stores and atomics, and HALF of the bytes of wide coalesced reads (128-byte requests tallied at 64:
    the guide's factor 2 applies to those only).  wide_read_bytes = the bytes the launch reads in 1 KiB runs (its
    instance records)."""
    return (fetch_kib + write_kib) * 1024.0 + 0.5 * wide_read_bytes


PROFILE_ROUND = "r06"


def slot_traffic(frames):
    """HBM bytes of `frames` stream-frames of configs[1] through the slot kernel (csrc/jd_slot.h), from the committed PMC passes of
    k_slot_batch - the same per-stream code as the pipeline's k_slot, as a plain launch over the 512-utterance batch of configs[2]
    (tools/run_leg.py c512slot under tools/leg_pmc.sh; the counters run kernels one after the other, and k_slot waits for the
    kernels beside it): counted bytes per stream-frame x frames.  None when the passes were taken on other sources."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "%s_c512slot_traffic.json" % PROFILE_ROUND)))
        if t.get("source_hash") != kernel_source_hash():
            return None
        return round(t["k_search_hbm_bytes_per_pass"] / t["frames_per_pass"] * frames, 1)
    except Exception:
        return None


def slot_l2_requests(fps):
    """The slot kernel's other roofline: requests at the L2s.  TCC_REQ of k_slot_batch from the committed PMC pass (same proxy and
    the same source-hash rule as slot_traffic) per stream-frame x the frames/s of this run, against what the chip's memory side was
    probed to carry: 49 G scattered 64-byte loads / s that miss the L2 (tools/traffic_probe.hip, profiles/r03_traffic_probe.log;
    requests that hit an L2 are cheaper, so the fraction is an upper bound of how near that wall the kernel runs)."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "%s_c512slot_traffic.json" % PROFILE_ROUND)))
        if t.get("source_hash") != kernel_source_hash():
            return None
        pm = json.load(open(os.path.join(ROOT, "profiles", "%s_slot_pmc_summary.json" % PROFILE_ROUND)))
        k = [v for n, v in pm.items() if n.startswith("k_slot_batch")][0]
        per_sf = k["TCC_REQ_sum"]["mean"] / t["frames_per_pass"]
        return {"per_stream_frame": round(per_sf, 1), "G_per_s": round(per_sf * fps / 1e9, 2), "probe_scattered_miss_ceiling_G_per_s": 49.0,
                "frac_of_probe": round(per_sf * fps / 49e9, 4), "from": "TCC_REQ_sum of k_slot_batch (proxy, as traffic) x this run's frames/s"}
    except Exception:
        return None


def leg_traffic(leg, launches):
    """HBM bytes per k_search launch of a workload, from the committed PMC passes of that workload on its own
    (profiles/r05_<leg>_traffic.json, tools/leg_pmc.sh): the bytes of all k_search launches of one pass over the
    step's launches, like `achieved`.  None when there is no such file or when it was taken on other sources."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "%s_%s_traffic.json" % (PROFILE_ROUND, leg))))
        if t.get("source_hash") != kernel_source_hash():
            return None
        return round(t["k_search_hbm_bytes_per_pass"] / max(1, launches), 1)
    except Exception:
        return None


def run_leg(name, am, net, feats, beam, max_hyps, dev, oracle_utts=0, passes=4, gnet=None, pmc_leg=None, oracle_feats=None,
            oracle_net=None, ahead=True, max_streams=0, two=None, pipe=None, caps=None, scoring="exact"):
    """One extra workload: warm-up pass + timed passes on one GPU (value = the MEDIAN pass), its own roofline.
    gnet: a network that exists already (composed on the device); net is then only asked for its size.
    pmc_leg: the name the leg's PMC passes are filed under (leg_traffic).  oracle_utts: that many utterances are
    decoded by the CPU oracle as well - of the batch itself, or oracle_feats (short utterances on the same graph,
    decoded by the same decoder after the timed passes, where the batch's own would take the oracle minutes);
    oracle_net: the oracle's copy of the graph when `net` is not a synthetic network object.  two: two batches in
    flight, like the headline (streams for two batches, announcements two passes ahead; default: when the passes
    are given; default off: it pays where a frame is a chain of dependent steps, not where it is bytes - the heavy legs
    lose by it, measured: the 14 M-arc graph 119 k -> 55 k frames/s, configs[3] 5.3 k -> 0.8 k).  pipe = (depth, slots): the
    batches through the resident search kernel, like the headline; every announced batch is decoded (depth calls more)."""
    import torch
    from juicer_amd import capi
    U = len(feats)
    t0 = time.perf_counter()
    if two is None:
        two = False
    depth = 0
    if pipe:
        depth, max_streams, two = pipe[0], pipe[1], False
    dec = capi.Decoder(gnet if gnet is not None else capi.Network.from_synth(net), capi.Models.from_htk(am), main_beam=beam,
                       max_hyps=max_hyps, device=dev.index, max_streams=(2 * U if two else U) if not max_streams else max_streams,
                       **(dict(zip(("max_slots", "max_paths", "max_items"), caps)) if caps else {}))   # (caps: per-stream arena capacities, jd_dec_set_capacity)
    if scoring == "fast":                                              # jd_dec_set_scoring: FMA distance + fp32 logAdd (scores within 1e-4)
        dec.set_scoring(capi.SCORE_FAST)
    if depth:
        dec.set_pipeline(capi.FLOW_RESIDENT, depth + 1, max_streams)
    offs = np.zeros(U + 1, dtype=np.int64)
    offs[1:] = np.cumsum([f.shape[0] for f in feats])
    d_feats = torch.from_numpy(np.concatenate(feats)).to(dev)
    torch.cuda.synchronize()
    stream = torch.cuda.current_stream().cuda_stream
    hyps, runs = None, []
    gmm_alone = None
    if two:
        dec.prefetch_scores(d_feats.data_ptr(), offs, stream)          # (the announcements run two passes ahead)
        passes += 1                                                    # (... and the pipeline takes a pass more to fill)
    if depth:
        for _ in range(depth):
            dec.prefetch_scores(d_feats.data_ptr(), offs, stream)
        passes += depth + 2                                            # (the first batches come back in a burst)
    n_calls = passes + depth                                           # (pipe: every announced batch is decoded)
    win = None                                                         # pipe: (time, frames the slots had advanced) at the steady window's ends
    for i in range(n_calls):
        if not depth:
            torch.cuda.synchronize()                                   # (a device-wide synchronisation waits for a resident kernel)
        t1 = time.perf_counter()
        if depth and i == depth + 2:
            win = [(t1, dec.pipeline_stats()["frames_searched"])]
        if ahead and (not depth or i < passes):                        # the next pass's table is scored beside this pass's search
            dec.prefetch_scores(d_feats.data_ptr(), offs, stream)
        hyps = dec.decode_batch_device(d_feats.data_ptr(), offs, stream)
        if not depth:
            torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        tm_i = dec.last_timing()
        if not tm_i["prefetched"]:
            gmm_alone = tm_i["gmm_ms"]                                 # (the warm-up pass scores its own table, on its own)
        if depth:
            if depth + 2 <= i < passes:
                tm_i = dict(tm_i); tm_i["search_ms"] = dt * 1e3; tm_i["search_launches"] = 1
                runs.append((dt, tm_i))
            if i == passes - 1 and win:
                win.append((time.perf_counter(), dec.pipeline_stats()["frames_searched"]))
        elif i > (1 if two else 0) or passes == 1:
            runs.append((dt, tm_i))
    dec.prefetch_scores(0, None)
    runs.sort(key=lambda r: r[0])
    best, tm = runs[(len(runs) - 1) // 2]                             # the median pass (the lower one of an even number)
    frames = int(offs[-1])
    if depth:                                                          # (the pipeline hands its batches back in bursts: the MEAN of the steady passes,
        # by the frames its slots really advanced in that window - jd_dec_pipeline_stats)
        best = sum(r[0] for r in runs) / len(runs)
        if win and len(win) == 2 and win[1][1] > win[0][1]:
            best = (win[1][0] - win[0][0]) * frames / float(win[1][1] - win[0][1])
        tm = dict(tm); tm["search_ms"] = best * 1e3
    st = {k: sum(h.stats[k] for h in hyps) for k in hyps[0].stats}
    out = {"workload": "%s: %d-arc composed C.L.G, %d tied states x %d mix, %d utterances, mainBeam %g, maxHyps %d"
                       % (name, net.n_arcs, am.n_gmm, am.max_mix, U, beam, max_hyps),
           "value": round(frames / best, 1), "unit": "frames/s", "xRT": round(frames / best / 100.0, 2),
           "frames_per_step": frames, "ms_per_step": round(best * 1e3, 3),
           "timed_passes": len(runs), "ms_per_step_min": round(runs[0][0] * 1e3, 3), "ms_per_step_max": round(runs[-1][0] * 1e3, 3),
           "search_ms": round(tm["search_ms"], 3), "gmm_ms": round(gmm_alone if gmm_alone is not None else tm["gmm_ms"], 3),
           "scored_ahead": bool(tm["prefetched"]), "batches_in_flight": (depth + 1) if depth else (2 if two else 1),
           "searched_ahead_frames": int(tm["ahead_frames"]), "decode_calls": n_calls,
           "per_stream_frame": {k: round(v / max(1, frames), 1) for k, v in st.items() if k.startswith("tot_")},
           "hyps_found": int(sum(int(h.n > 0) for h in hyps)),
           "setup_s": round(time.perf_counter() - t0 - sum(r[0] for r in runs), 1)}
    # (which kernel the leg's frames went through: the pipeline's slots, the slot kernel as a plain launch, else clusters of k_search)
    kernel = "k_slot" if depth else ("k_slot_batch" if tm.get("slot_launches", 0) > 0 else "k_search")
    out["roofline"] = roofline_of(st, am.max_n, tm, leg_traffic(pmc_leg, max(1, tm["search_launches"])) if pmc_leg else None,
                                  design_bytes(st, am.max_n, am.n_gmm, frames, row_in_lds=kernel != "k_search" and am.n_gmm <= 3072))
    out["roofline"]["kernel"] = kernel
    out["scoring"] = scoring
    if oracle_utts > 0:
        from oracle.oracle import OracleAM, OracleDecoder, OracleNet
        od = OracleDecoder(oracle_net if oracle_net is not None else OracleNet(net), OracleAM(am), main_beam=beam, max_hyps=max_hyps)
        if oracle_feats is not None:
            sample, got = list(oracle_feats)[:oracle_utts], dec.decode_batch(list(oracle_feats)[:oracle_utts])
            what = "%d extra short utterances on the leg's graph (the batch's own would take the oracle minutes)" % len(sample)
        else:
            sample, got = feats[:min(oracle_utts, U)], hyps
            what = "first %d utterances of the batch" % min(oracle_utts, U)
        secs, fr, same, found = 0.0, 0, 0, 0
        for u, x in enumerate(sample):
            o = od.decode(x)
            secs += o.cpu_seconds; fr += x.shape[0]; found += int(o.n > 0)
            same += int(got[u].n == o.n and np.array_equal(got[u].label, o.label) and np.array_equal(got[u].time, o.time))
        out["cpu_oracle"] = {"frames_per_s": round(fr / max(secs, 1e-9), 1), "utts": len(sample), "frames": fr, "identical_1best": same,
                             "oracle_hyps_found": found, "sample": what}
    dec.close()
    del d_feats
    torch.cuda.empty_cache()
    return out


def compose_leg(seed, dev, pushing=False, oracle_utts=0):
    """BASELINE.json configs[4] (separate C.L and G) as far as it is built: the two transducers are composed
    ON THE DEVICE (jd_net_compose) and the result is decoded by the static search; no oracle exists for the
    reference's on-the-fly decoder (it is not built and no longer compiles), so this leg has no cpu line."""
    from juicer_amd import capi, synth
    am = synth.make_models(seed, n_gmm=3000, n_hmm=2000, n_mix=16, n_tm=8, sep=0.6, with_tee=True)
    t0 = time.perf_counter()
    cl, g = synth.make_cl_g(seed, am, n_words=20000, n_succ=40, n_tri=200000, n_succ3=8, with_sp=True)
    t_gen = time.perf_counter() - t0
    ncl, ng = capi.Network.from_synth(cl, 1.0, 0.0), capi.Network.from_synth(g, 10.0, 0.0)
    best = None
    for _ in range(2):
        t0 = time.perf_counter()
        net = capi.Network.compose(ncl, ng, device=dev.index, max_states=1 << 26, max_arcs=1 << 27, pushing=pushing)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    feats = [synth.sample_utterance(seed + 100 + u, g, am, 8)[0] for u in range(64)]

    class _Size:                                                   # what run_leg prints about the graph
        n_arcs = net.n_arcs
    onet = None
    if oracle_utts > 0:                                            # the CPU oracle decodes the DEVICE-COMPOSED graph (its CSR, read back)
        from oracle.oracle import OracleNet
        c = net.csr()
        fs = np.nonzero(np.isfinite(c["fin_w"]))[0].astype(np.int32)
        onet = OracleNet.from_csr(net.n_states, net.init_state, c["row_ptr"], c["to"], c["w"], c["ilab"], c["olab"], fs, c["fin_w"][fs])
        del c
    out = run_leg("configs[4], composed on the device (lexicon tree o back-off trigram%s)" % (", weights pushed" if pushing else ""),
                  am, _Size, feats, 200.0, 0, dev, gnet=net, pmc_leg="clg" if (not pushing and seed == 0) else None,
                  oracle_utts=oracle_utts, oracle_net=onet)
    del onet
    if not pushing and os.environ.get("JD_BENCH_NO_LAZY") != "1":      # (tools/leg_pmc.sh counts the static leg's launches only)
        out["search_driven"] = lazy_part(ncl, ng, am, feats, dev, net.n_states, net.n_arcs, gnet=net)
    out["composition"] = {"pushing": bool(pushing), "cl_arcs": int(cl.n_arcs), "g_arcs": int(g.n_arcs), "states": net.n_states, "arcs": net.n_arcs,
                          "seconds_incl_pcie": round(best, 4), "arcs_per_s": round(net.n_arcs / best, 1),
                          "generator_seconds": round(t_gen, 1)}
    return out


def lazy_part(ncl, ng, am, feats, dev, full_states, full_arcs, gnet):
    """The same leg with nothing composed beforehand (jd_net_create_lazy): the search expands the composed
    states it reaches.  Cold = the first pass, while the graph grows; warm = the same utterances again."""
    from juicer_amd import capi
    models = capi.Models.from_htk(am)
    want = capi.Decoder(gnet, models, main_beam=200.0, device=dev.index, max_streams=len(feats)).decode_batch(feats)
    t0 = time.perf_counter()
    lz = capi.Network.lazy(ncl, ng, models, device=dev.index, max_states=1 << 22, max_arcs=1 << 23)
    t_create = time.perf_counter() - t0
    dec = capi.Decoder(lz, models, main_beam=200.0, device=dev.index, max_streams=len(feats))
    frames = sum(f.shape[0] for f in feats)
    times, cold_ms = [], None
    for _ in range(3):
        t0 = time.perf_counter()
        hyps = dec.decode_batch(feats)
        times.append(time.perf_counter() - t0)
        if cold_ms is None:
            cold_ms = dec.last_timing()["search_ms"]
    ns, na = lz.lazy_size()
    same = sum(int(a.n == b.n and np.array_equal(a.label, b.label) and np.array_equal(a.time, b.time)
                   and np.array_equal(np.asarray(a.score, np.float32).view(np.uint32), np.asarray(b.score, np.float32).view(np.uint32)))
               for a, b in zip(hyps, want))
    tm = dec.last_timing()
    dec.close()
    return {"create_s": round(t_create, 4), "cold_search_ms": round(cold_ms, 3), "cold_pass_wall_s_incl_arena_setup": round(times[0], 3), "warm_frames_per_s_incl_pcie": round(frames / min(times[1:]), 1),
            "warm_search_ms": round(tm["search_ms"], 3), "states_expanded": ns, "arcs_expanded": na,
            "fraction_of_full_composition": round(ns / max(1, full_states), 4),
            "identical_to_composed_first": "%d/%d" % (same, len(feats))}


def spawn_ranks(n):
    """`python bench.py --gpus N` outside torchrun: start the N ranks ourselves."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % n,
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--utts-per-gpu", type=int, default=64)
    ap.add_argument("--total-utts", type=int, default=0,
                    help="strong scaling (BASELINE.json configs[2] with 512): ONE batch of this many utterances dealt over the ranks by length")
    ap.add_argument("--arcs", type=int, default=1_000_000)
    ap.add_argument("--beam", type=float, default=150.0)
    ap.add_argument("--max-hyps", type=int, default=0)
    ap.add_argument("--cpu-sample-utts", type=int, default=12)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true")
    ap.add_argument("--no-score-ahead", action="store_true",
                    help="score every batch's table right before its search (serial) instead of beside the previous batch's search")
    ap.add_argument("--no-search-ahead", action="store_true",
                    help="one batch in flight: the decoder gets streams for one batch only and the announcements run one batch ahead")
    ap.add_argument("--pipeline-depth", type=int, default=9,
                    help="weak scaling: batches announced ahead of the one being decoded, through the resident slot kernel (DESIGN.md 3.6); "
                         "0 = two batches in flight, one k_search launch per step")
    ap.add_argument("--pipeline-slots", type=int, default=256,
                    help="one-workgroup slots of the resident pipeline, dealt one per CU (DESIGN.md 3.4 has the measured sweep: 240 / 256 / 272 / "
                         "288 slots 2.34 / 2.38 / 2.13 / 2.12 M frames/s)")
    ap.add_argument("--gather-every", type=int, default=0,
                    help="several ranks: 0 = the 1-best records of all timed steps travel in ONE RCCL all_gather at the end of the timed "
                         "region, behind jd_dec_quiesce; 1 = one all_gather per step (two batches in flight instead of the resident pipeline)")
    ap.add_argument("--scoring", choices=("exact", "fast"), default="exact",
                    help="exact (default): the reference's roundings, log-likelihoods bit-identical to the CPU oracle; fast: fused multiply-add "
                         "distance + fp32 logAdd on the hardware's exp / log (jd_dec_set_scoring: labels and times identical, scores within 1e-4)")
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()

    share_gpu = os.environ.get("JD_BENCH_SHARE_GPU") == "1"      # development: all ranks on GPU 0, gloo
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import torch
        have = torch.cuda.device_count()
        if have < args.gpus and not share_gpu:
            raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible" % (args.gpus, have))
        sys.exit(spawn_ranks(args.gpus))

    t_proc = time.perf_counter()
    import torch
    import torch.distributed as dist
    from juicer_amd import build as jbuild
    from juicer_amd import capi, parallel, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(1, args.gpus):
        raise SystemExit("WORLD_SIZE %d != --gpus %d" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: juicer_amd has no CPU fallback")
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    marks = [("start", t_proc)]                                    # wall-clock budget of the run, printed to stderr by rank 0

    def mark(what):
        marks.append((what, time.perf_counter()))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # stdout carries ONE JSON line: RCCL's version banner (NCCL_DEBUG=VERSION, exported on the GPU boxes) goes to stderr
        os.environ.setdefault("NCCL_DEBUG_FILE", "/dev/stderr")
        if os.environ.get("NCCL_DEBUG", "").upper() == "VERSION":
            del os.environ["NCCL_DEBUG"]
        import datetime
        # (a rank that dies leaves the others in a collective: they give up after ten minutes, not after RCCL's half hour)
        if share_gpu:
            dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=600))
        else:
            dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(seconds=600))
    if rank == 0:
        jbuild.build()
    bar_kw = {} if share_gpu else {"device_ids": [local_rank]}
    if world > 1:
        dist.barrier(**bar_kw)
    mark("import + process group")

    U = args.utts_per_gpu
    strong = args.total_utts > 0
    shard = None
    refs = None
    if strong:
        # one fixed batch: every rank generates the same utterances (seeded) and takes the ones dealt to it by
        # length (parallel.shard_lpt: longest first, each to the rank with the fewest frames so far)
        am, net, all_feats, all_refs = synth.config_c2(seed=args.seed, n_utts=args.total_utts, target_arcs=args.arcs)
        shards = parallel.shard_lpt([f.shape[0] for f in all_feats], world)
        shard = shards[rank]
        # what the 1-GPU measurements say every rank's share should take (the first real SCALE run can be read against it):
        # a rank's utterances through the slot kernel as a plain launch at the 512-utterances-on-one-GPU leg's rate (2.3 M frames/s),
        # but no less than its longest utterance's chain of frames (100 us per frame on one slot)
        predicted_rank_ms = [round(max(max([all_feats[u].shape[0] for u in sh] or [0]) * 100e-3 + 1.5,
                                       sum(all_feats[u].shape[0] for u in sh) / 2.3e6 * 1e3), 2) for sh in shards]
        per_rank = max(len(x) for x in shards)
        feats = [all_feats[u] for u in shard]
        refs = [all_refs[u] for u in shard]
        del all_feats
        U = max(1, len(feats))
    else:
        am, net, feats, refs = synth.config_c2(seed=args.seed, n_utts=U, target_arcs=args.arcs, utt_offset=rank * U)
        per_rank = U
    gnet = capi.Network.from_synth(net)
    gam = capi.Models.from_htk(am)
    mark("synthetic graph, models, utterances")
    # How the batches share the chip.  Weak scaling: batches of U utterances follow each other - through the slots of
    # a search kernel that stays (a batch is scored whole when it is announced, a slot takes the next queued utterance the moment its
    # own is through, a step hands back the oldest batch: still ITS results, decoded in full), or with two batches in flight.
    # Several ranks run the same path.  A collective's kernels must not be queued on a device whose search kernel STAYS (HIP maps
    # streams onto a few hardware queues: tools/resident_alias_probe.py), so the 1-best records of the K steps travel in ONE RCCL
    # all_gather at the end of the timed region, behind jd_dec_quiesce - utterances are independent, DecoderBatchTest.cpp:738-771: a
    # rank has its own results the moment its step returns.  --gather-every 1 keeps a collective per step and two batches in flight.
    ahead = not args.no_score_ahead
    two_in_flight = ahead and not args.no_search_ahead and not strong
    per_step_gather = world > 1 and args.gather_every == 1
    depth0 = args.pipeline_depth if (two_in_flight and args.pipeline_depth > 0 and not per_step_gather) else 0
    fail_spec = os.environ.get("JD_BENCH_FAIL", "")               # test knob "rank:phase" (phase: create | warmup | timed): that rank's pipeline attempt fails there
    fail_rank, fail_phase = (int(fail_spec.split(":")[0]), fail_spec.split(":")[1]) if ":" in fail_spec else (-1, "")
    empty_index = [] if shard is not None else None

    def any_rank(flag):
        """collective: did ANY rank raise?  (every rank takes the same decision, so the collectives that follow stay matched)"""
        if world == 1:
            return bool(flag)
        t = torch.tensor([1.0 if flag else 0.0], dtype=torch.float32, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return bool(t.item() > 0.0)

    def barrier():
        if world > 1:
            dist.barrier(**bar_kw)

    pipeline_error = None
    dec = None
    attempts = (depth0, 0) if depth0 else (0,)
    for ai, depth in enumerate(attempts):                          # (should the pipeline fail on ANY rank: the same measurement, two batches in flight, on ALL)
        err = None                                                 # this rank's first error of the attempt
        pending = []                                               # several ranks: steps whose records have not travelled yet
        hyps = allh = None
        ph = {"phase": "create"}

        def forced(phase):
            if depth and rank == fail_rank and fail_phase == phase:
                raise capi.JuicerAmdError(-5, "forced failure (JD_BENCH_FAIL=%s)" % fail_spec)
        try:
            forced("create")
            dec = capi.Decoder(gnet, gam, main_beam=args.beam, max_hyps=args.max_hyps, device=local_rank,
                               max_streams=min(U, 512) if strong else (args.pipeline_slots if depth else (2 * U if two_in_flight else U)))
            if args.scoring == "fast":
                dec.set_scoring(capi.SCORE_FAST)
            if depth:                                              # the interface: jd_dec_set_pipeline (include/juicer_amd.h)
                dec.set_pipeline(capi.FLOW_RESIDENT, depth + 1, args.pipeline_slots)
        except capi.JuicerAmdError as e:
            err = e
        offs = np.zeros(len(feats) + 1, dtype=np.int64)
        offs[1:] = np.cumsum([f.shape[0] for f in feats])
        frames_local = int(offs[-1])
        d_feats = torch.from_numpy(np.concatenate(feats) if feats else np.zeros((0, am.D), np.float32)).to(dev)     # inputs resident in HBM
        torch.cuda.synchronize()
        stream = torch.cuda.current_stream().cuda_stream

        def local_step():
            # One step = one pass over one batch: its search, and one scoring of a batch's likelihood table.  Batches follow each other,
            # so the table of a LATER batch (here: the same synthetic batch again, scored from its features every time) is scored beside
            # this batch's search (jd_dec_prefetch_scores) - K timed steps hold K searches and K scorings either way, and every step
            # returns ITS batch's results, decoded in full.
            forced(ph["phase"])
            if ahead:
                dec.prefetch_scores(d_feats.data_ptr(), offs, stream)
            return dec.decode_batch_device(d_feats.data_ptr(), offs, stream)

        def run_steps(n, each=None, acc=None):
            """n steps; a rank whose decoder failed keeps taking part in every collective (with empty records) - the others must
            not be left waiting in an all_gather for a rank that has raised"""
            nonlocal err, hyps, allh
            tm = {}
            for _ in range(n):
                ts = time.perf_counter()
                h = None
                if err is None:
                    try:
                        h = local_step()
                    except capi.JuicerAmdError as e:
                        err = e
                if world > 1:
                    rec, ix = (h, shard) if h is not None else ([], empty_index)
                    if per_step_gather:
                        try:
                            allh = parallel.gather_hyps(rec, per_rank, device=dev, index=ix)
                        except ValueError as e:                    # (a rank sent empty records: its utterances are missing)
                            err = err or capi.JuicerAmdError(-5, "gather: %s" % e)
                    else:
                        pending.append((rec, ix))
                if h is not None:
                    hyps = h
                    if world == 1:
                        allh = h
                    if each is not None:
                        each.append(round((time.perf_counter() - ts) * 1e3, 3))
                    if acc is not None:
                        tm = dec.last_timing()
                        for k in acc:
                            acc[k] += tm[k]
            return tm

        def travel():
            """the records of the steps since the last exchange: ONE all_gather (the resident kernel has left: quiesce)"""
            nonlocal err
            if not pending:
                return None
            try:
                return parallel.gather_hyps_steps(pending, per_rank, device=dev)
            except ValueError as e:
                err = err or capi.JuicerAmdError(-5, "gather: %s" % e)
                return None
            finally:
                del pending[:]

        def quiesce():
            """the resident kernel leaves - ALSO on a rank whose decoder has failed: the collective that follows moves device
            tensors, and anything queued on a device whose search kernel stays may wait for it for ever"""
            nonlocal err
            if dec is None:
                return
            try:
                dec.quiesce()
            except capi.JuicerAmdError as e:
                err = err or e

        # ---- fill + warm-up
        ph["phase"] = "warmup"
        if err is None:
            try:
                if depth:
                    for _ in range(depth):                         # (the announcements run `depth` batches ahead)
                        dec.prefetch_scores(d_feats.data_ptr(), offs, stream)
                elif two_in_flight:
                    dec.prefetch_scores(d_feats.data_ptr(), offs, stream)      # (the announcements run two batches ahead)
            except capi.JuicerAmdError as e:
                err = e
        run_steps((depth + 2 if depth else 0) + args.warmup)       # (the pipeline fills: its first batches come back in a burst)
        # (a device-wide synchronisation waits for every kernel on the device: the pipeline's resident kernel lets its running
        # commands run out and leaves - jd_dec_quiesce - and comes back with the first timed step, inside the brackets)
        quiesce()
        travel()
        if not any_rank(err is not None):
            mark("decoder, arenas, pipeline fill, warm-up (%d steps)" % ((depth + 2 if depth else 0) + args.warmup))
            # ---- the timed region
            ph["phase"] = "timed"
            barrier(); torch.cuda.synchronize()
            ps0 = dec.pipeline_stats()
            t0 = time.perf_counter()
            acc = {"gmm_ms": 0.0, "search_ms": 0.0, "gmm_wait_ms": 0.0, "search_launches": 0, "gmm_launches": 0, "relaunches": 0, "prefetched": 0,
                   "ahead_frames": 0}
            each = []
            tm = run_steps(args.steps, each, acc)
            quiesce()
            gathered = travel()                                    # (inside the timed region: the one collective of the K steps)
            if gathered:
                allh = gathered[-1]
            torch.cuda.synchronize(); barrier()
            elapsed = time.perf_counter() - t0
            if not any_rank(err is not None):
                ps1 = dec.pipeline_stats()
                mark("timed region (%d steps)" % args.steps)
                break
        # ---- some rank failed: every rank drops this attempt
        msg = str(err) if err is not None else "another rank's decoder failed"
        if rank == 0 or err is not None:
            print("bench.py[rank %d]: %s failed (%s)%s" % (rank, "the resident pipeline" if depth else "the decoder", msg,
                                                           ": measuring with two batches in flight on every rank" if ai + 1 < len(attempts) else ""),
                  file=sys.stderr, flush=True)
        try:
            if dec is not None:
                dec.close()
        except Exception:
            pass
        dec = None
        del d_feats
        torch.cuda.empty_cache()
        if ai + 1 == len(attempts):
            if world > 1:
                dist.destroy_process_group()
            raise SystemExit("bench.py: the decoder failed on rank %d: %s" % (rank, msg))
        pipeline_error = msg
    # `value` counts the frames the device really searched between the two brackets: with batches through the resident kernel
    # that is what its slots report having advanced (jd_dec_pipeline_stats: a batch handed back inside the region was partly
    # searched before it, batches announced inside it are partly searched behind it); with one launch per step it is K batches
    frames_timed = (ps1["frames_searched"] - ps0["frames_searched"]) if depth else frames_local * args.steps
    # (outside the timed region) what a caller gets who cannot announce that far ahead: ONE batch of 64 at a time.
    #   serial order:  nothing announced - the batch's table is scored, then it is searched (re-planned at will)
    #   one ahead:     the next batch announced before each decode - its table is scored beside this batch's search
    dec.set_pipeline(capi.FLOW_SERIAL)
    dec.prefetch_scores(0, None)
    dec.decode_batch_device(d_feats.data_ptr(), offs, stream)          # (a launch of this shape has been planned once)
    t1 = time.perf_counter()
    dec.decode_batch_device(d_feats.data_ptr(), offs, stream)
    torch.cuda.synchronize()
    serial_ms = (time.perf_counter() - t1) * 1e3
    tm_serial = dec.last_timing()
    one_ahead = []
    dec.prefetch_scores(d_feats.data_ptr(), offs, stream)
    for i in range(5):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        dec.prefetch_scores(d_feats.data_ptr(), offs, stream)
        dec.decode_batch_device(d_feats.data_ptr(), offs, stream)
        torch.cuda.synchronize()
        if i >= 2:
            one_ahead.append(((time.perf_counter() - t1) * 1e3, dec.last_timing()))
    one_ahead.sort(key=lambda r: r[0])
    one_ahead_ms, tm_one = one_ahead[len(one_ahead) // 2]
    dec.prefetch_scores(0, None)
    # (outside the timed region) which part of the likelihood table does the search read?  SURVEY.md 8d's Ug: the reference
    # scores a tied state only when a token that passed the emit threshold asks for it; every state of every frame is scored here
    dec.debug_cells(True)
    dec.decode_batch_device(d_feats.data_ptr(), offs, stream)
    cells_read, cells_total = dec.debug_cells(False)
    mark("single-batch figures, cells read")
    if world > 1:
        t = torch.tensor([elapsed, float(frames_local), float(frames_timed)], dtype=torch.float64, device=dev)
        tmax = t.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone(); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0]); frames_total = float(tsum[1]); frames_timed_total = float(tsum[2])
        n_gathered = len(allh)
    else:
        frames_total = float(frames_local); frames_timed_total = float(frames_timed)
        n_gathered = len(hyps)

    if rank != 0:
        if world > 1:
            dist.barrier(**bar_kw)                      # rank 0 finishes its CPU baseline first
            dist.destroy_process_group()
        return

    steps = args.steps
    fps = frames_timed_total / elapsed
    D, G, M, MN = am.D, am.n_gmm, am.max_mix, am.max_n
    st = {k: sum(h.stats[k] for h in hyps) for k in hyps[0].stats}       # one step's batch totals (rank 0)
    default_cfg = (args.arcs == 1_000_000 and args.beam == 150.0 and args.max_hyps == 0 and U == 64 and args.seed == 0 and not strong
                   and args.scoring == "exact")
    # HBM traffic per launch from the committed PMC passes (profiles/<round>_*_traffic.json, tools/collect_r06.sh: separate --pmc
    # runs, calibrated as calibrated_traffic says) - reported only for the default workload and only while the kernels' sources are
    # the ones the passes were taken on (else null)
    step_tm = dict(tm)
    step_tm["search_ms"] = acc["search_ms"] / steps
    step_tm["search_launches"] = max(1, acc["search_launches"] // steps)
    traffic = leg_traffic("c2", step_tm["search_launches"]) if default_cfg else None
    if depth:
        traffic = slot_traffic(frames_local) if default_cfg else None
        # (the resident kernel is busy all the time: a batch's share of it is the time the slots took for one batch's worth of the
        # frames they really advanced inside the brackets)
        step_tm["search_ms"] = elapsed * 1e3 * frames_local / max(1.0, float(frames_timed))
    slot_kernel = bool(depth) or tm.get("slot_launches", 0) > 0
    roofline = roofline_of(st, MN, step_tm, traffic, design_bytes(st, MN, G, frames_local, row_in_lds=slot_kernel and G <= 3072))
    if depth:
        roofline["kernel"] = "k_slot"
        roofline["traffic_is"] = "proxy: k_slot_batch counted on configs[2]'s 512-utterance batch, bytes per stream-frame x frames"
        roofline["launch"] = "ONE launch spans the timed region; a batch's share = region x batch frames / frames the slots advanced"
        roofline["l2_requests"] = slot_l2_requests(fps) if default_cfg else None
    roofline["frac_is"] = "SURVEY 8(d) bytes of the REFERENCE's work / time; frac_design: bytes this design requests; frac_measured: counted HBM bytes"
    gmm_flops = frames_local * G * M * (3.0 * D + 4.0)
    gmm_bytes = G * M * (2 * D + 1) * 4.0 + frames_local * D * 4.0 / max(1, tm["gmm_launches"])
    roofline["search_ms_per_step"] = round(step_tm["search_ms"], 3)
    # one batch's counters per stream-frame: the reference's figures (tot_insts_in .. tot_paths) and what the kernels took up (tot_recs_read ..)
    roofline["per_stream_frame"] = {k: round(v / max(1, frames_local), 1) for k, v in st.items() if k.startswith("tot_")}
    # the companion kernel is VALU-bound: per (frame pair, mixture) 4 packed fp32 instructions per dimension
    # + ~116 for the two logAdd steps, 4 cycles each on 1024 SIMDs; fast scoring: 2 packed FMAs per dimension + ~24
    per_mix = (4.0 * D + 116.0) if args.scoring == "exact" else (2.0 * D + 24.0)
    gmm_valu_ms = (frames_local / 128.0) * G * M * per_mix * 4.0 / 1024.0 / 2.4e9 * 1e3
    gmm_ms = tm_serial["gmm_ms"]                                   # the kernel on its own (the serial step behind the timed region)
    n_ahead = acc["prefetched"]
    roofline["gmm"] = {"kernel": ("jd_gmm_kernel39" if args.scoring == "exact" else "jd_gmm_fast39") if D == 39 else "jd_gmm_kernel", "bound": "valu",
                       "scoring": args.scoring,
                       "ms_per_step": round(gmm_ms, 3),
                       "valu_bound_ms": round(gmm_valu_ms, 3),
                       "frac": round(gmm_valu_ms / max(gmm_ms, 1e-9), 4),
                       "valu_tflops": round(gmm_flops / max(gmm_ms, 1e-9) / 1e9, 3),
                       "algorithmic_bytes_per_launch": round(gmm_bytes, 1),
                       "search_waited_ms_per_step": round(acc["gmm_wait_ms"] / steps, 3),
                       "scored_ahead_steps": int(n_ahead),
                       "cells_scored": int(cells_total), "cells_read_by_the_search": int(cells_read),
                       "cells_read_frac": round(cells_read / max(cells_total, 1), 4),
                       "span_beside_search_ms": round(acc["gmm_ms"] / steps, 3) if n_ahead else None,
                       "serial_order_ms_per_step": round(serial_ms, 3),
                       "serial_order_search_ms": round(tm_serial["search_ms"], 3)}

    # the same batch in the serial order (scored, then searched with re-planning at will): the kernel's own best figure
    ser = roofline_of(st, MN, tm_serial, leg_traffic("c2", max(1, tm_serial["search_launches"])) if default_cfg else None)
    roofline["serial_order"] = {"ms_per_step": round(serial_ms, 3), "search_ms": round(tm_serial["search_ms"], 3),
                                "launches": tm_serial["search_launches"], "achieved": ser["achieved"], "frac": ser["frac"],
                                "frac_measured": ser["frac_measured"]}

    # ---- CPU baseline: the oracle (a port of the reference algorithm) on a bounded sample, one core
    cpu = None
    wer = None
    if not args.no_cpu_baseline and world == 1:                   # (rank 0 at N = 1 only: the other ranks would wait for it)
        from oracle.oracle import OracleAM, OracleDecoder, OracleNet
        ns = min(args.cpu_sample_utts, U)
        od = OracleDecoder(OracleNet(net), OracleAM(am), main_beam=args.beam, max_hyps=args.max_hyps)
        secs, fr, same = 0.0, 0, 0
        for u in range(ns):
            o = od.decode(feats[u])
            secs += o.cpu_seconds; fr += feats[u].shape[0]
            g = hyps[u]
            same += int(g.n == o.n and np.array_equal(g.label, o.label) and np.array_equal(g.time, o.time))
        cpu = {"value": round(fr / secs, 1), "unit": "frames/s", "cores": 1, "kind": "port",
               "sample": "first %d utterances of the batch (%d frames), clock() around init..finish (DecoderSingleTest.cpp:259-300)" % (ns, fr),
               "identical_1best": "%d/%d" % (same, ns)}
        # the reference's two-thread organisation (WFSTDecoderLiteThreading + HTKFlatModelsThreading: a search
        # thread and a scoring thread), restated; wall time with two busy cores, on a third of the sample
        try:
            nt = max(1, ns // 3)
            wall, frt, same_t = 0.0, 0, 0
            for u in range(nt):
                o = od.decode(feats[u], threading=True)
                wall += o.cpu_seconds; frt += feats[u].shape[0]
                g = hyps[u]
                same_t += int(g.n == o.n and np.array_equal(g.label, o.label) and np.array_equal(g.time, o.time))
            cpu["two_thread_core"] = {"value": round(frt / wall, 1), "unit": "frames/s", "cores": 2, "kind": "port",
                                      "sample": "first %d utterances (%d frames), wall time" % (nt, frt), "identical_1best": "%d/%d" % (same_t, nt)}
        except RuntimeError as e:                                 # models with a skip into the exit state (refused like the reference)
            cpu["two_thread_core"] = {"error": str(e)}
        # the reference's own classes, measured in the build container (quoted constants: the GPU box has no /root/reference)
        cpu["reference_cpu"] = reference_cpu()
        mark("cpu_baseline (%d utterances, one core)" % ns)
        # BASELINE.json: "1-best WER vs ref" - every utterance of the step against the CPU oracle, and against the transcript the
        # utterances were sampled from (after the timed region; the oracle on all host cores)
        wer = wer_vs_oracle(net, am, feats, hyps, args.beam, args.max_hyps)
        if refs is not None:
            e_t = sum(word_errors(list(h.label[::-1]) if h.n > 0 else [], list(r)) for h, r in zip(hyps, refs))
            wer["wer_vs_sampled_transcript"] = round(e_t / max(1, sum(len(r) for r in refs)), 6)
        mark("wer_vs_oracle (%d utterances, %d threads)" % (wer["utterances"], wer["oracle_threads"]))
        # (the essentials again where the driver's record keeps them: inside cpu_baseline)
        cpu["wer_vs_oracle"] = wer["wer"]
        cpu["identical_1best_whole_step"] = "%d/%d" % (wer["identical_1best"], wer["utterances"])
        cpu["identical_scores_bitwise_whole_step"] = "%d/%d" % (wer["identical_scores_bitwise"], wer["utterances"])

    # what ONE batch of 64 costs a caller that does not announce nine batches ahead - beside `value`, not inside it
    one = roofline_of(st, MN, tm_one, leg_traffic("c2", max(1, tm_one["search_launches"])) if default_cfg else None)
    single_batch = {"serial_order": {"ms": round(serial_ms, 3), "frames_per_s": round(frames_local / serial_ms * 1e3, 1), "frac": ser["frac"],
                                     "what": "nothing announced: the batch's table is scored, then it is searched"},
                    "one_ahead": {"ms": round(one_ahead_ms, 3), "frames_per_s": round(frames_local / one_ahead_ms * 1e3, 1), "frac": one["frac"],
                                  "what": "the next batch announced before each decode (jd_dec_prefetch_scores); one batch on the chip at a time"},
                    "note": "measured behind the timed region on rank 0 (one pass / median of 3); not part of `value`"}
    name = "configs[1]" if default_cfg else "configs[1]-shaped (non-default size / pruning / scoring)"
    # The line: what a reader needs first comes first, and every string inside config / roofline / cpu_baseline stays under 128
    # characters (the driver's record keeps those objects and cuts longer strings)
    out = {"metric": "frames/sec decoded", "value": round(fps, 1), "unit": "frames/s", "n_gpus": world,
           "steps": steps, "warmup": args.warmup, "ms_per_step": round(elapsed / steps * 1e3, 3),
           "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f32",
           "data": "synthetic", "xRT": round(fps / 100.0, 1),
           "config": {"workload": "%s: %d-arc C.L.G, %dx%d-mix GMMs, D=%d, %s, beam %g, maxHyps %d"
                                  % (name, net.n_arcs, G, M, D, ("%d utts in all" % args.total_utts) if strong else ("%d utts/GPU" % U),
                                     args.beam, args.max_hyps),
                      "frames_per_step": int(frames_total), "utts_per_gpu": U, "gathered_hyps": n_gathered,
                      "parallelism": ("one batch of %d utterances dealt by length over %d rank(s)" % (args.total_utts, world)) if strong
                                     else "utterance-sharded x%d" % world,
                      "scoring": args.scoring,
                      "batches_in_flight": (depth + 1) if depth else (2 if two_in_flight else 1),
                      "pipeline": ("resident slot kernel: %d one-workgroup slots (one per CU, scoring beside them), %d batches announced ahead"
                                   % (args.pipeline_slots, depth)) if depth else None,
                      "pipeline_error": pipeline_error[:120] if pipeline_error else None,
                      "gather": None if world == 1 else ("one all_gather per step" if per_step_gather else
                                                         "ONE all_gather of the %d steps' records, inside the timed region, behind jd_dec_quiesce" % steps),
                      # ONE batch at a time (a caller that cannot announce nine batches ahead): measured behind the timed region
                      "single_batch_serial_ms": round(serial_ms, 3), "single_batch_serial_fps": round(frames_local / serial_ms * 1e3, 1),
                      "single_batch_one_ahead_ms": round(one_ahead_ms, 3), "single_batch_one_ahead_fps": round(frames_local / one_ahead_ms * 1e3, 1),
                      "predicted_rank_ms": predicted_rank_ms if strong else None,
                      "search_ahead_frames_per_step": int(acc["ahead_frames"] // max(steps, 1)),
                      "streams_per_gpu": dec.max_streams},
           "roofline": roofline, "cpu_baseline": cpu, "wer_vs_oracle": wer,
           "single_batch": single_batch,
           "frames_timed": int(frames_timed_total),
           "frames_timed_is": ("stream-frames the resident kernel's slots advanced between the two brackets (jd_dec_pipeline_stats), all ranks"
                               if depth else "steps x the frames of a batch, all ranks"),
           "ms_each_step": each if steps <= 32 else None}

    # ---- the other single-GPU workloads of BASELINE.json (not part of `value`)
    if world == 1 and not args.no_extra_legs:
        dec.close()
        del d_feats
        torch.cuda.empty_cache()
        legs = {}
        try:
            no = 0 if args.no_cpu_baseline else 2                   # utterances the CPU oracle decodes per leg
            pipe = (depth, args.pipeline_slots) if depth else None
            if args.scoring == "exact" and depth:                   # the headline with the scoring option: same search, cheaper table
                legs["configs1_fast_scoring"] = run_leg("configs[1], jd_dec_set_scoring(JD_SCORE_FAST): FMA distance + fp32 logAdd (scores within 1e-4)",
                                                        am, net, feats, args.beam, args.max_hyps, dev, oracle_utts=no, pipe=pipe, passes=24,
                                                        scoring="fast")
            legs["configs1_maxhyps6000"] = run_leg("configs[1] + histogram pruning", am, net, feats, args.beam, 6000, dev, oracle_utts=no,
                                                   two=two_in_flight, pipe=pipe, passes=24 if pipe else 4, pmc_leg="hyps" if default_cfg else None)
            if depth:                                               # the headline's batches with TWO of them in flight, one launch per step
                legs["configs1_two_batches_in_flight"] = run_leg("configs[1], two batches in flight (one k_search launch per step)",
                                                                 am, net, feats, args.beam, args.max_hyps, dev,
                                                                 two=True, pmc_leg="c2" if default_cfg else None)
            elif two_in_flight:                                     # ... or through the resident search kernel
                legs["configs1_through_the_resident_kernel"] = run_leg("configs[1], batches through the resident slot kernel", am, net, feats,
                                                                       args.beam, args.max_hyps, dev, pipe=(9, 256), passes=14)
            # configs[2]'s batch (512 utterances) on ONE GPU: more streams than the chip has CUs, so every pass is ONE launch of the
            # slot kernel, a workgroup per utterance, two per CU, the dispatcher dealing the next one when one leaves - k_slot_batch
            _, _, f512, _ = synth.config_c2(seed=args.seed, n_utts=512, target_arcs=args.arcs)
            legs["configs2_batch_512_on_one_gpu"] = run_leg("configs[2]'s 512-utterance batch on one GPU, 512 streams: one launch of the slot kernel per pass",
                                                            am, net, f512, args.beam, 0, dev, oracle_utts=no, max_streams=512, passes=4,
                                                            pmc_leg="c512slot" if default_cfg else None)
            del f512
            # the search's other record layout at bench size: HMMs of 1 .. 6 emitting states with skips (k_slot<6>, general predecessor loop)
            am6, n6, f6, _ = synth.config_c2_mixed(seed=args.seed, n_utts=64, target_arcs=args.arcs)
            legs["configs1_mixed_topologies"] = run_leg("configs[1]'s graph with HMMs of 1-6 emitting states (144-byte records, general topologies)", am6, n6, f6,
                                                        args.beam, 0, dev, oracle_utts=no, pipe=pipe, passes=24 if pipe else 4)
            del am6, n6, f6
            a4, n4, f4, _ = synth.config_c4(seed=args.seed, n_utts=64, n_words=10000, n_tri_hist=100_000)
            legs["north_star_10M_beam200"] = run_leg("north_star target (trigram-shaped)", a4, n4, f4, 200.0, 0, dev,
                                                     oracle_utts=no, pmc_leg="north" if args.seed == 0 else None)
            del a4, n4, f4
            a4, n4, f4, _ = synth.config_c4(seed=args.seed, n_utts=8)
            # (the oracle manages ~7 frames/s on this graph at beam 300: it gets two 2-word utterances of their own)
            short = [synth.sample_utterance_walk(args.seed + 5000 + u, n4, a4, 2)[0] for u in range(2)] if no else None
            legs["configs3_50M_beam300"] = run_leg("configs[3]", a4, n4, f4, 300.0, 0, dev, pmc_leg="c3" if args.seed == 0 else None,
                                                   oracle_utts=no, oracle_feats=short)
            del a4, n4, f4
            legs["configs4_device_composition"] = compose_leg(args.seed, dev, oracle_utts=no)
        except Exception as e:                                    # a leg must never take the headline down
            legs["error"] = repr(e)
        out["legs"] = legs
        mark("legs")
    marks.append(("end", time.perf_counter()))
    print("bench.py wall-clock budget (rank 0, %d rank(s)): %s; total %.1f s" % (
        world, "; ".join("%s %.1f s" % (marks[i][0], marks[i][1] - marks[i - 1][1]) for i in range(1, len(marks) - 1)),
        marks[-1][1] - marks[0][1]), file=sys.stderr, flush=True)
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier(**bar_kw)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
