#!/usr/bin/env python
"""bench.py - frames/sec decoded on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path (dense GMM scoring + token-passing search)
over one batch of synthetic utterances per GPU: BASELINE.json configs[1]
(~1M-arc composed WFST, 3000 tied states x 16 mixtures, 64 utterances per GPU,
mainBeam 150).  Features are resident in HBM before the timed region.
N>1: one rank per GPU (torchrun), utterances sharded across ranks, no data-path
collective; ONE RCCL all_gather of the padded 1-best records per step.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--utts-per-gpu", type=int, default=64)
    ap.add_argument("--arcs", type=int, default=1_000_000)
    ap.add_argument("--beam", type=float, default=150.0)
    ap.add_argument("--max-hyps", type=int, default=0)
    ap.add_argument("--cpu-sample-utts", type=int, default=12)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from juicer_amd import build as jbuild
    from juicer_amd import capi, parallel, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(1, args.gpus) and world > 1:
        raise SystemExit("WORLD_SIZE %d != --gpus %d" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: juicer_amd has no CPU fallback")
    # JD_BENCH_SHARE_GPU=1 (development only): all ranks use GPU 0 over gloo, to exercise the
    # multi-rank code path on a one-GPU box; the reported numbers are meaningless then.
    share_gpu = os.environ.get("JD_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    if rank == 0:
        jbuild.build()
    bar_kw = {} if share_gpu else {"device_ids": [local_rank]}
    if world > 1:
        dist.barrier(**bar_kw)

    U = args.utts_per_gpu
    am, net, feats, _ = synth.config_c2(seed=args.seed, n_utts=U, target_arcs=args.arcs,
                                        utt_offset=rank * U)
    gnet = capi.Network.from_synth(net)
    gam = capi.Models.from_htk(am)
    dec = capi.Decoder(gnet, gam, main_beam=args.beam, max_hyps=args.max_hyps, device=local_rank,
                       max_streams=U)
    offs = np.zeros(U + 1, dtype=np.int64)
    offs[1:] = np.cumsum([f.shape[0] for f in feats])
    frames_local = int(offs[-1])
    d_feats = torch.from_numpy(np.concatenate(feats)).to(dev)     # inputs resident in HBM
    torch.cuda.synchronize()
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        hyps = dec.decode_batch_device(d_feats.data_ptr(), offs, stream)
        allh = parallel.gather_hyps(hyps, U, device=dev) if world > 1 else hyps
        return hyps, allh

    def barrier():
        if world > 1:
            dist.barrier(**bar_kw)

    for _ in range(args.warmup):
        step()
    barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    gmm_ms = search_ms = 0.0
    tm = {}
    gmm_launches = search_steps = ksamples = 0
    kernel_us = [0.0] * 6
    hyps = None
    for _ in range(args.steps):
        hyps, allh = step()
        tm = dec.last_timing()
        gmm_ms += tm["gmm_ms"]; search_ms += tm["search_ms"]
        gmm_launches += tm["gmm_launches"]; search_steps += tm["search_steps"]
        ksamples += tm["kernel_samples"]
        kernel_us = [a + b for a, b in zip(kernel_us, tm["kernel_us"])]
    torch.cuda.synchronize(); barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed, float(frames_local)], dtype=torch.float64, device=dev)
        tmax = t.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone(); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0]); frames_total = float(tsum[1])
    else:
        frames_total = float(frames_local)

    if rank != 0:
        if world > 1:
            dist.barrier(**bar_kw)                      # rank 0 finishes its CPU baseline first
            dist.destroy_process_group()
        return

    steps = args.steps
    fps = frames_total * steps / elapsed
    # ---- roofline of the dominant kernel.  Algorithmic bytes per stream-frame (SURVEY.md 8d,
    # DESIGN.md 5) from the decoder's own work counters; durations from HIP events recorded on
    # the decoder's streams inside the timed region (every 32nd lock-step frame is bracketed
    # kernel by kernel; the GMM kernel is bracketed on every launch).
    D, G, M, MN = am.D, am.n_gmm, am.max_mix, am.max_n
    st = {k: sum(h.stats[k] for h in hyps) for k in hyps[0].stats}       # one step's batch totals
    lock_steps = max(1, search_steps // steps)                            # launches of each search kernel per step
    gmm_l = max(1, gmm_launches // steps)
    names = [n for n in capi.kernel_names(tm) if n and n != "k_boundary"]   # the frame boundary runs inside k_resolve
    expand_name = "k_expand_closure" if tm.get("closure_inline") else "k_expand<0>"
    per_launch_bytes = {
        # token read + write (16-B tokens), arc->hmm lookup, likelihood gather
        "k_phase_a": ((32.0 * MN + 4.0) * st["tot_insts_in"] + 4.0 * st["tot_proc_emit_hyps"]) / lock_steps,
        # exit token + CSR bounds, arc records + slot map, Path records
        expand_name: (24.0 * st["tot_proc_end_hyps"] + 20.0 * st["tot_arcs_visited"] + 20.0 * st["tot_paths"]) / lock_steps,
        # destination entry-token read-modify-write
        "k_resolve": 32.0 * st["tot_arcs_visited"] / lock_steps,
        # parameters once per launch + features
        "jd_gmm_kernel": G * M * (2 * D + 1) * 4.0 + frames_local * D * 4.0 / gmm_l,
    }
    avg_us = {n: (kernel_us[i] / ksamples if ksamples else 0.0) for i, n in enumerate(capi.kernel_names(tm)) if n in names}
    avg_us["jd_gmm_kernel"] = 1e3 * gmm_ms / max(1, gmm_launches)
    # share of a step's GPU time: sampled average x launches per step
    tot_ms = {n: avg_us[n] * lock_steps / 1e3 for n in names}
    tot_ms["jd_gmm_kernel"] = gmm_ms / steps
    # dominant kernel of the critical path: the search kernels of the lock-step frames.  The GMM
    # kernel scores one chunk ahead on its own stream with a deliberately bounded grid (it is
    # throttled so that it never holds the search's wave slots) - its duration is not step time.
    dom = max((k for k in per_launch_bytes if k != "jd_gmm_kernel"), key=lambda k: tot_ms[k])
    achieved = per_launch_bytes[dom] / (avg_us[dom] * 1e-6) / 1e9 if avg_us[dom] > 0 else 0.0
    search_bytes = (32.0 * MN * st["tot_insts_in"] + 4.0 * st["tot_insts_in"] + 4.0 * st["tot_proc_emit_hyps"]
                    + 24.0 * st["tot_proc_end_hyps"] + 52.0 * st["tot_arcs_visited"] + 20.0 * st["tot_paths"])
    gmm_flops = frames_local * G * M * (3.0 * D + 4.0)
    # HBM traffic per launch of that kernel from the committed PMC passes (profiles/, same
    # workload; separate --pmc runs).  (2*FETCH_SIZE + WRITE_SIZE) KiB: gfx950 FETCH_SIZE reports
    # half of a wide read (MI355X_MICROARCH.md, HBM); only valid for the default workload.
    traffic = None
    try:
        default_cfg = (args.arcs == 1_000_000 and args.beam == 150.0 and args.max_hyps == 0 and U == 64 and args.seed == 0)
        pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_c2_pmc_summary.json")))
        key = {"k_phase_a": "k_phase_a<4>", "jd_gmm_kernel": "jd_gmm_kernel<39>"}.get(dom, dom)
        if default_cfg and key in pmc:
            traffic = round((2.0 * pmc[key]["FETCH_SIZE"]["mean"] + pmc[key]["WRITE_SIZE"]["mean"]) * 1024.0, 1)
    except Exception:
        traffic = None
    roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic,
                "algorithmic_bytes_per_launch": round(per_launch_bytes[dom], 1),
                "avg_launch_us": round(avg_us[dom], 3), "launches_per_step": lock_steps if dom != "jd_gmm_kernel" else gmm_l,
                "sampled_launches": ksamples if dom != "jd_gmm_kernel" else gmm_launches,
                "kernels_ms_per_step": {k: round(v, 3) for k, v in tot_ms.items()},
                "kernels_avg_us": {k: round(v, 2) for k, v in avg_us.items()},
                "search_all_kernels": {"algorithmic_GB_per_step": round(search_bytes / 1e9, 3),
                                       "ms_per_step": round(search_ms / steps, 3),
                                       "GBps": round(search_bytes / max(search_ms / steps, 1e-9) / 1e6, 1)},
                "gmm_background": {"valu_tflops": round(gmm_flops / max(tot_ms["jd_gmm_kernel"], 1e-9) / 1e9, 3),
                                   "ms_per_step": round(tot_ms["jd_gmm_kernel"], 3),
                                   "note": "scores one chunk ahead on its own stream, bounded grid; overlapped with the search"}}

    # ---- CPU baseline: the oracle (a port of the reference algorithm) on a bounded sample
    cpu = None
    if not args.no_cpu_baseline:
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
               "sample": "first %d utterances of rank 0's batch (%d frames), clock() CPU time around "
                         "init..finish as DecoderSingleTest.cpp:259-300; GPU 1-best identical on %d/%d"
                         % (ns, fr, same, ns)}

    out = {"metric": "frames/sec decoded", "value": round(fps, 1), "unit": "frames/s", "n_gpus": world,
           "steps": steps, "warmup": args.warmup, "ms_per_step": round(elapsed / steps * 1e3, 3),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
           "data": "synthetic", "xRT": round(fps / 100.0, 1),
           "config": {"workload": "configs[1]: %d-arc composed C.L.G, %d tied states x %d mix, D=%d, "
                                  "%d utterances per GPU, mainBeam %g, maxHyps %d"
                                  % (net.n_arcs, G, M, D, U, args.beam, args.max_hyps),
                      "frames_per_step": int(frames_total), "utts_per_gpu": U, "parallelism": "utterance-sharded x%d" % world},
           "roofline": roofline, "cpu_baseline": cpu}
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier(**bar_kw)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
