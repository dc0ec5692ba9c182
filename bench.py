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
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    if rank == 0:
        jbuild.build()
    if world > 1:
        dist.barrier(device_ids=[local_rank])

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
            dist.barrier(device_ids=[local_rank])

    for _ in range(args.warmup):
        step()
    barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    gmm_ms = search_ms = 0.0
    gmm_launches = search_launches = 0
    hyps = None
    for _ in range(args.steps):
        hyps, allh = step()
        tm = dec.last_timing()
        gmm_ms += tm["gmm_ms"]; search_ms += tm["search_ms"]
        gmm_launches += tm["gmm_launches"]; search_launches += tm["search_launches"]
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
            dist.destroy_process_group()
        return

    steps = args.steps
    fps = frames_total * steps / elapsed
    # ---- roofline of the dominant kernel (algorithmic bytes: SURVEY.md section 8d / DESIGN.md)
    D, G, M, MN = am.D, am.n_gmm, am.max_mix, am.max_n
    st = {k: sum(h.stats[k] for h in hyps) for k in hyps[0].stats}
    search_bytes = (32.0 * MN * st["tot_insts_in"] + 4.0 * st["tot_insts_in"] + 4.0 * st["tot_proc_emit_hyps"]
                    + 24.0 * st["tot_proc_end_hyps"] + 52.0 * st["tot_arcs_visited"] + 20.0 * st["tot_paths"])
    gmm_l = max(1, gmm_launches // steps)
    gmm_bytes = gmm_l * G * M * (2 * D + 1) * 4.0 + frames_local * D * 4.0
    kernels = {
        "jd_search_kernel": dict(bytes_per_launch=search_bytes / max(1, search_launches // steps),
                                 ms_per_launch=search_ms / max(1, search_launches), total_ms=search_ms / steps),
        "jd_gmm_kernel": dict(bytes_per_launch=gmm_bytes / gmm_l,
                              ms_per_launch=gmm_ms / max(1, gmm_launches), total_ms=gmm_ms / steps),
    }
    dom = max(kernels, key=lambda k: kernels[k]["total_ms"])
    kd = kernels[dom]
    achieved = kd["bytes_per_launch"] / (kd["ms_per_launch"] * 1e-3) / 1e9 if kd["ms_per_launch"] > 0 else 0.0
    gmm_flops = frames_local * G * M * (3.0 * D + 4.0)
    roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": None,
                "algorithmic_bytes_per_launch": round(kd["bytes_per_launch"], 1),
                "avg_launch_ms": round(kd["ms_per_launch"], 4),
                "kernels_ms_per_step": {k: round(v["total_ms"], 3) for k, v in kernels.items()},
                "gmm_valu_tflops": round(gmm_flops / max(kernels["jd_gmm_kernel"]["total_ms"], 1e-9) / 1e9, 3)}

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
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
