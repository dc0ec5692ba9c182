#!/usr/bin/env python
"""the reference's OWN WFSTDecoderLite and WFSTDecoderLiteThreading, timed on this build's synthetic
configs[1] workload the way DecoderSingleTest.cpp:259-307 times a decode - and compared, hypothesis by hypothesis, with the
CPU oracle (oracle/juicer_oracle.c: the restatement every parity test is held against).

Run in the BUILD container only: it compiles /root/reference/src/*.cpp where they lie (nothing of the reference is copied into
this repository) against the stand-ins of tools/refbase/standins for the third-party headers the image lacks (Torch3
general.h / log_add.h, TracterObject.h) and a stub for the bison / flex generated MMF parser.  Such a build is NOT a reference
build: it pins nothing, `parity` stays "partial" (DESIGN.md 2).  What it gives: the reference CPU paths' own frames/s on this
host, and a differential check of the restatement - the oracle and the reference's classes decode the same models (loaded
through the reference's own binary readers from files this build's jd_am_save_jmbi / jd_net_save_jwnt write), the same
utterances, and must produce the same words, times and scores bit for bit.

    python tools/refbase/run_refbase.py [--utts 64] [--quick] [--out profiles/cpu_reference_baseline.json]

The machinery (compile recipe, driver invocation, comparisons) is tools/refbase/refdiff.py, shared with tests/test_refdiff_cpu.py.

Objects and the driver go to gpurun_out/refbase_build/ (scratch, not tracked, never sent to the GPU box).
"""
import argparse
import json
import os
import platform
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import refdiff  # noqa: E402

BEAMS = [dict(), dict(main_beam=200.0), dict(main_beam=150.0, end_beam=100.0, word_beam=80.0, start_beam=120.0), dict(main_beam=150.0, max_hyps=200),
         dict(max_hyps=300), dict(main_beam=120.0, end_beam=90.0, word_beam=70.0, start_beam=100.0, max_hyps=150)]     # tests/test_gpu_parity.py


def random_cases():
    """The graphs of tests/test_gpu_random_topology.py (what the HIP path is held to the oracle on): 48 small ones, 10 of 300-3000 states."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_gpu_random_topology as trt
    cases = []
    for seed in range(7000, 7048):
        am, net, feats, kw = trt._case(seed)
        cases.append(refdiff.diff_case("random topology %d (%d states, %d arcs)" % (seed, net.n_states, net.n_arcs), am, net, feats, kw,
                                       loader=("fsm" if seed % 2 else "jwnt")))
    for seed in range(8000, 8010):
        am, net, feats, kw, lm, pen = trt._big_case(seed)
        neutral = lm == 1.0 and pen == 0.0
        cases.append(refdiff.diff_case("random topology %d (%d states, %d arcs)" % (seed, net.n_states, net.n_arcs), am, net, feats, kw,
                                       loader=("jwnt" if neutral and seed % 2 == 0 else "fsm"), lm_scale=lm, ins_penalty=pen))
    return cases


def summarise(cases):
    return {"cases": cases, "cases_total": len(cases), "cases_identical": sum(int(c["ok"]) for c in cases),
            "cases_reference_crashed": sum(int(bool(c.get("reference_crashed"))) for c in cases),
            "cases_different": sum(int(not c["ok"] and not c.get("reference_crashed")) for c in cases),
            "utterances_total": sum(c["utterances"] for c in cases),
            "what_identical_means": "every utterance: words, times, every score and the totals bit for bit; the reference's five statistics "
                                    "(its protected totals, WFSTDecoderLite.h:150-154); with PartialTraceInterval the partial paths"}


def append_random(path):
    from juicer_amd import build as jbuild
    from oracle import oracle as orc
    jbuild.build(); orc.build(); refdiff.build()
    out = json.load(open(path))
    old = [c for c in out["differential"]["cases"] if not c["name"].startswith("random topology")]
    new = random_cases()
    out["differential"] = summarise(old + new)
    out["differential"]["random_topologies"] = {"cases": len(new), "identical": sum(int(c["ok"]) for c in new), "utterances": sum(c["utterances"] for c in new),
                                                "hyps_found": sum(c.get("hyps_found", 0) for c in new), "appended": "python tools/refbase/run_refbase.py --append-random"}
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
    print(json.dumps({k: v for k, v in out["differential"].items() if k != "cases"}, indent=1))
    print("not identical:", [c["name"] for c in new if not c["ok"]])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--utts", type=int, default=64, help="configs[1] utterances the two reference decoders are timed on (the whole batch)")
    ap.add_argument("--beam", type=float, default=150.0)
    ap.add_argument("--arcs", type=int, default=1_000_000)
    ap.add_argument("--quick", action="store_true", help="timing legs only (B1 / B2), no wider differential")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "cpu_reference_baseline.json"))
    ap.add_argument("--append-random", action="store_true",
                    help="only add the random graphs of arbitrary shape (tests/random_topology.py) to the differential of the existing --out file; "
                         "the timing legs and the other cases stay as they were measured")
    args = ap.parse_args()
    if not refdiff.available():
        raise SystemExit("tools/refbase runs in the build container only: %s is not there" % refdiff.REF)
    if args.append_random:
        return append_random(args.out)
    from juicer_amd import build as jbuild
    from juicer_amd import synth
    from oracle import oracle as orc
    jbuild.build(); orc.build()
    refdiff.build()
    t_all = time.time()
    am, net, feats, _ = synth.config_c2(seed=0, n_utts=args.utts, target_arcs=args.arcs)
    frames = int(sum(x.shape[0] for x in feats))
    out = {"what": "BASELINE.md B1 / B2: the reference's own WFSTDecoderLite (one thread) and WFSTDecoderLiteThreading (search thread + "
                   "scoring thread) on configs[1], compiled from /root/reference/src against stand-ins for Torch3 / Tracter headers "
                   "(tools/refbase) - a timing and differential aid, NOT a reference build: it pins nothing.  `differential`: the CPU oracle "
                   "(oracle/juicer_oracle.c) against the same classes on every code path the GPU parity tests use",
           "workload": "configs[1]: %d-arc composed C.L.G, %d tied states x %d mix, %d utterances of seed 0 (%d frames), mainBeam %g, "
                       "maxHyps 0, blockSize 5" % (net.n_arcs, am.n_gmm, am.max_mix, len(feats), frames, args.beam),
           "timing": "clock() around init .. finish per utterance, 20 frames of look-ahead (DecoderSingleTest.cpp:259-307; halved for the "
                     "two-thread decoder as at :303-307); taskset to one / two cores",
           "compile": "g++ " + " ".join(f for f in refdiff.FLAGS if not f.startswith("/") and f != "-I"),
           "host": {"cpu": platform.processor() or platform.machine(), "cores": os.cpu_count()}}
    try:
        out["host"]["model"] = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        pass
    beams = dict(main_beam=args.beam)
    b1 = refdiff.diff_case("configs[1], WFSTDecoderLite", am, net, feats, beams, cpus="2")
    out["oracle_port"] = {"frames_per_s": round(frames / max(b1.get("oracle_cpu_seconds", 0.0), 1e-9), 1), "cores": 1}
    if b1.get("error"):
        out["WFSTDecoderLite"] = b1
    else:
        out["WFSTDecoderLite"] = {"frames_per_s": round(frames / b1["ref_cpu_seconds"], 1), "xRT": round(frames / b1["ref_cpu_seconds"] / 100.0, 2), "cores": 1,
                                  "cpu_seconds": b1["ref_cpu_seconds"], "identical_to_oracle": b1["identical_hyps"], "identical_statistics": b1["identical_stats"],
                                  "utterances": b1["utterances"], "hyps_found": b1["hyps_found"]}
    # B2: SURVEY.md 8d reports this one as aborting on assert(fQueue[fEnd].next == -1) (HTKFlatModelsThreading.cpp:108): attempt, report
    t0 = time.time()
    b2 = refdiff.diff_case("configs[1], WFSTDecoderLiteThreading", am, net, feats, beams, threading=True, cpus="2,3", timeout=900)
    if b2.get("error"):
        b2["note"] = ("the reference's unsynchronised request queue (HTKFlatModelsThreading.cpp:100-133): SURVEY.md 8d saw the same; the oracle's "
                      "restatement of the two-thread organisation (jo_decode_utt_threading, a C11-atomics ring) is what bench.py times as "
                      "cpu_baseline.two_thread_core")
        out["WFSTDecoderLiteThreading"] = b2
    else:
        out["WFSTDecoderLiteThreading"] = {"frames_per_s": round(frames / b2["ref_cpu_seconds"], 1), "xRT": round(frames / b2["ref_cpu_seconds"] / 100.0, 2),
                                           "cores": 2, "cpu_seconds_halved": b2["ref_cpu_seconds"], "wall_seconds": round(time.time() - t0, 2),
                                           "identical_to_oracle": b2["identical_hyps"], "identical_statistics": b2["identical_stats"],
                                           "utterances": b2["utterances"]}
    if not args.quick:
        # ---- the wider differential (verdict r5, item 3): every code path the GPU tests use, the bench workloads at their sizes
        cases = []
        add = lambda *a, **k: cases.append(refdiff.diff_case(*a, **k))
        add("configs[1] + maxHyps 6000 (Histogram)", am, net, feats[:16], dict(main_beam=args.beam, max_hyps=6000))
        add("configs[1], PartialTraceInterval 30", am, net, feats[:8], dict(main_beam=args.beam), pti=30)
        add("configs[1], end / word / start beams + PartialTraceInterval 30", am, net, feats[:8],
            dict(main_beam=args.beam, end_beam=100.0, word_beam=80.0, start_beam=120.0), pti=30)
        add("configs[1] through the FSM TEXT constructor (WFSTNetwork.cpp:371-616)", am, net, feats[:8], dict(main_beam=args.beam), loader="fsm")
        add("configs[1] through the FSM TEXT constructor, lmScale 8, insPenalty -0.5", am, net, feats[:8], dict(main_beam=args.beam), loader="fsm",
            lm_scale=8.0, ins_penalty=-0.5)
        del am, net, feats
        a6, n6, f6, _ = synth.config_c2_mixed(seed=0, n_utts=4, target_arcs=args.arcs)
        add("configs[1]'s graph with HMMs of 1-6 emitting states", a6, n6, f6, dict(main_beam=args.beam))
        del a6, n6, f6
        a4, n4, f4, _ = synth.config_c4(seed=0, n_utts=2, n_words=5000, n_tri_hist=40_000)
        short = [synth.sample_utterance_walk(5000 + u, n4, a4, 2)[0] for u in range(3)]
        add("config_c4 (trigram-shaped, epsilon back-off, %d arcs), beam 300, short utterances" % n4.n_arcs, a4, n4, short, dict(main_beam=300.0))
        add("config_c4 (%d arcs), beam 200" % n4.n_arcs, a4, n4, f4[:2], dict(main_beam=200.0))
        del a4, n4, f4
        for nm, cfg in (("config_small (tee sp)", synth.config_small), ("config_mixed (1-6 emitting states, tee)", synth.config_mixed)):
            a, n, f, _ = cfg(n_utts=4)
            for kw in BEAMS:
                for loader in ("jwnt", "fsm"):
                    add("%s %s" % (nm, kw or "no pruning"), a, n, f, kw, loader=loader)
        a, n, f, _ = synth.config_toy()
        for kw in BEAMS:
            add("config_toy %s" % (kw or "no pruning"), a, n, f, kw)
        cases += random_cases()
        out["differential"] = summarise(cases)
    out["wall_seconds_all"] = round(time.time() - t_all, 1)
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
    brief = {k: v for k, v in out.items() if k != "differential"}
    if "differential" in out:
        brief["differential"] = {k: v for k, v in out["differential"].items() if k != "cases"}
        brief["differential"]["not_identical"] = [c for c in out["differential"]["cases"] if not c["ok"]]
    print(json.dumps(brief, indent=1))


if __name__ == "__main__":
    main()
