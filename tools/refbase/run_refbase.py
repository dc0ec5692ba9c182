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

    python tools/refbase/run_refbase.py [--utts 12] [--out profiles/cpu_reference_baseline.json]

Objects and the driver go to gpurun_out/refbase_build/ (scratch, not tracked, never sent to the GPU box).
"""
import argparse
import json
import os
import platform
import struct
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF = "/root/reference/src"
HERE = os.path.dirname(os.path.abspath(__file__))
BUILD = os.path.join(ROOT, "gpurun_out", "refbase_build")
TUS = ["WFSTDecoderLite", "WFSTDecoderLiteThreading", "WFSTNetwork", "WFSTLattice", "HTKFlatModels", "HTKFlatModelsThreading", "HTKModels",
       "Histogram", "BlockMemPool", "DecHypHistPool", "LogFile"]
# the reference's own definitions (src/CMakeLists.txt:3-5) and the oracle's compile discipline: -O2, no contraction
FLAGS = ["-O2", "-ffp-contract=off", "-fpermissive", "-w", "-DOPT_FLATMODEL", "-DOPT_SINGLE_BEST", "-DPARTIAL_DECODING", "-include", "time.h",
         "-I", os.path.join(HERE, "standins"), "-I", REF]


def build():
    os.makedirs(BUILD, exist_ok=True)
    objs = []
    for tu in TUS:
        o = os.path.join(BUILD, tu + ".o")
        subprocess.check_call(["g++"] + FLAGS + ["-c", os.path.join(REF, tu + ".cpp"), "-o", o])
        objs.append(o)
    for src in (os.path.join(HERE, "driver.cpp"), os.path.join(HERE, "standins", "htkparse_stub.cpp")):
        o = os.path.join(BUILD, os.path.basename(src)[:-4] + ".o")
        subprocess.check_call(["g++"] + FLAGS + ["-c", src, "-o", o])
        objs.append(o)
    exe = os.path.join(BUILD, "refbase_driver")
    subprocess.check_call(["g++", "-o", exe] + objs + ["-lpthread"])
    return exe


def run(exe, args, cpus, timeout):
    cmd = ["taskset", "-c", cpus, exe] + [str(a) for a in args]
    t0 = time.time()
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    try:
        so, se = p.communicate(timeout=timeout)
        rc = p.returncode
    except subprocess.TimeoutExpired:                                  # (a decoder that hangs: what it printed so far is kept)
        p.kill()
        so, se = p.communicate()
        rc = -999
    rows = [json.loads(l) for l in so.splitlines() if l.startswith("{") and l.rstrip().endswith("}")]
    return rc, rows, se[-600:], time.time() - t0


def same(row, o):
    """the reference's hypothesis against the oracle's: words, times, and every score bit for bit"""
    if row["n"] != o.n:
        return False
    if o.n <= 0:
        return True
    f32 = lambda a: np.asarray(a, np.float32).view(np.uint32)
    return (list(row["label"]) == list(o.label) and list(row["time"]) == list(o.time) and np.array_equal(f32(row["score"]), f32(o.score))
            and np.array_equal(f32(row["ac"]), f32(o.ac)) and np.array_equal(f32(row["lm"]), f32(o.lm)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--utts", type=int, default=12)
    ap.add_argument("--beam", type=float, default=150.0)
    ap.add_argument("--max-hyps", type=int, default=0)
    ap.add_argument("--arcs", type=int, default=1_000_000)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "cpu_reference_baseline.json"))
    args = ap.parse_args()
    if not os.path.isdir(REF):
        raise SystemExit("tools/refbase runs in the build container only: %s is not there" % REF)
    from juicer_amd import build as jbuild
    from juicer_amd import capi, synth
    from oracle import oracle as orc
    from oracle.oracle import OracleAM, OracleDecoder, OracleNet
    jbuild.build(); orc.build()
    exe = build()
    am, net, feats, _ = synth.config_c2(seed=0, n_utts=args.utts, target_arcs=args.arcs)
    jmbi, jwnt, featf = (os.path.join(BUILD, n) for n in ("models.jmbi", "net.jwnt", "feats.bin"))
    capi.Models.from_htk(am).save_jmbi(jmbi)
    capi.Network.from_synth(net).save_jwnt(jwnt)
    with open(featf, "wb") as f:
        f.write(struct.pack("<ii", len(feats), am.D))
        for x in feats:
            f.write(struct.pack("<i", x.shape[0]))
            f.write(np.ascontiguousarray(x, np.float32).tobytes())
    od = OracleDecoder(OracleNet(net), OracleAM(am), main_beam=args.beam, max_hyps=args.max_hyps)
    want, port_s = [], 0.0
    for x in feats:
        o = od.decode(x)
        want.append(o); port_s += o.cpu_seconds
    frames = int(sum(x.shape[0] for x in feats))
    common = [args.beam, 0.0, 0.0, 0.0, args.max_hyps, 1.0, 0.0]      # mainBeam, start / end / word beams off, lmScale 1, no penalty
    out = {"what": "BASELINE.md B1 / B2: the reference's own WFSTDecoderLite (one thread) and WFSTDecoderLiteThreading (search thread + "
                   "scoring thread) on configs[1], compiled from /root/reference/src against stand-ins for Torch3 / Tracter headers "
                   "(tools/refbase) - a timing and differential aid, NOT a reference build: it pins nothing",
           "workload": "configs[1]: %d-arc composed C.L.G, %d tied states x %d mix, first %d utterances of seed 0 (%d frames), mainBeam %g, "
                       "maxHyps %d, blockSize 5" % (net.n_arcs, am.n_gmm, am.max_mix, len(feats), frames, args.beam, args.max_hyps),
           "timing": "clock() around init .. finish per utterance, 20 frames of look-ahead (DecoderSingleTest.cpp:259-307; halved for the "
                     "two-thread decoder as at :303-307); taskset to one / two cores",
           "compile": "g++ " + " ".join(f for f in FLAGS if not f.startswith("/") and f != "-I"),
           "host": {"cpu": platform.processor() or platform.machine(), "cores": os.cpu_count()},
           "oracle_port": {"frames_per_s": round(frames / port_s, 1), "cores": 1}}
    try:
        out["host"]["model"] = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        pass
    rc, rows, err, wall = run(exe, [jmbi, jwnt, featf, 0] + common, "2", 3600)
    if rc == 0 and len(rows) == len(feats):
        secs = sum(r["cpu_s"] for r in rows)
        out["WFSTDecoderLite"] = {"frames_per_s": round(frames / secs, 1), "xRT": round(frames / secs / 100.0, 2), "cores": 1, "cpu_seconds": round(secs, 3),
                                  "identical_to_oracle": sum(int(same(r, o)) for r, o in zip(rows, want)), "utterances": len(rows),
                                  "hyps_found": sum(int(r["n"] > 0) for r in rows)}
    else:
        out["WFSTDecoderLite"] = {"error": "exit code %d, %d of %d utterances" % (rc, len(rows), len(feats)), "stderr": err}
    # B2: SURVEY.md 8d reports this one as aborting on assert(fQueue[fEnd].next == -1) (HTKFlatModelsThreading.cpp:108): attempt, report
    rc, rows, err, wall = run(exe, [jmbi, jwnt, featf, 1] + common, "2,3", 240)
    if rc == 0 and len(rows) == len(feats):
        secs = sum(r["cpu_s"] for r in rows)
        out["WFSTDecoderLiteThreading"] = {"frames_per_s": round(frames / secs, 1), "xRT": round(frames / secs / 100.0, 2), "cores": 2,
                                           "cpu_seconds_halved": round(secs, 3), "wall_seconds": round(wall, 2),
                                           "identical_to_oracle": sum(int(same(r, o)) for r, o in zip(rows, want)), "utterances": len(rows)}
    else:
        out["WFSTDecoderLiteThreading"] = {"error": ("no result within %.0f s (killed)" % wall if rc == -999 else "exit code %d" % rc) + " after %d of %d utterances" % (len(rows), len(feats)),
                                           "stderr": err,
                                           "note": "the reference's unsynchronised request queue (HTKFlatModelsThreading.cpp:100-133): SURVEY.md 8d saw the same; "
                                                   "the oracle's restatement of the two-thread organisation (jo_decode_utt_threading, a C11-atomics ring) is "
                                                   "what bench.py times as cpu_baseline.two_thread_core"}
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
