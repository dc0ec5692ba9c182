"""The differential check of the CPU oracle against the reference's OWN compiled classes (build container only).

tools/refbase compiles the hot-path translation units of /root/reference/src where they lie - nothing of the reference is
copied into this repository - against stand-ins for the third-party headers the image lacks (tools/refbase/standins) and
drives them with tools/refbase/driver.cpp.  A build against stand-ins is NOT a reference build: it pins nothing, `parity`
stays "partial" (DESIGN.md 2).  What it gives is a differential: the oracle (oracle/juicer_oracle.c, the restatement every
GPU parity test is held against) and the reference's classes decode the same models, the same network and the same
utterances and must agree on every word, time, score (bit for bit), on the reference's five statistics and on the partial
paths PARTIAL_DECODING recovers.

Used by tests/test_refdiff_cpu.py (every code path the GPU tests use, at sizes that take seconds) and by
tools/refbase/run_refbase.py (the bench workloads at their sizes -> profiles/cpu_reference_baseline.json).
"""
import json
import os
import struct
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
REF = "/root/reference/src"
HERE = os.path.dirname(os.path.abspath(__file__))
BUILD = os.path.join(ROOT, "gpurun_out", "refbase_build")          # scratch: not tracked, never sent to the GPU box
TUS = ["WFSTDecoderLite", "WFSTDecoderLiteThreading", "WFSTNetwork", "WFSTLattice", "HTKFlatModels", "HTKFlatModelsThreading", "HTKModels",
       "Histogram", "BlockMemPool", "DecHypHistPool", "LogFile"]
# the reference's own definitions (src/CMakeLists.txt:3-5) and the oracle's compile discipline: -O2, no contraction
FLAGS = ["-O2", "-ffp-contract=off", "-fpermissive", "-w", "-DOPT_FLATMODEL", "-DOPT_SINGLE_BEST", "-DPARTIAL_DECODING", "-include", "time.h",
         "-I", os.path.join(HERE, "standins"), "-I", REF]
REF_STATS = ("tot_active_emit_hyps", "tot_active_end_hyps", "tot_active_models", "tot_proc_emit_hyps", "tot_proc_end_hyps")


def available() -> bool:
    return os.path.isdir(REF)


def build() -> str:
    """g++ on the reference's sources where they lie + this build's driver and stand-ins -> gpurun_out/refbase_build/refbase_driver"""
    os.makedirs(BUILD, exist_ok=True)
    exe = os.path.join(BUILD, "refbase_driver")
    own = [os.path.join(HERE, "driver.cpp"), os.path.join(HERE, "standins", "htkparse_stub.cpp")]
    deps = own + [os.path.join(HERE, "standins", h) for h in ("general.h", "log_add.h", "TracterObject.h")] + [os.path.abspath(__file__)]
    if os.path.exists(exe) and all(os.path.getmtime(p) <= os.path.getmtime(exe) for p in deps):
        return exe
    objs, jobs = [], []
    for src in [os.path.join(REF, tu + ".cpp") for tu in TUS] + own:
        o = os.path.join(BUILD, os.path.basename(src).rsplit(".", 1)[0] + ".o")
        jobs.append(subprocess.Popen(["g++"] + FLAGS + ["-c", src, "-o", o]))
        objs.append(o)
    for j in jobs:
        if j.wait() != 0:
            raise RuntimeError("tools/refbase: compiling the reference's translation units failed")
    subprocess.check_call(["g++", "-o", exe] + objs + ["-lpthread"])
    return exe


def write_feats(path, feats, D):
    with open(path, "wb") as f:
        f.write(struct.pack("<ii", len(feats), D))
        for x in feats:
            f.write(struct.pack("<i", x.shape[0]))
            f.write(np.ascontiguousarray(x, np.float32).tobytes())


def write_symbols(path, prefix, n):
    """an AT&T symbol table for labels 0 .. n (WFSTAlphabet, src/WFSTNetwork.cpp:48-112: "name id" lines; no '#' names: nothing is auxiliary)"""
    with open(path, "w") as f:
        f.write("<eps> 0\n")
        for i in range(1, n + 1):
            f.write("%s%d %d\n" % (prefix, i, i))


def run_reference(am, net, feats, beams=None, loader="jwnt", lm_scale=1.0, ins_penalty=0.0, pti=0, threading=False, cpus=None,
                  timeout=3600, workdir=None):
    """The reference's decoder on (am, net, feats): one dict per utterance (the driver's JSON lines), plus (rc, stderr tail, wall).
    loader: "jwnt" - the network through WFSTNetwork::readBinary from the file this build's jd_net_save_jwnt writes, read with scale 1
    and penalty 0: ONLY for lm_scale == 1 and ins_penalty == 0 (the file holds the weights with the scale divided out, as the
    reference's writeBinary leaves them, src/WFSTNetwork.cpp:1106-1125: another scale is the FSM route's business; the JWNT round
    trip with a scale is held to the reference in tests/test_refdiff_cpu.py::test_network_loaded_by_the_reference) - or "fsm": the TEXT constructor
    (src/WFSTNetwork.cpp:371-616) from an AT&T text file + two symbol tables, the reference applying scale and penalty itself.
    Models always through HTKModels::readBinary from this build's jd_am_save_jmbi (the MMF text parser is bison / flex output)."""
    from juicer_amd import capi
    from juicer_amd import io as jio
    assert loader != "jwnt" or (lm_scale == 1.0 and ins_penalty == 0.0), "a scale / penalty goes through the FSM text route (see above)"
    exe = build()
    beams = dict(beams or {})
    tmp = workdir or tempfile.mkdtemp(prefix="refdiff_", dir=BUILD)
    os.makedirs(tmp, exist_ok=True)
    jmbi, featf = os.path.join(tmp, "models.jmbi"), os.path.join(tmp, "feats.bin")
    if hasattr(am, "priors"):                                          # hybrid ANN / HMM models (synth.HybridAM): a phone list + a priors file
        phones, priors = os.path.join(tmp, "phones.lst"), os.path.join(tmp, "priors.txt")
        with open(phones, "w") as f:
            f.write("".join("p%d\n" % i for i in range(len(am.priors))))
        with open(priors, "w") as f:
            f.write("".join("%.9g\n" % float(v) for v in am.priors))
        margs = ["phones=" + phones, "priors=" + priors, "spm=%d" % am.states_per_model]
        D = len(am.priors)
    else:
        capi.Models.from_htk(am).save_jmbi(jmbi)
        margs = ["models=" + jmbi]
        D = am.D
    write_feats(featf, feats, D)
    args = margs + ["feats=" + featf, "threading=%d" % int(bool(threading)), "main=%.9g" % beams.get("main_beam", 0.0),
            "start=%.9g" % beams.get("start_beam", 0.0), "end=%.9g" % beams.get("end_beam", 0.0), "word=%.9g" % beams.get("word_beam", 0.0),
            "maxhyps=%d" % beams.get("max_hyps", 0), "pti=%d" % pti]
    if loader == "fsm":
        fsm, ins, outs = (os.path.join(tmp, n) for n in ("net.fsm", "in.syms", "out.syms"))
        jio.write_fsm(fsm, net)
        write_symbols(ins, "m", int(len(am.priors) if hasattr(am, "priors") else am.n_hmm))
        write_symbols(outs, "w", int(max(1, np.max(net.olab) if net.n_arcs else 1)))
        args += ["fsm=" + fsm, "insyms=" + ins, "outsyms=" + outs, "lmscale=%.9g" % lm_scale, "inspen=%.9g" % ins_penalty]
    else:
        jwnt = os.path.join(tmp, "net.jwnt")
        capi.Network.from_synth(net, lm_scale, ins_penalty).save_jwnt(jwnt)
        args += ["net=" + jwnt, "lmscale=1", "inspen=0"]
    cmd = ([] if cpus is None else ["taskset", "-c", cpus]) + [exe] + args
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
    return rows, (rc, se[-600:], time.time() - t0)


def same_hyp(row, o) -> bool:
    """the reference's hypothesis against the oracle's: words, times, and every score bit for bit"""
    if row["n"] != o.n:
        return False
    if o.n <= 0:
        return True
    f32 = lambda a: np.asarray(a, np.float32).view(np.uint32)
    return bool(list(row["label"]) == list(o.label) and list(row["time"]) == list(o.time) and np.array_equal(f32(row["score"]), f32(o.score))
                and np.array_equal(f32(row["ac"]), f32(o.ac)) and np.array_equal(f32(row["lm"]), f32(o.lm))
                and np.array_equal(f32(row["tot"]), f32([o.tot_score, o.tot_ac, o.tot_lm])))


def same_stats(row, o) -> bool:
    """the reference's five statistics (WFSTDecoderLite.cpp:231-241: its protected totals) against the oracle's"""
    return all(int(row["stats"][k]) == int(o.stats[k]) for k in REF_STATS)


def diff_case(name, am, net, feats, beams=None, loader="jwnt", lm_scale=1.0, ins_penalty=0.0, pti=0, threading=False, cpus=None,
              timeout=3600):
    """One differential case: the reference's classes and the oracle on the same inputs.  Returns a summary dict
    {name, utterances, identical_hyps, identical_stats, identical_partial (pti > 0), hyps_found, ref_cpu_seconds,
    oracle_cpu_seconds, frames, ok} - ok = everything identical on every utterance."""
    from oracle.oracle import OracleAM, OracleDecoder, OracleNet
    rows, (rc, err, wall) = run_reference(am, net, feats, beams, loader, lm_scale, ins_penalty, pti, threading, cpus, timeout)
    oam = OracleAM.from_hybrid(am.priors, am.states_per_model) if hasattr(am, "priors") else OracleAM(am)
    od = OracleDecoder(OracleNet(net, lm_scale, ins_penalty), oam, **(beams or {}))
    out = {"name": name, "loader": loader, "beams": dict(beams or {}), "utterances": len(feats), "frames": int(sum(x.shape[0] for x in feats)),
           "threading": bool(threading)}
    if lm_scale != 1.0 or ins_penalty != 0.0:
        out["lm_scale"], out["ins_penalty"] = lm_scale, ins_penalty
    if rc != 0 or len(rows) != len(feats):
        out.update(ok=False, error=("no result within %.0f s (killed)" % wall if rc == -999 else "exit code %d" % rc)
                   + " after %d of %d utterances" % (len(rows), len(feats)), stderr=err)
        if rc in (-11, -6):                                            # SIGSEGV / abort INSIDE the reference's classes: nothing to compare
            out["reference_crashed"] = True
        return out
    hyp_ok = st_ok = part_ok = found = 0
    osec = 0.0
    for r, x in zip(rows, feats):
        if pti > 0:
            _, final = od.decode_partial(x, interval=pti)
            o = od.decode(x)
            part_ok += int([list(p) for p in final] == [list(p) for p in r["partial"]])
        else:
            o = od.decode(x)
        osec += o.cpu_seconds
        hyp_ok += int(same_hyp(r, o)); st_ok += int(same_stats(r, o)); found += int(o.n > 0)
    out.update(identical_hyps=hyp_ok, identical_stats=st_ok, hyps_found=found, ref_cpu_seconds=round(sum(r["cpu_s"] for r in rows), 3),
               oracle_cpu_seconds=round(osec, 3))
    if pti > 0:
        out.update(partial_interval=pti, identical_partial=part_ok)
    out["ok"] = hyp_ok == len(feats) and st_ok == len(feats) and (pti == 0 or part_ok == len(feats))
    return out


def reference_model_tables(am, workdir=None):
    """What the REFERENCE's own code makes of the models: HTKModels::readBinary on the file this build's jd_am_save_jmbi writes, then
    HTKFlatModels::init (src/HTKFlatModels.cpp:94-177) - the flat tables it scores with (det = gconst + log weight, means, INVERSE
    variances) - and its IModels view of every HMM (log transition matrix, SEIndex, tee log-probability).  Returns dict(det [G][M],
    mean [G][M][D], ivar [G][M][D], n_mix [G], hmm_n [H], tee [H], trans: list of [n][n] arrays, se: list of [n][2])."""
    from juicer_amd import capi
    exe = build()
    tmp = workdir or tempfile.mkdtemp(prefix="refmodels_", dir=BUILD)
    os.makedirs(tmp, exist_ok=True)
    jmbi, dump = os.path.join(tmp, "models.jmbi"), os.path.join(tmp, "models.dump")
    capi.Models.from_htk(am).save_jmbi(jmbi)
    subprocess.check_call([exe, "models=" + jmbi, "dumpmodels=" + dump], stdout=subprocess.DEVNULL)
    raw = open(dump, "rb").read()
    G, M, D, H = struct.unpack_from("<4i", raw, 0)
    o = 16
    det = np.zeros((G, M), np.float32); mean = np.zeros((G, M, D), np.float32); ivar = np.zeros((G, M, D), np.float32)
    n_mix = np.zeros(G, np.int32)
    for g in range(G):
        n_mix[g] = struct.unpack_from("<i", raw, o)[0]; o += 4
        det[g] = np.frombuffer(raw, np.float32, M, o); o += 4 * M
        mean[g] = np.frombuffer(raw, np.float32, M * D, o).reshape(M, D); o += 4 * M * D
        ivar[g] = np.frombuffer(raw, np.float32, M * D, o).reshape(M, D); o += 4 * M * D
    hmm_n = np.zeros(H, np.int32); tee = np.zeros(H, np.float32); trans, se = [], []
    for h in range(H):
        hmm_n[h] = struct.unpack_from("<i", raw, o)[0]; o += 4
        tee[h] = struct.unpack_from("<f", raw, o)[0]; o += 4
        n = int(hmm_n[h])
        trans.append(np.frombuffer(raw, np.float32, n * n, o).reshape(n, n).copy()); o += 4 * n * n
        se.append(np.frombuffer(raw, np.int16, 2 * n, o).reshape(n, 2).copy()); o += 4 * n
    assert o == len(raw), (o, len(raw))
    return dict(det=det, mean=mean, ivar=ivar, n_mix=n_mix, hmm_n=hmm_n, tee=tee, trans=trans, se=se)


def reference_log_likelihoods(am, feats0, n_frames=32, workdir=None):
    """The reference's own HTKFlatModels::calcOutput (calcGMMOutput + logAdd, src/HTKFlatModels.cpp:202-293) for every tied state of the
    first n_frames frames of one utterance: float32 [frames][n_gmm]."""
    from juicer_amd import capi
    exe = build()
    tmp = workdir or tempfile.mkdtemp(prefix="refll_", dir=BUILD)
    os.makedirs(tmp, exist_ok=True)
    jmbi, featf, dump = (os.path.join(tmp, n) for n in ("models.jmbi", "feats.bin", "ll.dump"))
    capi.Models.from_htk(am).save_jmbi(jmbi)
    write_feats(featf, [feats0], am.D)
    subprocess.check_call([exe, "models=" + jmbi, "feats=" + featf, "dumpll=" + dump, "llframes=%d" % n_frames], stdout=subprocess.DEVNULL)
    raw = open(dump, "rb").read()
    nf, G = struct.unpack_from("<2i", raw, 0)
    return np.frombuffer(raw, np.float32, nf * G, 8).reshape(nf, G).copy()


def reference_network_tables(am, net, loader="fsm", lm_scale=1.0, ins_penalty=0.0, workdir=None):
    """What the REFERENCE's own loader makes of a network - the TEXT constructor (src/WFSTNetwork.cpp:371-616: weights negated, scaled,
    the insertion penalty added by the reference itself) or readBinary of the JWNT file this build writes: dict(n_states, init, row_ptr,
    to, ilab, olab, w, fin_w (NaN: not final)), arcs in the order the reference's getTransitions walks them.  Also returns the files'
    paths (fsm, insyms, outsyms or jwnt) so that the product's loaders can be given the very same files."""
    from juicer_amd import capi
    from juicer_amd import io as jio
    exe = build()
    tmp = workdir or tempfile.mkdtemp(prefix="refnet_", dir=BUILD)
    os.makedirs(tmp, exist_ok=True)
    jmbi, dump = os.path.join(tmp, "models.jmbi"), os.path.join(tmp, "net.dump")
    capi.Models.from_htk(am).save_jmbi(jmbi)
    if loader == "fsm":
        files = tuple(os.path.join(tmp, n) for n in ("net.fsm", "in.syms", "out.syms"))
        jio.write_fsm(files[0], net)
        write_symbols(files[1], "m", int(am.n_hmm))
        write_symbols(files[2], "w", int(max(1, np.max(net.olab) if net.n_arcs else 1)))
        args = ["fsm=" + files[0], "insyms=" + files[1], "outsyms=" + files[2], "lmscale=%.9g" % lm_scale, "inspen=%.9g" % ins_penalty]
    else:
        files = (os.path.join(tmp, "net.jwnt"),)
        # (writeBinary takes the penalty and the scale out again before it writes, src/WFSTNetwork.cpp:1106-1125; readBinary puts back
        # the ones its constructor was given: the file holds unscaled weights whatever the writer's setting)
        capi.Network.from_synth(net, lm_scale, ins_penalty).save_jwnt(files[0])
        args = ["net=" + files[0], "lmscale=%.9g" % lm_scale, "inspen=%.9g" % ins_penalty]
    subprocess.check_call([exe, "models=" + jmbi, "dumpnet=" + dump] + args, stdout=subprocess.DEVNULL)
    raw = open(dump, "rb").read()
    ns, na, init = struct.unpack_from("<3i", raw, 0)
    o = 12
    row_ptr = np.zeros(ns + 1, np.int32); fin_w = np.zeros(ns, np.float32)
    to = np.zeros(na, np.int32); il = np.zeros(na, np.int32); ol = np.zeros(na, np.int32); w = np.zeros(na, np.float32)
    k = 0
    for q in range(ns):
        nt = struct.unpack_from("<i", raw, o)[0]; fin_w[q] = struct.unpack_from("<f", raw, o + 4)[0]; o += 8
        rec = np.frombuffer(raw, np.int32, 4 * nt, o).reshape(nt, 4); o += 16 * nt
        to[k:k + nt] = rec[:, 0]; il[k:k + nt] = rec[:, 1]; ol[k:k + nt] = rec[:, 2]; w[k:k + nt] = rec[:, 3].copy().view(np.float32)
        k += nt; row_ptr[q + 1] = k
    assert k == na and o == len(raw), (k, na, o, len(raw))
    return dict(n_states=ns, init=init, row_ptr=row_ptr, to=to, ilab=il, olab=ol, w=w, fin_w=fin_w), files
