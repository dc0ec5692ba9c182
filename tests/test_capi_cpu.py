"""CPU-side tests of the C ABI library: it loads, exports every declared symbol, the host
preparation arithmetic equals the oracle's, and compute entry points fail loudly without a
GPU (no fallback).  No kernel is launched here."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_exports_every_declared_symbol(built):
    from juicer_amd import capi
    hdr = open(os.path.join(ROOT, "include", "juicer_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(jd_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 25
    L = capi.lib()
    missing = [s for s in declared if not hasattr(L, s)]
    assert not missing, missing
    assert sorted(capi.EXPORTS) == declared
    assert b"gfx950" in L.jd_version()


def test_am_preparation_matches_oracle(built):
    from juicer_amd import capi, synth
    from oracle.oracle import OracleAM
    for am in (synth.config_small()[0], synth.make_models(3, n_gmm=40, n_hmm=30, n_mix=5, D=13, n_tm=7, with_tee=True)):
        g, o = capi.Models.from_htk(am), OracleAM(am)
        for a, b in zip(g.flat(), o.flat()):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
        gt, ot = g.trans(), o.trans()
        assert np.array_equal(gt[0].view(np.uint32), ot[0].view(np.uint32))
        assert np.array_equal(gt[1], ot[1]) and np.array_equal(gt[2].view(np.uint32), ot[2].view(np.uint32))
        assert (g.n_hmms, g.n_gmms, g.vec_size, g.max_states) == (am.n_hmm, am.n_gmm, am.D, am.max_n)


def test_net_preparation_matches_oracle(built):
    from juicer_amd import capi, synth
    from oracle.oracle import OracleNet
    _, net, _, _ = synth.config_small()
    for scale, pen in ((1.0, 0.0), (7.5, -3.25)):
        g = capi.Network.from_synth(net, scale, pen)
        o = OracleNet(net, scale, pen).arrays()
        c = g.csr()
        assert g.n_arcs == net.n_arcs and g.init_state == int(net.src[0])
        # synthetic arcs are sorted by source state, so oracle file order == CSR order
        assert np.array_equal(c["row_ptr"][:-1], o["first"]) and np.array_equal(np.diff(c["row_ptr"]), o["cnt"])
        assert np.array_equal(c["to"], o["to"]) and np.array_equal(c["ilab"], o["ilab"])
        assert np.array_equal(c["olab"], o["olab"])
        assert np.array_equal(c["w"].view(np.uint32), o["w"].view(np.uint32))
        assert np.array_equal(np.isfinite(c["fin_w"]), o["final_ind"] >= 0)


def test_fsm_text_loader(built, tmp_path):
    """AT&T text FSM (WFSTNetwork.cpp:414-447 sscanf cascade): 5/4-field arcs, 2/1-field finals."""
    from juicer_amd import capi, synth
    _, net, _, _ = synth.config_toy()
    p = tmp_path / "toy.fsm"
    with open(p, "w") as f:
        for i in range(net.n_arcs):
            if net.w_file[i] == 0.0:
                f.write("%d %d %d %d\n" % (net.src[i], net.dst[i], net.ilab[i], net.olab[i]))
            else:
                f.write("%d %d %d %d %.9g\n" % (net.src[i], net.dst[i], net.ilab[i], net.olab[i], net.w_file[i]))
        for s, w in zip(net.fstate, net.fweight_file):
            f.write("%d %.9g\n" % (s, w))
        f.write("\n# trailing junk line\n")
    insyms = tmp_path / "in.syms"
    insyms.write_text("<eps> 0\n" + "".join("h%d %d\n" % (i, i) for i in range(1, 6)))
    a = capi.Network.from_fsm_file(str(p), str(insyms), None, 2.0, -1.0).csr()
    b = capi.Network.from_synth(net, 2.0, -1.0).csr()
    for k in a:
        assert np.array_equal(a[k].view(np.uint32) if a[k].dtype == np.float32 else a[k],
                              b[k].view(np.uint32) if b[k].dtype == np.float32 else b[k]), k
    bad = tmp_path / "aux.syms"
    bad.write_text("<eps> 0\n#0 1\nh 9\n")
    with pytest.raises(capi.JuicerAmdError) as ei:
        capi.Network.from_fsm_file(str(p), str(bad), None)
    assert ei.value.code == capi.JD_EFORMAT


def test_errors_are_codes_not_exits(built):
    from juicer_amd import capi, synth
    am, net, _, _ = synth.config_toy()
    import copy
    bad = copy.deepcopy(net)
    for f in ("src", "dst", "ilab", "olab", "w_file"):    # first arc moved to the end: state 0 is split
        a = getattr(bad, f); setattr(bad, f, np.concatenate([a[1:], a[:1]]))
    with pytest.raises(capi.JuicerAmdError) as ei:
        capi.Network.from_synth(bad)
    assert ei.value.code == capi.JD_EFORMAT
    bad_am = copy.deepcopy(am)
    bad_am.hmm_gmm[0, 1] = 999
    with pytest.raises(capi.JuicerAmdError):
        capi.Models.from_htk(bad_am)


def test_no_cpu_fallback(built):
    """Without a GPU the product path must refuse to run (JD_ENODEV), never fall back."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from juicer_amd import capi, synth
    am, net, feats, _ = synth.config_toy()
    g, m = capi.Network.from_synth(net), capi.Models.from_htk(am)
    with pytest.raises(capi.JuicerAmdError) as ei:
        capi.Decoder(g, m)
    assert ei.value.code == capi.JD_ENODEV
    with pytest.raises(capi.JuicerAmdError) as ei:
        m.score_frames(feats[0][:4])
    assert ei.value.code == capi.JD_ENODEV
    cl, gr = synth.make_cl_g(1, am, n_words=5, n_succ=2)
    with pytest.raises(capi.JuicerAmdError) as ei:
        capi.Network.compose(capi.Network.from_synth(cl), capi.Network.from_synth(gr))
    assert ei.value.code == capi.JD_ENODEV
    with pytest.raises(capi.JuicerAmdError) as ei:
        capi.Network.lazy(capi.Network.from_synth(cl), capi.Network.from_synth(gr), m)
    assert ei.value.code == capi.JD_ENODEV


def test_product_does_not_reference_oracle():
    """The product package and headers never import / link the oracle."""
    for base in ("juicer_amd", "include"):
        for dp, _, fns in os.walk(os.path.join(ROOT, base)):
            for fn in fns:
                if fn.endswith((".py", ".cpp", ".hip", ".h", ".hpp")):
                    txt = open(os.path.join(dp, fn), errors="ignore").read()
                    assert "juicer_oracle" not in txt and "from oracle" not in txt and "import oracle" not in txt, fn


def test_mmf_text_loader_shared_macros(built, tmp_path):
    """HTK MMF text -> same prepared parameters as the array entry point (bit-exact)."""
    from juicer_amd import capi, io as jio, synth
    am = synth.make_models(21, n_gmm=30, n_hmm=14, n_mix=3, D=13, n_tm=5, with_tee=True)
    p = tmp_path / "m.mmf"
    jio.write_mmf(p, am)
    a, b = capi.Models.from_mmf_file(str(p)), capi.Models.from_htk(am)
    assert (a.n_hmms, a.n_gmms, a.vec_size, a.max_states, a.max_mix, a.n_tm) == \
           (b.n_hmms, b.n_gmms, b.vec_size, b.max_states, b.max_mix, b.n_tm)
    for x, y in zip(a.flat(), b.flat()):
        assert np.array_equal(x.view(np.uint32), y.view(np.uint32))
    for x, y in zip(a.trans(), b.trans()):
        assert np.array_equal(x.view(np.uint32) if x.dtype == np.float32 else x,
                              y.view(np.uint32) if y.dtype == np.float32 else y)
    for x, y in zip(a.topology(), b.topology()):
        assert np.array_equal(x, y)


def test_mmf_text_loader_inline_and_single_mixture(built, tmp_path):
    """Inline states / <TRANSP>, and the implicit single-mixture form (no <NUMMIXES>/<MIXTURE>)."""
    from juicer_amd import capi, io as jio, synth
    am = synth.make_models(22, n_gmm=12, n_hmm=9, n_mix=1, D=5, n_tm=3)
    p = tmp_path / "m.mmf"
    jio.write_mmf(p, am, inline_every=3)
    a, b = capi.Models.from_mmf_file(str(p)), capi.Models.from_htk(am)
    assert a.n_hmms == b.n_hmms and a.n_gmms == b.n_gmms + 3 * 3      # 3 inline HMMs x 3 states duplicated
    (adet, amu, aiv), (bdet, bmu, biv) = a.flat(), b.flat()
    (atrp, ase, atee), (btrp, bse, btee) = a.trans(), b.trans()
    ahn, ahg, aht, _ = a.topology()
    bhn, bhg, bht, _ = b.topology()
    assert np.array_equal(ahn, bhn) and np.array_equal(atee.view(np.uint32), btee.view(np.uint32))
    for h in range(a.n_hmms):
        for j in range(1, ahn[h] - 1):
            ga, gb = ahg[h, j], bhg[h, j]
            assert np.array_equal(adet[ga].view(np.uint32), bdet[gb].view(np.uint32))
            assert np.array_equal(amu[ga], bmu[gb]) and np.array_equal(aiv[ga].view(np.uint32), biv[gb].view(np.uint32))
        n = ahn[h]
        assert np.array_equal(atrp[aht[h]][:n, :n].view(np.uint32), btrp[bht[h]][:n, :n].view(np.uint32))
        assert np.array_equal(ase[aht[h]][:n], bse[bht[h]][:n])


def test_mmf_text_loader_errors(built, tmp_path):
    from juicer_amd import capi
    p = tmp_path / "bad.mmf"
    p.write_text('~o <VECSIZE> 3 <USER><DIAGC>\n~m "mix1" <MEAN> 3 0 0 0 <VARIANCE> 3 1 1 1\n')
    with pytest.raises(capi.JuicerAmdError) as ei:
        capi.Models.from_mmf_file(str(p))
    assert ei.value.code == capi.JD_EFORMAT
    p.write_text('~o <VECSIZE> 3 <USER><DIAGC>\n~h "a" <BEGINHMM> <NUMSTATES> 3 <STATE> 2 <MEAN> 2 0 0 <VARIANCE> 2 1 1\n')
    with pytest.raises(capi.JuicerAmdError):
        capi.Models.from_mmf_file(str(p))
    with pytest.raises(capi.JuicerAmdError):
        capi.Models.from_mmf_file(str(tmp_path / "missing.mmf"))


def test_adapter_compiles_inside_the_juicer_tree():
    """include/juicer_amd_decoder.hpp has two branches; the one a Juicer maintainer would use
    (#ifdef DECODER_H: derive from Juicer::IDecoder, return Juicer::DecHyp, Juicer::WFSTLattice)
    is compiled here against an own-written mock of Decoder.h's declarations
    (tests/mock_juicer/Decoder.h; the real header needs the absent Torch3), the stand-alone
    branch against nothing.  Compile-only: no GPU, no link."""
    import subprocess
    import tempfile
    inc = os.path.join(ROOT, "include")
    mock = os.path.join(ROOT, "tests", "mock_juicer")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-fsyntax-only", "-I", mock, "-I", inc,
                           os.path.join(mock, "adapter_in_tree.cpp")])
    subprocess.check_call(["g++", "-std=c++98", "-Wall", "-fsyntax-only", "-I", mock, "-I", inc, "-x", "c++",
                           "-include", "Decoder.h", os.path.join(inc, "juicer_amd_decoder.hpp")])   # Juicer is C++98-era
    # ... and with WFSTNetwork.h / HTKFlatModels.h in front of it as well: the exact-signature constructor and its bridge
    subprocess.check_call(["g++", "-std=c++98", "-Wall", "-fsyntax-only", "-I", mock, "-I", inc, "-x", "c++", "-include", "Decoder.h",
                           "-include", "WFSTNetwork.h", "-include", "HTKFlatModels.h", os.path.join(inc, "juicer_amd_decoder.hpp")])
    with tempfile.NamedTemporaryFile("w", suffix=".cpp") as f:
        f.write('#include "juicer_amd_decoder.hpp"\nJuicerAmd::IDecoder *p = 0;\n'
                'int main() { return DHHTYPE == 1 ? 0 : 1; }\n')          # DecHypHistPool.h:106
        f.flush()
        subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-fsyntax-only", "-I", inc, f.name])


def test_exact_signature_seam_round_trip(built, tmp_path):
    """`GpuWFSTDecoder(WFSTNetwork*, IModels*, real, real, real, real, int)` - WFSTDecoderLite's own constructor
    signature (WFSTDecoderLite.h:81-89, call site juicer.cpp:582-586) - walks the objects juicer.cpp has already built
    into the C ABI through WFSTNetwork's public getters (WFSTNetwork.h:129-167) and HTKFlatModels' tables.  Here the
    walk is RUN (no GPU: netFromJuicer / modelsFromJuicer only) against mocks of those declarations
    (tests/mock_juicer/) filled from a synthetic case, and what the C ABI then holds must be the same arrays, bit
    for bit - with and without the 4-aligned vector stride of OPT_ALIGN4 (HTKFlatModels.cpp:117-125)."""
    import subprocess
    from bridge_helper import build_program, read_arrays, write_case
    from juicer_amd import capi, synth
    exe = build_program(tmp_path)
    am, net, feats, _ = synth.config_mixed(n_utts=1)                  # 3..8-state HMMs, skips, a tee model
    gnet, gam = capi.Network.from_synth(net, 7.5, -2.0), capi.Models.from_htk(am)
    for pad in (0, 1):
        write_case(tmp_path / "case.bin", gnet, gam, feats[0], (0.0, 150.0, 0.0, 0.0, 0.0), pad=pad)
        subprocess.check_call([exe, str(tmp_path / "case.bin"), str(tmp_path / "out.bin")])
        i4, f4, i2 = np.int32, np.float32, np.int16
        (meta, row, to, w, il, ol, fin, nm, det, mean, ivar, hn, hg, ht, tee, trP, se) = read_arrays(
            tmp_path / "out.bin", [i4, i4, i4, f4, i4, i4, f4, i4, f4, f4, f4, i4, i4, i4, f4, f4, i2])
        c = gnet.csr()
        assert meta.tolist() == [gnet.n_states, gnet.init_state, gam.n_gmms, gam.max_mix, gam.n_hmms, gam.max_states, gam.n_tm, gam.vec_size]
        for got, want in ((row, c["row_ptr"]), (to, c["to"]), (il, c["ilab"]), (ol, c["olab"])):
            assert np.array_equal(got, want)
        assert np.array_equal(w.view(np.uint32), c["w"].view(np.uint32))
        assert np.array_equal(np.isfinite(fin), np.isfinite(c["fin_w"])) and np.array_equal(fin[np.isfinite(fin)], c["fin_w"][np.isfinite(fin)])
        hn0, hg0, ht0, nm0 = gam.topology()
        det0, mean0, ivar0 = gam.flat()
        trP0, se0, tee0 = gam.trans()
        assert np.array_equal(nm, nm0) and np.array_equal(hn, hn0) and np.array_equal(ht, ht0)
        for h in range(gam.n_hmms):                                    # (entry / exit slots carry no tied state)
            assert np.array_equal(hg.reshape(hg0.shape)[h, 1:hn0[h] - 1], hg0[h, 1:hn0[h] - 1])
        for g in range(gam.n_gmms):
            k = nm0[g]
            assert np.array_equal(det.reshape(det0.shape)[g, :k].view(np.uint32), det0[g, :k].view(np.uint32))
            assert np.array_equal(mean.reshape(mean0.shape)[g, :k].view(np.uint32), mean0[g, :k].view(np.uint32))
            assert np.array_equal(ivar.reshape(ivar0.shape)[g, :k].view(np.uint32), ivar0[g, :k].view(np.uint32))
        assert np.array_equal(tee.view(np.uint32), tee0.view(np.uint32))
        for t in np.unique(ht0):
            n = int(hn0[np.nonzero(ht0 == t)[0][0]])
            assert np.array_equal(trP.reshape(trP0.shape)[t, :n, :n].view(np.uint32), trP0[t, :n, :n].view(np.uint32))
            assert np.array_equal(se.reshape(se0.shape)[t, 1:n], se0[t, 1:n])


def test_csr_input_is_validated(built):
    """jd_net_create_csr refuses malformed CSR input instead of walking out of bounds later."""
    from juicer_amd import capi
    ok = dict(n_states=3, init_state=0, row_ptr=[0, 2, 3, 3], to=[1, 2, 2], w=[0.5, 1.0, 0.25], ilab=[1, 2, 1],
              olab=[0, 1, 0], fstate=[2], fweight=[0.0])
    assert capi.Network.from_csr(**ok).n_arcs == 3
    for bad in (dict(row_ptr=[1, 2, 3, 3]), dict(row_ptr=[0, 2, 1, 3]), dict(row_ptr=[0, 0, 0, 0]),
                dict(to=[1, 2, 7]), dict(init_state=5)):
        with pytest.raises(capi.JuicerAmdError) as e:
            capi.Network.from_csr(**{**ok, **bad})
        assert e.value.code == capi.JD_EINVAL


def test_closure_path_counts_against_the_recursion(built):
    """collectPaths' count trigger counts the reference's Path objects without writing them: per state, what ONE arriving token
    makes propagateToken create in its closure (WFSTDecoderLite.cpp:497-509 inside the recursion of :533-541, :583-599: a Path
    per labelled epsilon arc and per labelled arc of a tee model, recursively, with multiplicity).  The library's post-order
    walk against the recursion written out in Python, on graphs with a flat 400-word fan-out, a lexicon tree and HMMs of mixed
    size; a label-less cycle is reported."""
    import sys
    from juicer_amd import capi, synth
    sys.setrecursionlimit(20000)
    cases = [synth.config_small(seed=31, n_utts=1, n_words=400, n_succ=5, n_gmm=120, n_hmm=45, hub="flat"),
             synth.config_small(seed=7, n_utts=1, hub="tree"), synth.config_mixed(seed=12, n_utts=1, n_words=120, n_succ=6)]
    for am, net, _, _ in cases:
        gnet, gam = capi.Network.from_synth(net), capi.Models.from_htk(am)
        got, acyclic = gnet.closure_path_counts(gam)
        assert acyclic
        c = gnet.csr()
        tee = np.zeros(am.n_hmm, bool)
        for h in range(am.n_hmm):
            n = int(am.hmm_nstates[h]); a = am.transp[am.hmm_tm[h]]
            sucs = [j for j in range(n) if a[0, j] > 0]
            tee[h] = (n - 1) in sucs[1:]                               # (HTKModels.cpp:581-593: not the FIRST successor of state 0)
        memo = {}

        def count(q):
            if q in memo:
                return memo[q]
            tot = 0
            for a in range(int(c["row_ptr"][q]), int(c["row_ptr"][q + 1])):
                il, ol, to = int(c["ilab"][a]), int(c["olab"][a]), int(c["to"][a])
                if il == 0 or tee[il - 1]:
                    tot += (1 if ol != 0 else 0) + count(to)
            memo[q] = min(tot, 1 << 20)
            return memo[q]
        want = np.array([count(q) for q in range(gnet.n_states)], np.int64)
        assert np.array_equal(got.astype(np.int64), want) and want.max() > 0
    # a cycle of epsilon arcs: the reference would not come back from it; the counts are not used (the rule then runs on this build's own records)
    cyc = capi.Network.from_arcs(src=[0, 1, 2, 2], dst=[1, 2, 1, 3], ilab=[1, 0, 0, 1], olab=[0, 5, 0, 0], w_file=[0.0, 0.0, 0.0, 0.0],
                                 fstate=[3], fweight_file=[0.0])
    am, _, _, _ = synth.config_toy()
    _, acyclic = cyc.closure_path_counts(capi.Models.from_htk(am))
    assert not acyclic


def test_network_loaders_on_random_topologies(built, tmp_path):
    """Graphs of arbitrary shape (tests/random_topology.py: parallel arcs, self loops, epsilon arcs with and without labels, arcs into the
    initial state, finals anywhere) through every way into the library: arrays, the AT&T text file (with a scale and a penalty), the JWNT
    file this build writes and reads - the same CSR each time, and the oracle's (what the parity tests on these graphs ride on)."""
    import sys as _sys
    _sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import random_topology as rt
    from juicer_amd import capi, io as jio, synth
    from oracle.oracle import OracleNet
    bits = lambda a: a.view(np.uint32) if a.dtype == np.float32 else a
    am = synth.make_models(3, n_gmm=40, n_hmm=20, n_mix=2, n_tm=4, with_tee=True)
    n_parallel = n_self = 0
    for seed in range(9000, 9012):
        net = rt.random_net(seed, am, n_states=int(10 + 9 * (seed % 7)), arcs_per_state=3.0)
        pairs = list(zip(net.src.tolist(), net.dst.tolist()))
        n_parallel += len(pairs) - len(set(pairs)); n_self += sum(s == d for s, d in pairs)
        for scale, pen in ((1.0, 0.0), (6.5, -1.75)):
            c = capi.Network.from_synth(net, scale, pen).csr()
            o = OracleNet(net, scale, pen).arrays()
            assert np.array_equal(np.diff(c["row_ptr"]), o["cnt"])
            # (the oracle keeps the file's order - the initial state's arcs first -, the CSR is by state number: row by row)
            rows = np.concatenate([np.arange(o["first"][q], o["first"][q] + o["cnt"][q]) for q in range(net.n_states)]).astype(np.int64)
            for k in ("to", "ilab", "olab"): assert np.array_equal(c[k], o[k][rows]), (seed, k)
            assert np.array_equal(bits(c["w"]), bits(o["w"][rows])) and np.array_equal(np.isfinite(c["fin_w"]), o["final_ind"] >= 0)
            p = str(tmp_path / ("n%d.fsm" % seed))
            jio.write_fsm(p, net)
            f = capi.Network.from_fsm_file(p, None, None, scale, pen).csr()
            for k in c: assert np.array_equal(bits(f[k]), bits(c[k])), (seed, scale, k)
        g0 = capi.Network.from_synth(net)
        pj = str(tmp_path / ("n%d.jwnt" % seed))
        g0.save_jwnt(pj)
        j = capi.Network.from_jwnt_file(pj).csr()
        c = g0.csr()
        for k in c: assert np.array_equal(bits(j[k]), bits(c[k])), (seed, "jwnt", k)
        assert g0.init_state == int(net.src[0])
    assert n_parallel > 0 and n_self > 0, "the generator made no parallel arcs / self loops: nothing was tested"


def test_ctypes_mirrors_have_the_headers_layout(built, tmp_path):
    """juicer_amd/capi.py mirrors five structs of include/juicer_amd.h by hand; jd_stats grew twice this round.  A C program compiled
    against the header says what the sizes and the field offsets ARE (gcc, the compiler a caller of the C ABI uses)."""
    import ctypes as C
    import subprocess
    from juicer_amd import capi
    pairs = [("jd_stats", capi.Stats), ("jd_hyp", capi.CHyp), ("jd_timing", capi.Timing), ("jd_pipe_stats", capi.PipeStats), ("jd_broker_stats", capi.BrokerStats)]
    src = ["#include <stdio.h>", "#include <stddef.h>", '#include "juicer_amd.h"', "int main(void) {"]
    for cname, cls in pairs:
        src.append('  printf("%s size %%zu\\n", sizeof(%s));' % (cname, cname))
        for f in cls._fields_:
            src.append('  printf("%s %s %%zu\\n", offsetof(%s, %s));' % (cname, f[0], cname, f[0]))
    src += ["  return 0;", "}"]
    c = tmp_path / "layout.c"
    c.write_text("\n".join(src) + "\n")
    exe = str(tmp_path / "layout")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run(["gcc", "-std=c99", "-Wall", "-I", os.path.join(root, "include"), "-o", exe, str(c)], check=True)   # (a field the header lacks does not compile)
    want = {}
    for line in subprocess.run([exe], check=True, capture_output=True, text=True).stdout.splitlines():
        a, b, v = line.split()
        want[(a, b)] = int(v)
    for cname, cls in pairs:
        assert C.sizeof(cls) == want[(cname, "size")], (cname, C.sizeof(cls), want[(cname, "size")])
        for f in cls._fields_:
            assert getattr(cls, f[0]).offset == want[(cname, f[0])], (cname, f[0])
