#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the COMPILED REFERENCE (oracle/_ref/libjref.so).

Run in the dev container (where /root/reference exists):
    python tests/make_golden.py
Each fixture holds a seeded synthetic model *as the reference's own loader left
it in memory* (exported through the product-side flattening code), the input
frames, and the reference's outputs.  The fixtures are small (<200 kB each) and
are what pins oracle/ and the HIP engine on machines without the reference.
"""
import sys
import tempfile
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from julius_amd import synth  # noqa: E402
from oracle import pyoracle  # noqa: E402

OUT = ROOT / "tests" / "golden"


def save(name, **kw):
    OUT.mkdir(parents=True, exist_ok=True)
    np.savez_compressed(OUT / name, **{k: v for k, v in kw.items() if v is not None})
    print("wrote", name, {k: getattr(v, "shape", v) for k, v in kw.items() if v is not None})


def model_arrays(m):
    keys = ("mean", "ivar", "gconst", "st_off", "ent_dens", "ent_logw", "st_book")
    d = {k: m[k] for k in keys if m.get(k) is not None}
    d["nbook"] = np.int32(m.get("nbook", 0))
    if m.get("book_size"):
        d["book_size"] = np.int32(m["book_size"])
    return d


def main():
    ref = pyoracle.Ref()
    tmp = Path(tempfile.mkdtemp())

    # G1: plain GMM, uniform mixtures, gprune none (calc_mix + gprune_none)
    m = synth.make_gmm(S=24, M=8, D=39, seed=11)
    synth.write_hmmdefs(tmp / "g1", m)
    am = ref.am_load(tmp / "g1", gprune="none")
    fr = synth.make_frames(m, T=40, seed=12)
    save("gmm_plain_none.npz", frames=fr, out=am.outprob(fr), **model_arrays(am.export()))

    # G2: ragged mixture counts + NULL densities, D=25, gprune none and safe(3)
    m = synth.make_gmm(S=21, M=6, D=25, seed=21, ragged=True, null_frac=0.1)
    synth.write_hmmdefs(tmp / "g2", m, kind="MFCC_E_D_N_Z")
    am = ref.am_load(tmp / "g2", gprune="none")
    fr = synth.make_frames(m, T=33, seed=22)
    ex = am.export()
    out_none = am.outprob(fr)
    am2 = ref.am_load(tmp / "g2", gprune="safe", gprune_num=3)
    save("gmm_ragged.npz", frames=fr, out=out_none, out_safe3=am2.outprob(fr), **model_arrays(ex))

    # G3: tied-mixture, 3 codebooks x 32, safe top-2 / top-4 and none
    m = synth.make_tied_gmm(S=18, nbook=3, K=32, D=39, seed=31)
    synth.write_hmmdefs(tmp / "g3", m)
    fr = synth.make_frames(m, T=30, seed=32, noise=2.0)
    am = ref.am_load(tmp / "g3", gprune="none", gprune_num=32)
    ex = am.export()
    o_none = am.outprob(fr)
    am2 = ref.am_load(tmp / "g3", gprune="safe", gprune_num=2)
    o_s2 = am2.outprob(fr)
    c2 = am2.tmix_cache(fr, 1, 2)
    am4 = ref.am_load(tmp / "g3", gprune="safe", gprune_num=4)
    o_s4 = am4.outprob(fr)
    save("gmm_tied.npz", frames=fr, out_none=o_none, out_safe2=o_s2, out_safe4=o_s4,
         cache2_score=c2[0], cache2_id=c2[1], cache2_num=c2[2], **model_arrays(ex))

    # G4: pseudo-phone state sets (outprob_cd max / avg / nbest)
    m = synth.make_gmm(S=30, M=4, D=39, seed=41)
    synth.write_hmmdefs(tmp / "g4", m)
    fr = synth.make_frames(m, T=25, seed=42)
    rng = np.random.default_rng(43)
    sizes = rng.integers(1, 9, size=12)
    set_off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    states = np.concatenate([rng.choice(30, size=n, replace=False) for n in sizes]).astype(np.int32)
    cds = {}
    for meth in ("max", "avg", "nbest"):
        am = ref.am_load(tmp / "g4", gprune="none", cdset=meth, cdmax=3)
        cds["cd_" + meth] = am.outprob_cd(fr, set_off, states)
    scores = am.outprob(fr)
    save("cdset.npz", scores=scores, set_off=set_off, states=states, **cds)

    # D1: DNN (reference FMA path, calc_dnn_fma.c) -- 48 -> 3 x 64 sigmoid -> 40
    dnn = synth.make_dnn(dims=(48, 64, 64, 64, 40), seed=51)
    r = ref.dnn_load(dnn, tmp)
    fr = np.random.default_rng(52).normal(0, 1.5, (24, 48)).astype(np.float32)
    simd = ref.lib.jref_simd_string().decode()
    assert "FMA" in simd, "golden DNN vectors must come from the FMA path"
    save("dnn_small.npz", frames=fr, out=r.outprob(fr), dims=dnn["dims"], prior=dnn["prior"],
         **{f"w{l}": dnn["w"][l] for l in range(4)}, **{f"b{l}": dnn["b"][l] for l in range(4)})

    # tables
    o = pyoracle.Oracle()
    tbl = o.log_tbl()
    save("tables.npz", addlog_idx=np.arange(0, 500000, 997, dtype=np.int32),
         addlog_val=tbl[::997], logistic_idx=np.arange(0, 320001, 641, dtype=np.int32),
         logistic_val=o.logistic_tbl()[::641])



def beam_fixture(ref, tmp, name, seed, beam, extra, nutt=3, **task_kw):
    """A complete first-pass case: the reference recogniser loads a synthetic
    tied-state triphone task (hmmdefs + HMMList + dict + ARPA 2-gram), the
    product-side shim flattens its lexicon (-> lexicon blob), and the reference's
    own first pass produces the word trellis / pass-1 sentence per utterance."""
    from julius_amd import lexblob
    d = tmp / name
    task = synth.make_triphone_task(d, seed=seed, **task_kw)
    args = ["-h", task["hmmdefs"], "-hlist", task["hmmlist"], "-v", task["dict"], "-nlr", task["arpa"],
            "-input", "htkparam", "-1pass", "-gprune", "none", "-b", str(beam)] + list(extra)
    eng = pyoracle.RefEngine(ref, args)
    eng.save_lexicon(d / "lex.blob")
    lex = lexblob.load(d / "lex.blob")
    am = ref.am_load(task["hmmdefs"], task["hmmlist"])
    model = am.export()
    bs = float(extra[extra.index("-bs") + 1]) if "-bs" in extra else -1.0
    out = dict(beam_width=np.int32(eng.beam_width), score_pruning_width=np.float32(bs), nutt=np.int32(nutt),
               args=np.array(" ".join(str(a) for a in args[10:])))
    for k in ("mean", "ivar", "gconst", "st_off", "ent_dens", "ent_logw"):
        out["am_" + k] = model[k]
    for k, v in lex.items():
        out["lex_" + k] = np.asarray(v)
    for u in range(nutt):
        fr, _ = synth.make_utterance(task, nwords=4 + u, seed=1000 * seed + u)
        synth.write_htk_param(d / "u.mfc", fr)
        tr, (wseq, sc) = eng.recognize(d / "u.mfc")
        out[f"u{u}_frames"] = fr
        for k, v in tr.items():
            out[f"u{u}_tr_{k}"] = v
        out[f"u{u}_wseq"] = wseq
        out[f"u{u}_score"] = np.float32(sc)
    save(name + ".npz", **out)


def grammar_fixture(ref, tmp, name, seed, beam, extra, nutt=3, ncat=3, wrap=True, **task_kw):
    """First pass under a DFA grammar (one lexicon tree per category, category-pair constraint
    between words): cross-word triphone task + the grammar of synth.make_triphone_grammar()."""
    from julius_amd import lexblob
    d = tmp / name
    task = synth.make_triphone_grammar(synth.make_triphone_task(d, seed=seed, **task_kw), ncat=ncat, seed=seed, wrap=wrap)
    args = ["-h", task["hmmdefs"], "-hlist", task["hmmlist"], "-dfa", task["dfa"], "-v", task["gdict"],
            "-input", "htkparam", "-1pass", "-gprune", "none", "-b", str(beam)] + list(extra)
    eng = pyoracle.RefEngine(ref, args)
    eng.save_lexicon(d / "lex.blob")
    lex = lexblob.load(d / "lex.blob")
    model = ref.am_load(task["hmmdefs"], task["hmmlist"]).export()
    bs = float(extra[extra.index("-bs") + 1]) if "-bs" in extra else -1.0
    out = dict(beam_width=np.int32(eng.beam_width), score_pruning_width=np.float32(bs), nutt=np.int32(nutt),
               args=np.array(" ".join(str(a) for a in args[10:])))
    for k in ("mean", "ivar", "gconst", "st_off", "ent_dens", "ent_logw"):
        out["am_" + k] = model[k]
    for k, v in lex.items():
        out["lex_" + k] = np.asarray(v)
    for u in range(nutt):
        fr, _ = synth.make_triphone_grammar_utterance(task, nwords=3 + u, seed=1000 * seed + u)
        synth.write_htk_param(d / "u.mfc", fr)
        tr, (wseq, sc) = eng.recognize(d / "u.mfc")
        out[f"u{u}_frames"] = fr
        for k, v in tr.items():
            out[f"u{u}_tr_{k}"] = v
        out[f"u{u}_wseq"] = wseq
        out[f"u{u}_score"] = np.float32(sc)
    save(name + ".npz", **out)


def main_beam():
    ref = pyoracle.Ref()
    tmp = Path(tempfile.mkdtemp())
    # B1: rank beam only; lexicon tree with shared roots (factoring) and isolated roots
    beam_fixture(ref, tmp, "beam_rank", seed=0, beam=120, extra=["-sepnum", "5"])
    # B2: rank + score beam, IWCD max, different LM weights
    beam_fixture(ref, tmp, "beam_score", seed=3, beam=80, extra=["-sepnum", "3", "-bs", "60", "-iwcd1", "max",
                                                                  "-lmp", "5.0", "-1.0"])
    # B3: every word separated from the tree (all roots isolated), IWCD avg, narrow beam
    beam_fixture(ref, tmp, "beam_isolated", seed=4, beam=40, extra=["-iwcd1", "avg"], nword=40)
    # B4: DFA grammar, per-category trees, word insertion penalty, N-best state sets
    grammar_fixture(ref, tmp, "beam_grammar", seed=6, beam=100, extra=["-penalty1", "-2.0", "-iwcd1", "best", "3"],
                    nword=50)
    # B5: grammar whose sentences may start with any word: more initial tokens than the beam is wide
    grammar_fixture(ref, tmp, "beam_grammar_free", seed=7, beam=30, extra=["-penalty1", "-1.0", "-iwcd1", "max", "-bs", "80"],
                    nword=70, wrap=False)


def main_gms():
    """S1: Gaussian mixture selection (-gshmm, -gsnum 4 / 24): what outprob_state() returns for every
    state of the triphone model once gms_state() stands in front of it."""
    ref = pyoracle.Ref()
    tmp = Path(tempfile.mkdtemp())
    task = synth.make_triphone_task(tmp, seed=5, nword=60)
    gpath, _ = synth.make_gs_model(task, seed=5)
    frs = [synth.make_utterance(task, nwords=2 + u, seed=50 + u)[0] for u in range(3)]
    full = ref.am_load(task["hmmdefs"], task["hmmlist"]).export()
    out = {"full_" + k: v for k, v in model_arrays(full).items()}
    out["frames"] = np.concatenate(frs)
    out["utt_off"] = np.cumsum([0] + [len(f) for f in frs]).astype(np.int32)
    for nbest in (4, 24):
        am = ref.am_load(task["hmmdefs"], task["hmmlist"], gshmm=gpath, gms_num=nbest)
        gs = am.gms()
        out.update({"gs_" + k: v for k, v in model_arrays(gs["model"]).items()})
        out["state2gs"] = gs["state2gs"]
        out["out_%d" % nbest] = np.concatenate([am.outprob(f) for f in frs])
    save("gms.npz", **out)


def main_rejgmm():
    """V1: GMM-based input verification (-gmm, -gmmnum 5 / 20): gmm_proceed()'s per-frame model scores
    through the reference's own entry points, and gc->gmm_score[] / the winner after whole inputs."""
    ref = pyoracle.Ref()
    tmp = Path(tempfile.mkdtemp())
    task = synth.make_triphone_task(tmp, seed=11, nword=60)
    gpath, _, names = synth.make_rejection_gmm(tmp, task["model"]["centre"], seed=11, null_frac=0.1)
    frs = [synth.make_utterance(task, nwords=2 + u, seed=60 + u)[0] for u in range(3)]
    out = dict(frames=np.concatenate(frs), utt_off=np.cumsum([0] + [len(f) for f in frs]).astype(np.int32))
    for num in (5, 20):
        eng = pyoracle.RefEngine(ref, ["-h", task["hmmdefs"], "-hlist", task["hmmlist"], "-v", task["dict"], "-nlr", task["arpa"],
                                       "-input", "htkparam", "-gprune", "none", "-b", "120", "-gmm", str(gpath),
                                       "-gmmnum", str(num), "-gmmreject", "noise,cough"])
        info = eng.gmm_info()
        out.update(model_arrays(info["model"]))
        out["model_state"] = info["model_state"]
        out["frame_scores_%d" % num] = np.concatenate([eng.gmm_frame_scores(f) for f in frs])
        sums, winners = [], []
        for u, f in enumerate(frs):
            synth.write_htk_param(tmp / "u.mfc", f)
            eng.recognize(tmp / "u.mfc")
            sc, mi, cm, valid, fc = eng.gmm_result()
            assert fc == len(f)
            sums.append(sc)
            winners.append(mi)
        out["utt_scores_%d" % num] = np.stack(sums)
        out["winner_%d" % num] = np.array(winners, np.int32)
    save("rejgmm.npz", **out)


def main_multipath():
    """B6: a model that needs multipath handling (two entry arcs, state skips, early exits): non-emitting
    word-begin / word-end nodes, frame 0 through get_back_trellis_proceed()."""
    ref = pyoracle.Ref()
    tmp = Path(tempfile.mkdtemp())
    trans = np.array([[0, .7, .3, 0, 0], [0, .5, .3, .2, 0], [0, 0, .5, .3, .2], [0, 0, 0, .6, .4], [0, 0, 0, 0, 0]])
    beam_fixture(ref, tmp, "beam_multipath", seed=8, beam=100, extra=["-sepnum", "4", "-bs", "90"], nword=50, trans=trans)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("am", "all"):
        main()
    if what in ("beam", "all"):
        main_beam()
    if what in ("multipath", "all"):
        main_multipath()
    if what in ("gms", "all"):
        main_gms()
    if what in ("rejgmm", "all"):
        main_rejgmm()
