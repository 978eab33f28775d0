"""Julius' BINARY HMM definition read directly (julius_amd/csrc/readers.hip), without a Julius
process -- checked against the reference's own reader: the file is written by the reference's write_binhmm()
(what mkbinhmm does), loaded by Julius' read_binhmm() inside jamd_export, flattened by the shim, and the direct
conversion must give the same bytes.  Host code only: runs without a GPU."""
import ctypes as C
import subprocess

import numpy as np
import pytest

from julius_amd import lexblob, lib, synth
from oracle import pyoracle

EXPORT = pyoracle.HERE.parent / "julius_amd" / "jamd_export"


def _ref():
    if not pyoracle.REF_SO.exists() or not EXPORT.exists():
        pytest.skip("oracle/_ref not built")
    r = pyoracle.Ref()
    r.lib.jref_write_binhmm.argtypes = [C.c_char_p, C.c_char_p]
    return r


def _blob_records(path, magic):
    """{name: (dtype, bytes)} of a JAMD* container."""
    raw = open(path, "rb").read()
    assert raw[:8] == magic
    n = int(np.frombuffer(raw[8:12], np.int32)[0])
    at, out = 12, {}
    for _ in range(n):
        name = raw[at:at + 24].split(b"\0")[0].decode()
        dtype, count = np.frombuffer(raw[at + 24:at + 32], np.int32)
        nbytes = (int(count) * (1 if dtype == 2 else 4) + 3) & ~3
        out[name] = (int(dtype), raw[at + 32:at + 32 + nbytes])
        at += 32 + nbytes
    assert at == len(raw)
    return out


@pytest.mark.parametrize("kind", ["triphone", "tied_mixture"])
def test_binhmm_direct_equals_export(tmp_path, kind):
    ref = _ref()
    if kind == "triphone":
        task = synth.make_triphone_task(tmp_path, seed=71, nword=60, nphone=8, S=120, M=4)
        extra = ["-hlist", task["hmmlist"], "-v", task["dict"], "-nlr", task["arpa"]]
    else:                                     # BASELINE configs[0] shape: <TMix> codebooks
        task = synth.make_grammar_task(tmp_path, seed=72)
        extra = ["-dfa", task["dfa"], "-v", task["dict"]]
    binhmm = tmp_path / "model.binhmm"
    assert ref.lib.jref_write_binhmm(str(task["hmmdefs"]).encode(), str(binhmm).encode()) == 0
    # Julius' own reader on the binary file -> shim -> blob
    subprocess.run([str(EXPORT), "-h", str(binhmm)] + [str(a) for a in extra] + ["-input", "htkparam", "-jamdout", str(tmp_path / "exp")],
                   check=True, capture_output=True)
    # the product's direct reader
    L = lib.load()
    rc = L.jamd_binhmm_to_blob(str(binhmm).encode(), str(tmp_path / "direct.am").encode())
    assert rc == 0, L.jamd_last_error()
    assert open(tmp_path / "direct.am", "rb").read() == open(tmp_path / "exp.am", "rb").read()
    # and the same model as the ASCII hmmdefs gives (ids and density numbering may differ: compare scores)
    a = lexblob.load_gmm(tmp_path / "direct.am")
    assert a["mean"].shape[1] == 39 and len(a["st_off"]) - 1 > 10
    # error paths
    (tmp_path / "junk").write_bytes(b"JBINHMMV2\0_Q\0" + b"\0" * 64)
    assert L.jamd_binhmm_to_blob(str(tmp_path / "junk").encode(), str(tmp_path / "x").encode()) != 0
    (tmp_path / "trunc").write_bytes(open(binhmm, "rb").read()[:3000])
    assert L.jamd_binhmm_to_blob(str(tmp_path / "trunc").encode(), str(tmp_path / "x").encode()) != 0


@pytest.mark.gpu
def test_device_model_from_binhmm(engine, tmp_path):
    """jamd_gmm_load_binhmm(): same scores as the descriptor path on the same model."""
    ref = _ref()
    task = synth.make_triphone_task(tmp_path, seed=74, nword=60, nphone=8, S=120, M=4)
    binhmm = tmp_path / "model.binhmm"
    assert ref.lib.jref_write_binhmm(str(task["hmmdefs"]).encode(), str(binhmm).encode()) == 0
    fr = synth.make_frames(task["model"], T=200, seed=3)
    am = ref.am_load(str(binhmm)).export()                     # Julius' read_binhmm() -> flat
    want = lib.Gmm(engine, am).outprob_host(fr)
    got = lib.Gmm.from_binhmm(engine, binhmm).outprob_host(fr)
    assert np.array_equal(got, want)


# ---------------------------------------------------------------------------------------------- binary N-gram
def _bingram_task(tmp_path, with_rl, seed=73):
    ref = _ref()
    ref.lib.jref_write_bingram.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p]
    task = synth.make_triphone_task(tmp_path, seed=seed, nword=80, nphone=8, S=120, M=2, with_rl3=with_rl)
    bingram = tmp_path / "lm.bingram"
    rc = ref.lib.jref_write_bingram(str(task["arpa"]).encode(), str(task["arpa_rl"]).encode() if with_rl else None,
                                    str(bingram).encode())
    assert rc == 0
    subprocess.run([str(EXPORT), "-h", task["hmmdefs"], "-hlist", task["hmmlist"], "-v", task["dict"], "-d", str(bingram),
                    "-input", "htkparam", "-jamdout", str(tmp_path / "exp")], check=True, capture_output=True)
    return task, bingram


@pytest.mark.parametrize("with_rl", [False, True])
def test_bingram_tables_equal_the_exported_lexicon(tmp_path, with_rl):
    """SURVEY 8f N3, LM half: the binary N-gram mkbingram writes (v5), read directly (csrc/readers.hip restating
    libsent/src/ngram/ngram_read_bin.c:240-365), gives byte for byte the first-pass tables jamd_export wrote through
    Julius' own reader -- forward 2-gram alone (DIR_LR), and backward 3-gram + additional forward 2-gram (the -nlr/-nrl
    pair: bi_prob_additional(), ngram_access.c:351).  A binary N-gram over another vocabulary is refused."""
    task, bingram = _bingram_task(tmp_path, with_rl)
    L = lib.load()
    same = C.c_int(-1)
    rc = L.jamd_bingram_check(str(tmp_path / "exp.lex").encode(), str(bingram).encode(), C.byref(same))
    assert rc == 0 and same.value == 1, L.jamd_last_error()
    recs = _blob_records(tmp_path / "exp.lex", b"JAMDLEX1")
    names = recs["ng_wname"][1].split(b"\0")
    assert b"<s>" in names and b"</s>" in names
    # another vocabulary (one word more in the LM): refused, with the reason
    other = tmp_path / "o"
    other.mkdir()
    task2 = synth.make_triphone_task(other, seed=73, nword=81, nphone=8, S=120, M=2, with_rl3=with_rl)
    ref = _ref()
    assert ref.lib.jref_write_bingram(str(task2["arpa"]).encode(), str(task2["arpa_rl"]).encode() if with_rl else None,
                                      str(other / "lm.bingram").encode()) == 0
    assert L.jamd_bingram_check(str(tmp_path / "exp.lex").encode(), str(other / "lm.bingram").encode(), None) != 0
    assert b"vocabulary" in L.jamd_last_error()
    # a retrained N-gram over the SAME vocabulary is accepted, its tables differ
    lines = open(task["arpa"]).read().replace("-0.3", "-0.4")
    (tmp_path / "lm2.arpa").write_text(lines)
    assert ref.lib.jref_write_bingram(str(tmp_path / "lm2.arpa").encode(), str(task["arpa_rl"]).encode() if with_rl else None,
                                      str(tmp_path / "lm2.bingram").encode()) == 0
    rc = L.jamd_bingram_check(str(tmp_path / "exp.lex").encode(), str(tmp_path / "lm2.bingram").encode(), C.byref(same))
    assert rc == 0


@pytest.mark.gpu
@pytest.mark.parametrize("with_rl", [False, True])
def test_lexicon_with_ngram_read_directly(engine, tmp_path, with_rl):
    """jamd_lexicon_load_ngram(PREFIX.lex, lm.bingram): tree half from jamd_export, N-gram half from the binary N-gram;
    with the N-gram the tree was built from the first pass gives jamd_lexicon_load()'s result exactly."""
    task, bingram = _bingram_task(tmp_path, with_rl, seed=75)
    am = lib.Gmm.from_file(engine, tmp_path / "exp.am")
    outs = []
    for lx in (lib.Lexicon.from_file(engine, tmp_path / "exp.lex"), lib.Lexicon.from_file(engine, tmp_path / "exp.lex", bingram=bingram)):
        bm = lib.Beam(engine, lx, 200, -1.0, max_utts=2)
        scs = [am.outprob_host(synth.make_utterance(task, nwords=4 + u, seed=900 + u)[0]) for u in range(2)]
        res, tre = bm.pass1_host(scs)
        outs.append([(r.status, r.score, list(r.wseq[:r.wnum]), t.tobytes()) for r, t in zip(res, tre)])
        assert all(r.status == 0 for r in res)
        bm.close()
    assert outs[0] == outs[1]


# ------------------------------------------------------------------------- a retrained N-gram under an exported tree
def _retrain_task(tmp_path, with_rl=False, seed=77, sepnum=5):
    """A 300-word task exported with -sepnum 5 (so that most words stay IN the tree and shared nodes carry 1-gram
    factoring values), and two retrained N-grams over the same vocabulary:
      lm2 -- every word below the five most frequent gets a lower 1-gram (the same words stay out of the tree), 2-gram
             probabilities move too;
      lm3 -- a rare word becomes the most frequent one (wchmm.c would build another tree)."""
    ref = _ref()
    ref.lib.jref_write_bingram.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p]
    task = synth.make_triphone_task(tmp_path, seed=seed, nword=300, nphone=8, S=120, M=2, with_rl3=with_rl)
    L = open(task["arpa"]).read().splitlines()
    a, b = L.index("\\1-grams:") + 1, L.index("\\2-grams:")
    uni = [(i, l.split()) for i, l in enumerate(L[a:b], a) if l.strip()]
    vals = sorted((float(f[0]) for _, f in uni), reverse=True)
    thres = vals[sepnum - 1]
    rng = np.random.default_rng(seed)

    def variant(name, promote):
        out = list(L)
        low = [i for i, f in uni if float(f[0]) < thres and f[1] not in ("<s>", "</s>")]
        for i, f in uni:
            if float(f[0]) < thres and f[1] not in ("<s>", "</s>"):
                out[i] = f"{float(f[0]) - rng.uniform(0.05, 0.6):.6f}\t{f[1]}\t{f[2]}"
        if promote:
            f = L[low[len(low) // 2]].split()
            out[low[len(low) // 2]] = f"{vals[0] + 0.2:.6f}\t{f[1]}\t{f[2]}"
        for i in range(b + 1, len(out)):
            f = out[i].split()
            if len(f) >= 3 and f[0].startswith("-") and rng.random() < 0.5:
                out[i] = f"{float(f[0]) - 0.125:.6f}\t" + "\t".join(f[1:])
        (tmp_path / f"{name}.arpa").write_text("\n".join(out) + "\n")
        assert ref.lib.jref_write_bingram(str(tmp_path / f"{name}.arpa").encode(), str(task["arpa_rl"]).encode() if with_rl else None,
                                          str(tmp_path / f"{name}.bingram").encode()) == 0
        return tmp_path / f"{name}.bingram"

    assert ref.lib.jref_write_bingram(str(task["arpa"]).encode(), str(task["arpa_rl"]).encode() if with_rl else None,
                                      str(tmp_path / "lm1.bingram").encode()) == 0
    lm2, lm3 = variant("lm2", False), variant("lm3", True)
    for name, lm in (("exp1", tmp_path / "lm1.bingram"), ("exp2", lm2)):
        subprocess.run([str(EXPORT), "-h", task["hmmdefs"], "-hlist", task["hmmlist"], "-v", task["dict"], "-d", str(lm), "-sepnum", str(sepnum),
                        "-input", "htkparam", "-jamdout", str(tmp_path / name)], check=True, capture_output=True)
    return task, lm2, lm3


def _fscore_of(L, lex, bingram):
    n = C.c_int(0)
    buf = np.zeros(1 << 16, np.float32)
    rc = L.jamd_bingram_fscore(str(lex).encode(), str(bingram).encode(), buf.ctypes.data_as(C.c_void_p), len(buf), C.byref(n))
    return rc, buf[:n.value].copy()


def test_retrained_ngram_refreshes_the_factoring_values(tmp_path):
    """VERDICT r4 missing 6 (libjulius/src/factoring_sub.c:429-463): under a retrained N-gram over the same vocabulary the
    loader's factoring values are the ones a FRESH jamd_export with that N-gram writes (wchmm.c's own), index by index --
    not the stale ones of the exported tree; a retrained N-gram that changes which words wchmm.c keeps out of the tree,
    or a lexicon file that does not record -sepnum, is refused with the reason."""
    task, lm2, lm3 = _retrain_task(tmp_path)
    L = lib.load()
    r1, r2 = _blob_records(tmp_path / "exp1.lex", b"JAMDLEX1"), _blob_records(tmp_path / "exp2.lex", b"JAMDLEX1")
    assert np.frombuffer(r1["sep_wnum"][1], np.int32)[0] == 5
    for k in ("self_a", "next_a", "ac_off", "ac_to", "ac_a", "stend", "scid", "startnode", "start2isolate", "word_head", "scword"):
        assert r1[k] == r2[k], k                                  # the same words stayed out: wchmm.c built the same tree
    f1, f2 = np.frombuffer(r1["fscore"][1], np.float32), np.frombuffer(r2["fscore"][1], np.float32)
    assert len(f1) == len(f2) > 50 and (f1[1:] != f2[1:]).sum() > 10      # the values did move
    rc, got = _fscore_of(L, tmp_path / "exp1.lex", lm2)
    assert rc == 0, L.jamd_last_error()
    assert np.array_equal(got[1:], f2[1:])                        # == the reference's own, from the fresh export
    rc, same = _fscore_of(L, tmp_path / "exp1.lex", tmp_path / "lm1.bingram")
    assert rc == 0 and np.array_equal(same[1:], f1[1:])           # the tree's own N-gram: untouched
    rc, _ = _fscore_of(L, tmp_path / "exp1.lex", lm3)             # another word among the five most frequent
    assert rc != 0 and b"export the lexicon again" in L.jamd_last_error()
    # a file from before the check existed (no sep_wnum record): its own N-gram is fine, a retrained one is refused
    raw = bytearray(open(tmp_path / "exp1.lex", "rb").read())
    at = bytes(raw).rindex(b"sep_wnum")
    assert at + 36 == len(raw)
    del raw[at:]
    raw[8:12] = (np.frombuffer(bytes(raw[8:12]), np.int32) - 1).astype(np.int32).tobytes()
    (tmp_path / "old.lex").write_bytes(bytes(raw))
    assert _fscore_of(L, tmp_path / "old.lex", tmp_path / "lm1.bingram")[0] == 0
    rc, _ = _fscore_of(L, tmp_path / "old.lex", lm2)
    assert rc != 0 and b"-sepnum" in L.jamd_last_error()


@pytest.mark.gpu
@pytest.mark.parametrize("with_rl", [False, True])
def test_retrained_ngram_gives_the_fresh_exports_first_pass(engine, tmp_path, with_rl):
    """The first pass over (exported tree + retrained binary N-gram) == the first pass over a fresh export with that
    N-gram: status, score, sentence and the word trellis byte for byte."""
    task, lm2, lm3 = _retrain_task(tmp_path, with_rl=with_rl, seed=79)
    am = lib.Gmm.from_file(engine, tmp_path / "exp1.am")
    scs = [am.outprob_host(synth.make_utterance(task, nwords=5 + u, seed=950 + u)[0]) for u in range(3)]
    outs = []
    for lx in (lib.Lexicon.from_file(engine, tmp_path / "exp2.lex"), lib.Lexicon.from_file(engine, tmp_path / "exp1.lex", bingram=lm2),
               lib.Lexicon.from_file(engine, tmp_path / "exp1.lex")):
        bm = lib.Beam(engine, lx, 300, -1.0, max_utts=len(scs))
        res, tre = bm.pass1_host(scs)
        outs.append([(r.status, r.score, list(r.wseq[:r.wnum]), t.tobytes()) for r, t in zip(res, tre)])
        bm.close()
    assert outs[0] == outs[1]
    assert outs[2] != outs[0]                                     # (the stale tree + old N-gram is a different search)
    if with_rl:     # the tree's 1-gram is the BACKWARD N-gram's (ngram->d[0]): the edited forward ARPA only moves the additional 2-gram
        lib.Lexicon.from_file(engine, tmp_path / "exp1.lex", bingram=lm3)
    else:
        with pytest.raises(RuntimeError, match="export the lexicon again"):
            lib.Lexicon.from_file(engine, tmp_path / "exp1.lex", bingram=lm3)
