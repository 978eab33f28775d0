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
