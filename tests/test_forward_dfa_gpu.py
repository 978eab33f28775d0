"""Grammars with a FORWARD DFA beside the reversed one (the `.dfa.forward` file recent mkdfa.pl writes; VERDICT r4 missing 4):
tokens carry a state of it, an initial token takes the arc of its category out of the grammar's first state
(libjulius/src/beam.c:1739-1747), a cross-word transition takes the arc of the next word's category and is dropped when
there is none (:2412-2422), word-internal transitions inherit the state (:2120).

The grammar of synth.make_forward_grammar() bounds the sentence length -- something the category-pair test of the first pass
cannot see -- so the forward automaton really cuts transitions.  The device's first pass (exact-order kernel, and the
strict-order kernel as the second implementation; multipath lexicons -- `-multipath` -- through their own frame) must give the compiled reference's word trellis, sentence and score on
sentences inside the language and on category chains that are too long for it; the canonical-tie kernel refuses such a lexicon."""
import numpy as np
import pytest

from beamutil import assert_trellis_equal
from julius_amd import lexblob, lib, synth
from oracle import pyoracle

pytestmark = pytest.mark.gpu


def _task(ref, tmp_path, seed, beam, nword=80, maxwords=3, multipath=False):
    task = synth.make_forward_grammar(synth.make_triphone_task(tmp_path, seed=seed, nword=nword), ncat=3, maxwords=maxwords, seed=seed)
    eng = pyoracle.RefEngine(ref, [str(a) for a in ["-h", task["hmmdefs"], "-hlist", task["hmmlist"], "-dfa", task["dfa"], "-v", task["gdict"],
                                                      "-input", "htkparam", "-1pass", "-gprune", "none", "-b", beam]] + (["-multipath"] if multipath else []))
    eng.save_lexicon(tmp_path / "lex.blob")
    lex = lexblob.load(tmp_path / "lex.blob")
    am = ref.am_load(task["hmmdefs"], task["hmmlist"]).export()
    return eng, lex, am, task


@pytest.mark.parametrize("multipath", [False, True], ids=["plain", "multipath"])
@pytest.mark.parametrize("seed,beam", [(11, 200), (12, 40), (13, 12)])
def test_forward_dfa_first_pass_equals_compiled_reference(engine, oracle, ref, tmp_path, seed, beam, multipath):
    eng, lex, am, task = _task(ref, tmp_path, seed, beam, multipath=multipath)
    assert bool(lex["lm_type"] & 0x100) == multipath
    assert lex["nfwd"] > 5 and len(lex["fwd_to"]) == lex["fwd_off"][-1] and len(lex["init_to_state"]) == lex["ninit"]
    utts = [synth.make_forward_grammar_utterance(task, seed=100 * seed + u, nwords=None if u < 3 else 4 + u)[0] for u in range(6)]
    scores = [oracle.gmm_outprob(am, fr) for fr in utts]
    want = []
    for fr in utts:
        synth.write_htk_param(tmp_path / "u.mfc", fr)
        want.append(eng.recognize(tmp_path / "u.mfc"))
    for lx in (lib.Lexicon(engine, lex), lib.Lexicon.from_file(engine, tmp_path / "lex.blob")):      # descriptor and file loader
        for mode in ("exact", "strict"):
            bm = lib.Beam(engine, lx, beam, -1.0, max_utts=len(utts), atoms_per_utt=1 << 16)
            assert bm.order_mode() == "exact"
            if mode == "strict":
                bm.set_order_mode("strict")
            res, tre = bm.pass1_host(scores)
            found = 0
            for r, atoms, (rtr, (rw, rs)) in zip(res, tre, want):
                assert_trellis_equal(atoms, rtr)
                if len(rw):
                    found += 1
                    assert r.status == 0 and list(r.wseq[:r.wnum]) == list(rw) and r.score == rs, mode
                else:
                    assert r.status != 0
            assert found >= 3 or beam < 40
            with pytest.raises(RuntimeError, match="forward DFA"):
                bm.set_order_mode("fast")
            bm.close()
    # the automaton did something: the same lexicon with the forward DFA taken out decodes the long chains differently
    plain = dict(lex, nfwd=0)
    bm = lib.Beam(engine, lib.Lexicon(engine, plain), beam, -1.0, max_utts=len(utts), atoms_per_utt=1 << 16)
    res2, tre2 = bm.pass1_host(scores)
    assert any(len(a) != len(b) for a, b in zip(tre, tre2))
    bm.close()


def test_forward_dfa_streaming_and_batch(engine, oracle, ref, tmp_path):
    """The state travels in the token record: input pushed in ragged chunks gives the one-shot trellis."""
    eng, lex, am, task = _task(ref, tmp_path, 21, 60, maxwords=4)
    utts = [synth.make_forward_grammar_utterance(task, seed=2100 + u, nwords=None if u % 2 else 6)[0] for u in range(3)]
    scores = [oracle.gmm_outprob(am, fr) for fr in utts]
    lx = lib.Lexicon(engine, lex)
    one = lib.Beam(engine, lx, 60, -1.0, max_utts=3, atoms_per_utt=1 << 16)
    res1, tre1 = one.pass1_host(scores)
    S = scores[0].shape[1]
    bm = lib.Beam(engine, lx, 60, -1.0, max_utts=3, atoms_per_utt=1 << 16)
    bm.stream_begin(3)
    pos = [0, 0, 0]
    chunks = [1, 7, 0, 19, 10000]
    for ci, c in enumerate(chunks):
        part, off = [], [0]
        for u, sc in enumerate(scores):
            n = min(len(sc) - pos[u], c + 3 * u if c else 0)
            part.append(sc[pos[u]:pos[u] + n]); pos[u] += n; off.append(off[-1] + n)
        rows = np.concatenate(part) if off[-1] else np.zeros((1, S), np.float32)
        d = lib.DevBuf(engine, rows.nbytes).upload(rows)
        bm.stream_push_dev(d.ptr, S, np.array(off, np.int32), final=ci == len(chunks) - 1)
        bm.results(3)
        d.free()
    for u, r in enumerate(bm.results(3)):
        assert (r.status, r.natom, r.score) == (res1[u].status, res1[u].natom, res1[u].score)
        a, b = lexblob.canonical_trellis(bm.trellis(u)), lexblob.canonical_trellis(tre1[u])
        assert all(np.array_equal(a[k], b[k]) for k in a)
    synth.write_htk_param(tmp_path / "u.mfc", utts[0])
    rtr, _ = eng.recognize(tmp_path / "u.mfc")
    assert_trellis_equal(tre1[0], rtr)
    one.close(); bm.close()
