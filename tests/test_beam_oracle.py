"""CPU: the first-pass oracle (oracle/jamd_oracle_beam.c) against
  (1) the committed golden fixtures the compiled reference produced
      (tests/make_golden.py beam), and
  (2) the compiled reference itself on fresh seeded tasks (when oracle/_ref is built).
Bit-exact: word ids, frame indices, predecessor links, float scores."""
import numpy as np
import pytest

from beamutil import assert_trellis_equal, load_beam_golden, ref_grammar_task, ref_task
from julius_amd import lexblob, synth


@pytest.mark.parametrize("name", ["beam_rank.npz", "beam_score.npz", "beam_isolated.npz",
                                  "beam_grammar.npz", "beam_grammar_free.npz", "beam_multipath.npz"])
def test_oracle_matches_golden(oracle, name):
    g = load_beam_golden(name)
    for u in g["utts"]:
        sc = oracle.gmm_outprob(g["am"], u["frames"])
        atoms, wseq, score, rc, died = oracle.beam_pass1(g["lex"], sc, g["beam_width"], g["score_pruning_width"])
        assert rc == 0 and died == -1
        assert_trellis_equal(atoms, u["trellis"])
        assert np.array_equal(wseq, u["wseq"])
        assert score == u["score"]


def test_lexblob_roundtrip(tmp_path):
    g = load_beam_golden("beam_rank.npz")
    lexblob.save(g["lex"], tmp_path / "x.blob")
    back = lexblob.load(tmp_path / "x.blob")
    for k, v in g["lex"].items():
        if isinstance(v, np.ndarray):
            assert np.array_equal(back[k], v), k
        else:
            assert back[k] == pytest.approx(v), k


def test_beam_death_is_reported(oracle):
    """All-LOG_ZERO acoustic scores kill every token: status 2 at frame 1
    (get_back_trellis_proceed() returning FALSE, beam.c:3012-3015)."""
    g = load_beam_golden("beam_score.npz")      # IWCD max: an all-LOG_ZERO set stays LOG_ZERO (no NaN)
    S = len(g["am"]["st_off"]) - 1
    sc = np.full((5, S), -1000000.0, np.float32)
    atoms, wseq, score, rc, died = oracle.beam_pass1(g["lex"], sc, 50)
    assert rc == 2 and died == 1 and len(wseq) == 0


@pytest.mark.parametrize("seed,beam,extra", [
    (11, 200, ["-sepnum", "5"]),
    (12, 25, ["-sepnum", "2"]),                                # very narrow rank beam
    (13, 150, ["-sepnum", "4", "-bs", "40"]),                  # score beam
    (14, 100, ["-sepnum", "0", "-iwcd1", "avg"]),              # whole vocabulary in the tree
    (15, 100, ["-sepnum", "3", "-iwcd1", "best", "2", "-lmp", "6.0", "-3.0"]),
    (16, 150, ["-sepnum", "4", "-transp", "-1.5"]),            # transparent words (task built with ntransparent=12)
    (17, 150, ["-sepnum", "4", "-unk"]),                       # 10 dictionary words outside the LM -> <unk>
    (18, 150, ["-sepnum", "4", "-rl3"]),                       # forward 2-gram + backward 3-gram: bi_prob_additional()
])
def test_oracle_matches_reference_live(oracle, ref, tmp_path, seed, beam, extra):
    kw = (dict(ntransparent=12) if "-transp" in extra else dict(nunk=10) if "-unk" in extra
          else dict(with_rl3=True) if "-rl3" in extra else {})
    extra = [x for x in extra if x not in ("-unk", "-rl3")]
    eng, lex, am, task = ref_task(ref, tmp_path, seed, beam, extra, **kw)
    if "with_rl3" in kw:
        assert lex["ng_mode"] == 2            # JAMD_NG_ADDITIONAL
    if "nunk" in kw:
        assert lex["ng_unk_id"] < lex["ng_nword"] and lex["ng_unk_num_log"] == pytest.approx(1.0)   # log10(10)
    if "-transp" in extra:
        assert int(np.sum(lex["is_transparent"])) == 12 and lex["lm_penalty_trans"] == -1.5
    bs = float(extra[extra.index("-bs") + 1]) if "-bs" in extra else -1.0
    for u in range(3):
        fr, _ = synth.make_utterance(task, nwords=3 + 2 * u, seed=100 * seed + u)
        synth.write_htk_param(tmp_path / "u.mfc", fr)
        tr, (wseq, score) = eng.recognize(tmp_path / "u.mfc")
        sc = oracle.gmm_outprob(am, fr)
        atoms, owseq, oscore, rc, died = oracle.beam_pass1(lex, sc, eng.beam_width, bs)
        assert rc in (0, 1)
        assert_trellis_equal(atoms, tr)
        if rc == 0:
            assert np.array_equal(owseq, wseq) and oscore == score


@pytest.mark.parametrize("seed,beam,extra,wrap", [
    (31, 120, ["-penalty1", "-2.0"], True),                              # <s> WORD+ </s>, N-best state sets (default)
    (32, 60, ["-iwcd1", "max", "-bs", "70"], True),
    (33, 40, ["-iwcd1", "avg", "-penalty1", "-1.0"], False),             # any word may start: 60+ initial tokens > beam
])
def test_oracle_matches_reference_grammar(oracle, ref, tmp_path, seed, beam, extra, wrap):
    """DFA grammar with per-category lexicon trees (beam.c grammar branches: init_nodescore()
    :1669-1757, beam_inter_word() category-pair test :2404-2412, find_1pass_result() :433-455)."""
    eng, lex, am, task = ref_grammar_task(ref, tmp_path, seed, beam, extra, wrap=wrap, nword=70)
    assert lex["lm_type"] == 1 and lex["ncat"] == 5 and (lex["ninit"] == 1) == wrap
    bs = float(extra[extra.index("-bs") + 1]) if "-bs" in extra else -1.0
    for u in range(3):
        fr, _ = synth.make_triphone_grammar_utterance(task, nwords=2 + 2 * u, seed=100 * seed + u)
        synth.write_htk_param(tmp_path / "u.mfc", fr)
        rtr, (rwseq, rscore) = eng.recognize(tmp_path / "u.mfc")
        sc = oracle.gmm_outprob(am, fr)
        atoms, wseq, score, rc, died = oracle.beam_pass1(lex, sc, eng.beam_width, bs)
        assert rc == 0
        assert_trellis_equal(atoms, rtr)
        assert np.array_equal(wseq, rwseq) and score == rscore


def test_oracle_matches_reference_c1_grammar(oracle, ref, tmp_path):
    """BASELINE configs[0] shape: tied-mixture monophones + 100-word loop grammar."""
    from oracle import pyoracle
    task = synth.make_grammar_task(tmp_path, seed=3)
    eng = pyoracle.RefEngine(ref, ["-h", task["hmmdefs"], "-dfa", task["dfa"], "-v", task["dict"], "-input", "htkparam",
                                   "-1pass", "-gprune", "safe", "-tmix", "2", "-b", "150"])
    eng.save_lexicon(tmp_path / "lex.blob")
    lex = lexblob.load(tmp_path / "lex.blob")
    am = ref.am_load(task["hmmdefs"], gprune="safe", gprune_num=2).export()
    for u in range(3):
        fr, _ = synth.make_grammar_utterance(task, nwords=2 + u, seed=u)
        synth.write_htk_param(tmp_path / "u.mfc", fr)
        rtr, (rwseq, rscore) = eng.recognize(tmp_path / "u.mfc")
        sc = oracle.gmm_outprob(am, fr, pyoracle.GPRUNE_SAFE, 2)
        atoms, wseq, score, rc, died = oracle.beam_pass1(lex, sc, eng.beam_width, -1.0)
        assert rc == 0
        assert_trellis_equal(atoms, rtr)
        assert np.array_equal(wseq, rwseq) and score == rscore


@pytest.mark.parametrize("triphone,multipath", [(True, False), (False, False), (True, True)])
def test_oracle_matches_reference_wordlist(oracle, ref, tmp_path, triphone, multipath):
    """Isolated word recognition (-w): every listed word starts with a token, no cross-word
    transition, best word on the last frame (beam.c:1762-1788, :2875, find_1pass_result_word())."""
    from oracle import pyoracle
    task = synth.make_wordlist_task(tmp_path, seed=5, triphone=triphone)
    args = ["-h", task["hmmdefs"]] + (["-hlist", task["hmmlist"]] if triphone else []) + [
        "-w", task["wordlist"], "-wsil", "silB", "silE", "silB", "-input", "htkparam", "-gprune", "none", "-b", "60"]
    eng = pyoracle.RefEngine(ref, args + (["-multipath"] if multipath else []))
    eng.save_lexicon(tmp_path / "lex.blob")
    lex = lexblob.load(tmp_path / "lex.blob")
    assert lex["lm_type"] == (0x102 if multipath else 2)
    am = ref.am_load(task["hmmdefs"], hmmlist=task["hmmlist"] if triphone else None).export()
    for u in range(4):
        fr, _ = synth.make_wordlist_utterance(task, seed=u)
        synth.write_htk_param(tmp_path / "u.mfc", fr)
        rtr, _ = eng.recognize(tmp_path / "u.mfc")
        st, fw, fs = eng.final_result()
        atoms, wseq, score, rc, died = oracle.beam_pass1(lex, oracle.gmm_outprob(am, fr), eng.beam_width, -1.0)
        assert rc == 0
        assert_trellis_equal(atoms, rtr)
        assert list(wseq) == list(fw) and score == fs


SKIP_TRANS = np.array([[0, 1, 0, 0, 0], [0, .5, .3, .2, 0], [0, 0, .5, .3, .2], [0, 0, 0, .6, .4], [0, 0, 0, 0, 0]])
SPLIT_TRANS = np.array([[0, .7, .3, 0, 0], [0, .5, .3, .2, 0], [0, 0, .5, .3, .2], [0, 0, 0, .6, .4], [0, 0, 0, 0, 0]])


@pytest.mark.parametrize("seed,beam,extra", [
    (41, 200, ["-sepnum", "5"]),
    (42, 30, ["-sepnum", "2"]),
    (43, 150, ["-sepnum", "4", "-bs", "40", "-iwcd1", "max"]),
    (44, 120, ["-sepnum", "0", "-iwcd1", "avg", "-transp", "-1.5"]),
    (45, 150, ["-sepnum", "4", "skip"]),       # state-skip and early-exit arcs: the model itself needs multipath
    (46, 150, ["-sepnum", "4", "split"]),      # two entry arcs as well
    (53, 150, ["-sepnum", "10", "-iwcd1", "max", "iwsp"]),   # -iwsp -spmodel sp (the reference's DNN recipe): a skippable pause behind every word
])
def test_oracle_matches_reference_multipath(oracle, ref, tmp_path, seed, beam, extra):
    """-multipath: non-emitting word-begin / word-end nodes, frame 0 through get_back_trellis_proceed(),
    the beam applied to the new tokens BEFORE the cross-word step and one transition-only call at the
    end (beam.c:2747-2836, :2930-2943, :3066-3073; pass1.c:239).  The device first pass does not take
    these lexicons yet; the oracle is pinned here for it."""
    kw = dict(ntransparent=12) if "-transp" in extra else {}
    if extra[-1] in ("skip", "split"):         # need_multipath is found by the loader (rdhmmdef.c:357): no -multipath
        kw["trans"] = SKIP_TRANS if extra[-1] == "skip" else SPLIT_TRANS
        extra = extra[:-1]
        eng, lex, am, task = ref_task(ref, tmp_path, seed, beam, list(extra), **kw)
        assert len(lex["ac_to"]) > 0
    elif extra[-1] == "iwsp":
        extra = extra[:-1]
        eng, lex, am, task = ref_task(ref, tmp_path, seed, beam, list(extra) + ["-multipath", "-iwsp", "-spmodel", "sp"], sp=True, nword=80, **kw)
        assert len(lex["ac_to"]) > 0
    else:
        eng, lex, am, task = ref_task(ref, tmp_path, seed, beam, list(extra) + ["-multipath"], **kw)
    assert eng.multipath == 1 and lex["lm_type"] == 0x100
    assert (lex["out_kind"] == 4).sum() >= lex["nword"]           # every word ends in a non-emitting node
    bs = float(extra[extra.index("-bs") + 1]) if "-bs" in extra else -1.0
    for u in range(3):
        fr, _ = synth.make_utterance(task, nwords=3 + 2 * u, seed=100 * seed + u)
        synth.write_htk_param(tmp_path / "u.mfc", fr)
        tr, (wseq, score) = eng.recognize(tmp_path / "u.mfc")
        sc = oracle.gmm_outprob(am, fr)
        atoms, owseq, oscore, rc, died = oracle.beam_pass1(lex, sc, eng.beam_width, bs)
        assert rc in (0, 1)
        assert_trellis_equal(atoms, tr)
        if rc == 0:
            assert np.array_equal(owseq, wseq) and oscore == score


@pytest.mark.parametrize("seed,beam,extra,wrap", [
    (51, 120, ["-penalty1", "-2.0"], True),
    (52, 40, ["-iwcd1", "avg", "-penalty1", "-1.0"], False),
])
def test_oracle_matches_reference_multipath_grammar(oracle, ref, tmp_path, seed, beam, extra, wrap):
    eng, lex, am, task = ref_grammar_task(ref, tmp_path, seed, beam, list(extra) + ["-multipath"], wrap=wrap, nword=70)
    assert lex["lm_type"] == 0x101
    for u in range(3):
        fr, _ = synth.make_triphone_grammar_utterance(task, nwords=2 + 2 * u, seed=100 * seed + u)
        synth.write_htk_param(tmp_path / "u.mfc", fr)
        rtr, (rwseq, rscore) = eng.recognize(tmp_path / "u.mfc")
        sc = oracle.gmm_outprob(am, fr)
        atoms, wseq, score, rc, died = oracle.beam_pass1(lex, sc, eng.beam_width, -1.0)
        assert rc == 0
        assert_trellis_equal(atoms, rtr)
        assert np.array_equal(wseq, rwseq) and score == rscore


@pytest.mark.parametrize("lm", ["grammar", "ngram"])
def test_reference_refuses_words_made_of_tee_models_only(tmp_path, lm):
    """The one lexicon kind the frame-parallel multipath frame leaves to the strict-order kernel -- a root that reaches a
    word-end node along its own arcs, i.e. a word made of tee models only (csrc/beam_exact_mp.h, jamd_lexicon::mp_parallel)
    -- is a lexicon the reference itself never builds: wchmm_add_word() rejects the word ("WORD SKIPPING TRANSITION NOT
    ALLOWED ... This type of word skipping is not supported", libjulius/src/wchmm.c:1345-1362), the tree is not built
    (m_fusion.c "error in bulding wchmm" with an N-gram; a grammar is left without its tree and a beam width of 0).
    Shown on the dictation kits' form of such a word: a pause word pronounced `sp`, with an `sp` model that has the
    entry -> exit skip."""
    import shutil
    import subprocess
    from oracle import pyoracle
    julius = pyoracle.HERE / "_ref" / "bin" / "julius"
    if not julius.exists() or shutil.which("stdbuf") is None:
        pytest.skip("oracle/_ref/bin/julius not built (or no stdbuf: the N-gram case ends in a crash that loses buffered output)")
    task = synth.make_triphone_task(tmp_path, seed=5, sp=True, nword=40, ntee=1)
    (tmp_path / "empty.list").write_text("")
    args = ["stdbuf", "-o0", "-e0", julius, "-h", task["hmmdefs"], "-hlist", task["hmmlist"], "-input", "htkparam",
            "-filelist", tmp_path / "empty.list", "-1pass", "-b", "50", "-multipath"]
    if lm == "grammar":
        g = synth.make_triphone_grammar(task, ncat=3, seed=5)
        args += ["-dfa", g["dfa"], "-v", g["gdict"]]
    else:
        args += ["-nlr", task["arpa"], "-v", task["dict"]]
    out = subprocess.run([str(a) for a in args], capture_output=True, text=True, timeout=120)
    log = out.stdout + out.stderr
    assert "WORD SKIPPING TRANSITION NOT ALLOWED" in log and "[T00])" in log and "failed to add word" in log
    if lm == "grammar":
        assert "trellis beam width = 0" in log                       # multigram_build() gave up before set_beam_width()
    else:
        assert "error in bulding wchmm" in log and out.returncode != 0
