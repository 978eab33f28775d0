"""GPU: the HIP first pass (beam.hip, through the C ABI) against the golden
fixtures of the compiled reference, the oracle, and -- when oracle/_ref is
present -- the reference recogniser itself.  Bit-exact trellis: word ids,
begin/end frames, predecessor links and float scores.  (jamd_pass1_result.ties
counts exact float ties, the only place where the engine's canonical rule and the
reference's visiting order could pick different -- equally scored -- histories;
these fixtures contain a few and still match.)"""
import numpy as np
import pytest

from beamutil import (assert_grammar_fast, assert_trellis_equal, assert_trellis_equal_modulo_ties, load_beam_golden, ref_grammar_task,
                      ref_task)
from julius_amd import lib, synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["beam_rank.npz", "beam_score.npz", "beam_isolated.npz"])
def test_golden_batch(engine, oracle, name):
    """All utterances of a fixture in ONE launch (one workgroup each)."""
    g = load_beam_golden(name)
    lx = lib.Lexicon(engine, g["lex"])
    bm = lib.Beam(engine, lx, g["beam_width"], g["score_pruning_width"], max_utts=len(g["utts"]))
    scores = [oracle.gmm_outprob(g["am"], u["frames"]) for u in g["utts"]]
    res, tre = bm.pass1_host(scores)
    for r, atoms, u in zip(res, tre, g["utts"]):
        assert r.status == 0 and r.frames == len(u["frames"])
        assert_trellis_equal(atoms, u["trellis"])
        assert np.array_equal(np.array(r.wseq[:r.wnum]), u["wseq"])
        assert r.score == u["score"]


def test_end_to_end_device_flow(engine):
    """HIP GMM scores stay on the device and feed the HIP beam: frames in,
    pass-1 sentence out, identical to the reference's."""
    g = load_beam_golden("beam_rank.npz")
    gm = lib.Gmm(engine, g["am"])
    lx = lib.Lexicon(engine, g["lex"])
    bm = lib.Beam(engine, lx, g["beam_width"], g["score_pruning_width"], max_utts=len(g["utts"]))
    frames = np.concatenate([u["frames"] for u in g["utts"]])
    off = np.zeros(len(g["utts"]) + 1, np.int32)
    off[1:] = np.cumsum([len(u["frames"]) for u in g["utts"]])
    d_fr = lib.DevBuf(engine, frames.nbytes).upload(frames)
    d_sc = lib.DevBuf(engine, 4 * len(frames) * gm.S)
    gm.outprob_dev(d_fr.ptr, len(frames), d_sc.ptr)
    bm.pass1_dev(d_sc.ptr, gm.S, off)
    res = bm.results()
    for i, (r, u) in enumerate(zip(res, g["utts"])):
        assert r.status == 0
        assert np.array_equal(np.array(r.wseq[:r.wnum]), u["wseq"])
        assert r.score == u["score"]
        assert_trellis_equal(bm.trellis(i), u["trellis"])


def test_beam_death(engine):
    g = load_beam_golden("beam_score.npz")      # IWCD max: an all-LOG_ZERO set stays LOG_ZERO (no NaN)
    S = len(g["am"]["st_off"]) - 1
    lx = lib.Lexicon(engine, g["lex"])
    bm = lib.Beam(engine, lx, 50, -1.0, max_utts=1)
    res, _ = bm.pass1_host([np.full((5, S), -1000000.0, np.float32)])
    assert res[0].status == 2 and res[0].died_at == 1


def test_atom_overflow_is_reported(engine, oracle):
    g = load_beam_golden("beam_rank.npz")
    lx = lib.Lexicon(engine, g["lex"])
    bm = lib.Beam(engine, lx, g["beam_width"], -1.0, max_utts=1, atoms_per_utt=50)
    res, _ = bm.pass1_host([oracle.gmm_outprob(g["am"], g["utts"][0]["frames"])])
    assert res[0].status == 3


def test_work_area_is_reusable(engine, oracle):
    """Two batches through the same jamd_beam: the node table must come back clean."""
    g = load_beam_golden("beam_score.npz")
    lx = lib.Lexicon(engine, g["lex"])
    bm = lib.Beam(engine, lx, g["beam_width"], g["score_pruning_width"], max_utts=2)
    sc = [oracle.gmm_outprob(g["am"], u["frames"]) for u in g["utts"]]
    for pick in ([0, 1], [2, 0], [1]):
        res, tre = bm.pass1_host([sc[i] for i in pick])
        for r, atoms, i in zip(res, tre, pick):
            assert r.status == 0
            assert_trellis_equal(atoms, g["utts"][i]["trellis"])


@pytest.mark.parametrize("seed,beam,extra,task_kw", [
    (21, 300, ["-sepnum", "5"], {}),
    (22, 30, ["-sepnum", "2"], {}),
    (23, 600, ["-sepnum", "10", "-bs", "80"], dict(nword=300, nphone=12, S=200)),   # beam > 512 threads
    (24, 100, ["-sepnum", "0", "-iwcd1", "avg"], {}),
    (25, 2000, ["-sepnum", "20"], dict(nword=600, nphone=14, S=260, M=2)),          # no rank pruning at all
    (26, 150, ["-sepnum", "4", "-transp", "-1.5"], dict(ntransparent=12)),          # transparent words
    (27, 150, ["-sepnum", "4"], dict(nunk=10)),                                     # words outside the LM -> <unk>
    (28, 150, ["-sepnum", "4"], dict(with_rl3=True)),                               # LR 2-gram + RL 3-gram (additional area)
    (29, 5000, ["-sepnum", "20"], dict(nword=600, nphone=14, S=260, M=2)),          # survivor image too large for LDS: global path
    (30, 1200, ["-sepnum", "10", "-bs", "90"], dict(nword=500, nphone=12, S=200)),  # LDS survivors, but no room for the cell table
])
def test_vs_oracle_and_reference_live(engine, oracle, ref, tmp_path, seed, beam, extra, task_kw):
    eng, lex, am, task = ref_task(ref, tmp_path, seed, beam, extra, **task_kw)
    bs = float(extra[extra.index("-bs") + 1]) if "-bs" in extra else -1.0
    utts = [synth.make_utterance(task, nwords=2 + 3 * u, seed=100 * seed + u)[0] for u in range(4)]
    scores = [oracle.gmm_outprob(am, fr) for fr in utts]
    lx = lib.Lexicon(engine, lex)
    bm = lib.Beam(engine, lx, eng.beam_width, bs, max_utts=len(utts), atoms_per_utt=1 << 19)
    res, tre = bm.pass1_host(scores)
    for fr, sc, r, atoms in zip(utts, scores, res, tre):
        oatoms, owseq, oscore, rc, died = oracle.beam_pass1(lex, sc, eng.beam_width, bs)
        assert r.status == rc
        from julius_amd import lexblob
        assert_trellis_equal_modulo_ties(atoms, lexblob.canonical_trellis(oatoms), r.ties)
        synth.write_htk_param(tmp_path / "u.mfc", fr)
        rtr, (rwseq, rscore) = eng.recognize(tmp_path / "u.mfc")
        assert_trellis_equal_modulo_ties(atoms, rtr, r.ties)
        if rc == 0:
            assert np.array_equal(np.array(r.wseq[:r.wnum]), rwseq) and r.score == rscore


def test_dnn_scores_feed_the_beam(engine, oracle):
    """configs[3] flow at a small size: MFMA DNN scores stay on the device and drive
    the first pass; compared with the oracle DNN + oracle first pass (bit-exact scores,
    identical trellis).  The lexicon is the synthetic one (julius_amd.synth.make_lexicon)."""
    from julius_amd import lexblob
    dnn = synth.make_dnn(dims=(48, 64, 64, 120), seed=5)
    lex = synth.make_lexicon(nword=300, nphone=10, S=120, seed=5, sepnum=10)
    rng = np.random.default_rng(5)
    utts = [rng.normal(0, 1, (T, 48)).astype(np.float32) for T in (40, 75)]
    net = lib.Dnn(engine, dnn)
    lx = lib.Lexicon(engine, lex)
    bm = lib.Beam(engine, lx, 150, -1.0, max_utts=2)
    frames = np.concatenate(utts)
    off = np.array([0, 40, 115], np.int32)
    d_fr = lib.DevBuf(engine, frames.nbytes).upload(frames)
    d_sc = lib.DevBuf(engine, 4 * len(frames) * net.S)
    net.outprob_dev(d_fr.ptr, len(frames), d_sc.ptr)
    bm.pass1_dev(d_sc.ptr, net.S, off)
    res = bm.results()
    for u, fr in enumerate(utts):
        sc = oracle.dnn_outprob(dnn, fr)
        oatoms, owseq, oscore, rc, died = oracle.beam_pass1(lex, sc, 150, -1.0)
        assert res[u].status == rc
        assert_trellis_equal_modulo_ties(bm.trellis(u), lexblob.canonical_trellis(oatoms), res[u].ties)
        if rc == 0:
            assert list(res[u].wseq[:res[u].wnum]) == list(owseq) and res[u].score == oscore


def test_full_size_lexicon_vs_oracle(engine, oracle):
    """BASELINE-size lexicon (20 000 words, ~254k nodes, beam 800): the device first
    pass against the CPU restatement on the same synthetic task."""
    from julius_amd import lexblob
    S = 3000
    lex = synth.make_lexicon(nword=20000, nphone=40, S=S, seed=0)
    model = synth.make_gmm(S=S, M=2, D=39, seed=0)
    fr, ws = synth.make_lexicon_utterance(lex, model, nwords=6, seed=3)
    sc = oracle.gmm_outprob(model, fr)
    lx = lib.Lexicon(engine, lex)
    bm = lib.Beam(engine, lx, 800, -1.0, max_utts=1)
    res, tre = bm.pass1_host([sc])
    oatoms, owseq, oscore, rc, died = oracle.beam_pass1(lex, sc, 800, -1.0)
    assert res[0].status == rc == 0
    assert list(res[0].wseq[:res[0].wnum]) == list(owseq) == ws and res[0].score == oscore
    assert_trellis_equal_modulo_ties(tre[0], lexblob.canonical_trellis(oatoms), res[0].ties_node + res[0].ties_cut + res[0].ties_wordend)


def test_full_size_reference_built_lexicon_vs_compiled_reference(engine, ref, tmp_path):
    """BASELINE configs[2] at full size with nothing python-made between the files and the search: dictionary
    (20 000 words) + ARPA 2-gram + tied-state triphone hmmdefs (3000 states) are loaded by the reference's
    own readers, the tree lexicon is built by `libjulius/src/wchmm.c:1749 build_wchmm2`, flattened by the
    shim, and the device first pass (HIP GMM scores -> exact-order kernel) must give the word trellis of
    the compiled reference's `julius -1pass` on the same parameter files, entry by entry."""
    eng, lex, am, task = ref_task(ref, tmp_path, 0, 800, [], nphone=40, S=3000, M=16, nword=20000, nvar=25, maxlen=8,
                                  nbigram_per_word=10)
    assert lex["nnode"] > 200000 and eng.beam_width == 800
    utts = [synth.make_utterance(task, nwords=4 + 5 * u, seed=7000 + u)[0] for u in range(4)]
    gm = lib.Gmm(engine, am)
    scores = [gm.outprob_host(fr) for fr in utts]
    lx = lib.Lexicon(engine, lex)
    bm = lib.Beam(engine, lx, 800, -1.0, max_utts=len(utts), atoms_per_utt=1 << 17)
    assert bm.order_mode() == "exact"
    res, tre = bm.pass1_host(scores)
    bm.set_order_mode("fast")
    resf, tref = bm.pass1_host(scores)
    for fr, r, atoms, rf in zip(utts, res, tre, resf):
        synth.write_htk_param(tmp_path / "u.mfc", fr)
        rtr, (rwseq, rscore) = eng.recognize(tmp_path / "u.mfc")
        assert r.status == 0 and len(rtr["wid"]) > 1000
        assert_trellis_equal(atoms, rtr)                                   # exact, ties included
        assert np.array_equal(np.array(r.wseq[:r.wnum]), rwseq) and r.score == rscore
        assert rf.status == 0 and rf.score == rscore                       # canonical-tie kernel: same best path score


def test_full_size_batch_properties(engine):
    """Size-independent properties at BASELINE size (20 000-word lexicon, beam 800, a batch of 24
    utterances scored by the HIP GMM kernel on the device): the result of an utterance does not
    depend on the batch it is in, on its position, or on the run (no race in the LDS cell table,
    the atomics or the slot allocation); streaming in chunks equals the one-shot call."""
    from julius_amd import lexblob
    S = 3000
    lex = synth.make_lexicon(nword=20000, nphone=40, S=S, seed=0)
    model = synth.make_gmm(S=S, M=2, D=39, seed=0)
    utts = [synth.make_lexicon_utterance(lex, model, nwords=3 + u % 6, seed=40 + u)[0] for u in range(24)]
    gm = lib.Gmm(engine, model)
    scores = [gm.outprob_host(fr) for fr in utts]
    lx = lib.Lexicon(engine, lex)
    bm = lib.Beam(engine, lx, 800, -1.0, max_utts=len(utts))
    res1, tre1 = bm.pass1_host(scores)
    assert all(r.status == 0 for r in res1)
    canon1 = [lexblob.canonical_trellis(t) for t in tre1]
    perm = list(np.random.default_rng(1).permutation(len(utts)))
    res2, tre2 = bm.pass1_host([scores[i] for i in perm])          # other order, same work area
    for k, i in enumerate(perm):
        a, b = res1[i], res2[k]
        # (the tie COUNTERS may differ between runs: a tie between two losing candidates is only seen
        # when they meet before the winner arrives; a tie between the two best is always seen)
        assert (a.status, a.natom, a.wnum, a.score) == (b.status, b.natom, b.wnum, b.score)
        c2 = lexblob.canonical_trellis(tre2[k])
        assert all(np.array_equal(canon1[i][key], c2[key]) for key in canon1[i])
    solo = lib.Beam(engine, lx, 800, -1.0, max_utts=1)             # alone in its own work area
    r3, t3 = solo.pass1_host([scores[5]])
    assert (r3[0].natom, r3[0].score) == (res1[5].natom, res1[5].score)
    c3 = lexblob.canonical_trellis(t3[0])
    assert all(np.array_equal(canon1[5][key], c3[key]) for key in c3)
    # streaming, 37 frames at a time
    bm.stream_begin(len(scores))
    done = [0] * len(scores)
    while any(d < len(sc) for d, sc in zip(done, scores)):
        take = [min(37, len(sc) - d) for d, sc in zip(done, scores)]
        rows = np.concatenate([sc[d:d + k] for sc, d, k in zip(scores, done, take)])
        off = np.zeros(len(scores) + 1, np.int32)
        off[1:] = np.cumsum(take)
        done = [d + k for d, k in zip(done, take)]
        buf = lib.DevBuf(engine, rows.nbytes).upload(rows)
        bm.stream_push_dev(buf.ptr, S, off, final=all(d >= len(sc) for d, sc in zip(done, scores)))
        buf.free()
    for i, r in enumerate(bm.results(len(scores))):
        assert (r.status, r.natom, r.wnum, r.score) == (res1[i].status, res1[i].natom, res1[i].wnum, res1[i].score)
        c4 = lexblob.canonical_trellis(bm.trellis(i))
        assert all(np.array_equal(canon1[i][key], c4[key]) for key in c4)


@pytest.mark.parametrize("name", ["beam_rank.npz", "beam_score.npz", "beam_isolated.npz",
                                  "beam_grammar.npz", "beam_grammar_free.npz"])
@pytest.mark.parametrize("mode", ["strict", "exact", "exact_serial"])
def test_strict_order_golden(engine, oracle, name, mode):
    """The modes that follow the reference's visiting order -- strict (sequential, one lane per
    utterance), exact (frame-parallel, beam_exact.hip) and exact with the heap's extraction loop run
    sequentially: EXACT equality with the reference's golden trellis, no tie caveat."""
    g = load_beam_golden(name)
    lx = lib.Lexicon(engine, g["lex"])
    bm = lib.Beam(engine, lx, g["beam_width"], g["score_pruning_width"], max_utts=len(g["utts"]))
    bm.set_order_mode(mode)
    res, tre = bm.pass1_host([oracle.gmm_outprob(g["am"], u["frames"]) for u in g["utts"]])
    for r, atoms, u in zip(res, tre, g["utts"]):
        assert r.status == 0
        assert_trellis_equal(atoms, u["trellis"])
        assert np.array_equal(np.array(r.wseq[:r.wnum]), u["wseq"]) and r.score == u["score"]


@pytest.mark.parametrize("seed,beam,extra,task_kw", [
    (22, 30, ["-sepnum", "2"], {}),
    (24, 100, ["-sepnum", "0", "-iwcd1", "avg"], {}),
    (23, 600, ["-sepnum", "10", "-bs", "80"], dict(nword=300, nphone=12, S=200)),
    (25, 2000, ["-sepnum", "20"], dict(nword=600, nphone=14, S=260, M=2)),   # the case where the fast kernel's
    (26, 150, ["-sepnum", "4", "-transp", "-1.5"], dict(ntransparent=12)),   # tie rule dropped one atom of 34 842;
])                                                                              # transparent words
@pytest.mark.parametrize("mode", ["strict", "exact"])
def test_strict_order_exact_vs_reference_live(engine, oracle, ref, tmp_path, seed, beam, extra, task_kw, mode):
    eng, lex, am, task = ref_task(ref, tmp_path, seed, beam, extra, **task_kw)
    bs = float(extra[extra.index("-bs") + 1]) if "-bs" in extra else -1.0
    utts = [synth.make_utterance(task, nwords=2 + 3 * u, seed=100 * seed + u)[0] for u in range(4)]
    scores = [oracle.gmm_outprob(am, fr) for fr in utts]
    lx = lib.Lexicon(engine, lex)
    bm = lib.Beam(engine, lx, eng.beam_width, bs, max_utts=len(utts), atoms_per_utt=1 << 17)
    bm.set_order_mode(mode)
    res, tre = bm.pass1_host(scores)
    for fr, r, atoms in zip(utts, res, tre):
        synth.write_htk_param(tmp_path / "u.mfc", fr)
        rtr, (rwseq, rscore) = eng.recognize(tmp_path / "u.mfc")
        assert_trellis_equal(atoms, rtr)                      # exact, ties included
        if r.status == 0:
            assert np.array_equal(np.array(r.wseq[:r.wnum]), rwseq) and r.score == rscore


@pytest.mark.parametrize("beam", [1, 7, 120])
def test_edge_lengths_in_one_batch(engine, oracle, beam):
    """Utterances of 1, 2, 3 frames, a normal one and beam widths down to 1 in one launch
    (get_back_trellis_init() alone, one _proceed(), ...), against the oracle."""
    from julius_amd import lexblob
    g = load_beam_golden("beam_rank.npz")
    sc_full = oracle.gmm_outprob(g["am"], g["utts"][0]["frames"])
    scores = [sc_full[:1], sc_full[:2], sc_full[:3], sc_full]
    lx = lib.Lexicon(engine, g["lex"])
    bm = lib.Beam(engine, lx, beam, -1.0, max_utts=len(scores))
    for mode in ("fast", "strict", "exact"):
        strict = mode != "fast"
        bm.set_order_mode(mode)
        res, tre = bm.pass1_host(scores)
        for sc, r, atoms in zip(scores, res, tre):
            oatoms, owseq, oscore, rc, died = oracle.beam_pass1(g["lex"], sc, beam, -1.0)
            assert r.status == rc and r.frames == len(sc)
            if strict:
                assert_trellis_equal(atoms, lexblob.canonical_trellis(oatoms))
            else:
                assert_trellis_equal_modulo_ties(atoms, lexblob.canonical_trellis(oatoms), r.ties)
            if rc == 0:
                assert list(r.wseq[:r.wnum]) == list(owseq) and r.score == oscore


@pytest.mark.parametrize("chunks", [[1] * 40 + [10000], [7, 1, 50, 0, 3, 10000], [10000]])
def test_streaming_equals_one_shot(engine, oracle, chunks):
    """jamd_beam_stream_*: the utterances arrive in pieces (frame by frame, ragged chunks,
    empty pushes); the word trellis and the pass-1 result equal the one-shot call's."""
    g = load_beam_golden("beam_score.npz")
    scores = [oracle.gmm_outprob(g["am"], u["frames"]) for u in g["utts"]]
    S = scores[0].shape[1]
    lx = lib.Lexicon(engine, g["lex"])
    one = lib.Beam(engine, lx, g["beam_width"], g["score_pruning_width"], max_utts=len(scores))
    res1, tre1 = one.pass1_host(scores)
    bm = lib.Beam(engine, lx, g["beam_width"], g["score_pruning_width"], max_utts=len(scores))
    bm.stream_begin(len(scores))
    pos = [0] * len(scores)
    for ci, c in enumerate(chunks):
        # utterances advance by different amounts: utterance u gets c + u frames (until it runs out)
        part, off = [], [0]
        for u, sc in enumerate(scores):
            n = min(len(sc) - pos[u], (c + u) if c else 0)
            part.append(sc[pos[u]:pos[u] + n]); pos[u] += n; off.append(off[-1] + n)
        final = ci == len(chunks) - 1
        rows = np.concatenate(part) if off[-1] else np.zeros((1, S), np.float32)
        d = lib.DevBuf(engine, rows.nbytes).upload(rows)
        bm.stream_push_dev(d.ptr, S, np.array(off, np.int32), final=final)
        mid = bm.results(len(scores))
        assert all(r.frames == p for r, p in zip(mid, pos))
        d.free()
    res2 = bm.results(len(scores))
    for u in range(len(scores)):
        a, b = res1[u], res2[u]
        assert (a.status, a.natom, a.wnum, a.score, a.frames) == (b.status, b.natom, b.wnum, b.score, b.frames)
        assert list(a.wseq[:a.wnum]) == list(b.wseq[:b.wnum])
        from julius_amd import lexblob
        assert_trellis_equal(bm.trellis(u), lexblob.canonical_trellis(tre1[u]))
        assert_trellis_equal(bm.trellis(u), g["utts"][u]["trellis"])


@pytest.mark.parametrize("name", ["beam_grammar.npz", "beam_grammar_free.npz"])
def test_grammar_golden_batch(engine, oracle, name):
    """DFA grammar, one lexicon tree per category: initial tokens from every sentence-initial
    word, category-pair constraint at every word boundary, best word on the last frame."""
    g = load_beam_golden(name)
    lx = lib.Lexicon(engine, g["lex"])
    bm = lib.Beam(engine, lx, g["beam_width"], g["score_pruning_width"], max_utts=len(g["utts"]))
    scores = [oracle.gmm_outprob(g["am"], u["frames"]) for u in g["utts"]]
    res, tre = bm.pass1_host(scores)
    for r, atoms, u in zip(res, tre, g["utts"]):
        assert r.status == 0 and r.frames == len(u["frames"])
        assert_grammar_fast(atoms, u["trellis"], r, u["wseq"], u["score"])
    # the same utterances pushed in pieces
    lens = [len(x) for x in scores]
    bm.stream_begin(len(scores))
    done = [0] * len(scores)
    while any(d < n for d, n in zip(done, lens)):
        take = [min(13, n - d) for d, n in zip(done, lens)]
        rows = np.concatenate([sc[d:d + k] for sc, d, k in zip(scores, done, take)])
        off = np.zeros(len(scores) + 1, np.int32)
        off[1:] = np.cumsum(take)
        done = [d + k for d, k in zip(done, take)]
        buf = lib.DevBuf(engine, max(rows.nbytes, 4)).upload(rows)
        bm.stream_push_dev(buf.ptr, rows.shape[1], off, final=all(d >= n for d, n in zip(done, lens)))
    for i, (r, r0, atoms) in enumerate(zip(bm.results(), res, tre)):
        assert (r.status, r.wnum, r.score, r.natom) == (r0.status, r0.wnum, r0.score, r0.natom)
        from julius_amd import lexblob
        assert_trellis_equal(bm.trellis(i), lexblob.canonical_trellis(atoms))   # emission order within a frame is free


@pytest.mark.parametrize("seed,beam,extra,wrap", [
    (41, 150, ["-penalty1", "-2.0"], True),
    (42, 50, ["-iwcd1", "max", "-bs", "70"], False),
    (43, 1500, ["-iwcd1", "avg"], False),                      # no rank pruning
])
@pytest.mark.parametrize("mode", ["fast", "strict", "exact"])
def test_grammar_vs_reference_live(engine, oracle, ref, tmp_path, seed, beam, extra, wrap, mode):
    strict = mode != "fast"
    eng, lex, am, task = ref_grammar_task(ref, tmp_path, seed, beam, extra, wrap=wrap, nword=90)
    bs = float(extra[extra.index("-bs") + 1]) if "-bs" in extra else -1.0
    utts = [synth.make_triphone_grammar_utterance(task, nwords=2 + 2 * u, seed=100 * seed + u)[0] for u in range(4)]
    scores = [oracle.gmm_outprob(am, fr) for fr in utts]
    lx = lib.Lexicon(engine, lex)
    bm = lib.Beam(engine, lx, eng.beam_width, bs, max_utts=len(utts), atoms_per_utt=1 << 16)
    bm.set_order_mode(mode)
    res, tre = bm.pass1_host(scores)
    for fr, r, atoms in zip(utts, res, tre):
        synth.write_htk_param(tmp_path / "u.mfc", fr)
        rtr, (rwseq, rscore) = eng.recognize(tmp_path / "u.mfc")
        assert r.status == 0
        if strict:
            assert_trellis_equal(atoms, rtr)
            assert np.array_equal(np.array(r.wseq[:r.wnum]), rwseq) and r.score == rscore
        else:
            assert_grammar_fast(atoms, rtr, r, rwseq, rscore)


@pytest.mark.parametrize("mode", ["fast", "strict", "exact"])
@pytest.mark.parametrize("triphone", [True, False])
def test_wordlist_vs_reference_live(engine, oracle, ref, tmp_path, triphone, mode):
    """Isolated word recognition (-w word list) on the device."""
    strict = mode != "fast"
    from oracle import pyoracle
    task = synth.make_wordlist_task(tmp_path, seed=7, triphone=triphone, nword=80)
    args = ["-h", task["hmmdefs"]] + (["-hlist", task["hmmlist"]] if triphone else []) + [
        "-w", task["wordlist"], "-wsil", "silB", "silE", "silB", "-input", "htkparam", "-gprune", "none", "-b", "80"]
    eng = pyoracle.RefEngine(ref, args)
    eng.save_lexicon(tmp_path / "lex.blob")
    from julius_amd import lexblob
    lex = lexblob.load(tmp_path / "lex.blob")
    am = ref.am_load(task["hmmdefs"], hmmlist=task["hmmlist"] if triphone else None).export()
    utts = [synth.make_wordlist_utterance(task, seed=u)[0] for u in range(5)]
    lx = lib.Lexicon(engine, lex)
    bm = lib.Beam(engine, lx, eng.beam_width, -1.0, max_utts=len(utts))
    bm.set_order_mode(mode)
    res, tre = bm.pass1_host([oracle.gmm_outprob(am, fr) for fr in utts])
    for fr, r, atoms in zip(utts, res, tre):
        synth.write_htk_param(tmp_path / "u.mfc", fr)
        rtr, _ = eng.recognize(tmp_path / "u.mfc")
        st, fw, fs = eng.final_result()
        assert r.status == 0 and r.wnum == 1
        if strict:
            assert_trellis_equal(atoms, rtr)
            assert [r.wseq[0]] == list(fw) and r.score == fs
        else:       # branches whose logical triphones share a physical model tie exactly: see assert_grammar_fast
            assert_grammar_fast(atoms, rtr, r, np.array(fw), fs)


# ---- multipath lexicons ------------------------------------------------------------------------------
# beam_strict_mp_kernel (strict-order only): first hardware run at the start of round 2
# (profiles/r02a_pending_multipath_tests.txt); the CPU restatement of the same frame loop is pinned to the
# reference in tests/test_beam_oracle.py.

def test_multipath_strict_golden(engine, oracle, monkeypatch):
    g = load_beam_golden("beam_multipath.npz")
    assert g["lex"]["lm_type"] == 0x100
    lx = lib.Lexicon(engine, g["lex"])
    bm = lib.Beam(engine, lx, g["beam_width"], g["score_pruning_width"], max_utts=len(g["utts"]))
    bm.set_order_mode("fast")
    with pytest.raises(lib.JamdError):                       # the canonical-tie kernel does not take these
        bm.pass1_host([oracle.gmm_outprob(g["am"], g["utts"][0]["frames"])])
    bm.set_strict_order(True)
    res, tre = bm.pass1_host([oracle.gmm_outprob(g["am"], u["frames"]) for u in g["utts"]])
    for r, atoms, u in zip(res, tre, g["utts"]):
        assert r.status == 0
        assert_trellis_equal(atoms, u["trellis"])
        assert np.array_equal(np.array(r.wseq[:r.wnum]), u["wseq"]) and r.score == u["score"]


@pytest.mark.parametrize("kind,seed,beam,extra", [
    ("ngram", 47, 150, ["-sepnum", "4", "-bs", "60", "-multipath"]),
    ("ngram", 48, 40, ["-sepnum", "0", "-iwcd1", "avg", "-multipath"]),
    ("grammar", 49, 100, ["-penalty1", "-2.0", "-multipath"]),
])
def test_multipath_strict_vs_reference_live(engine, oracle, ref, tmp_path, monkeypatch, kind, seed, beam, extra):
    if kind == "ngram":
        eng, lex, am, task = ref_task(ref, tmp_path, seed, beam, extra)
        utts = [synth.make_utterance(task, nwords=2 + 2 * u, seed=100 * seed + u)[0] for u in range(3)]
    else:
        eng, lex, am, task = ref_grammar_task(ref, tmp_path, seed, beam, extra, nword=70)
        utts = [synth.make_triphone_grammar_utterance(task, nwords=2 + 2 * u, seed=100 * seed + u)[0] for u in range(3)]
    bs = float(extra[extra.index("-bs") + 1]) if "-bs" in extra else -1.0
    lx = lib.Lexicon(engine, lex)
    bm = lib.Beam(engine, lx, eng.beam_width, bs, max_utts=len(utts), atoms_per_utt=1 << 17)
    bm.set_strict_order(True)
    res, tre = bm.pass1_host([oracle.gmm_outprob(am, fr) for fr in utts])
    for fr, r, atoms in zip(utts, res, tre):
        synth.write_htk_param(tmp_path / "u.mfc", fr)
        rtr, (rwseq, rscore) = eng.recognize(tmp_path / "u.mfc")
        assert_trellis_equal(atoms, rtr)
        if r.status == 0:
            assert np.array_equal(np.array(r.wseq[:r.wnum]), rwseq) and r.score == rscore
