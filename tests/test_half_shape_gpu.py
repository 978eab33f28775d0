"""GPU: the exact-order first pass in its HALF workgroup shape (512 threads and half a CU's LDS, two utterances per
CU -- `jamd_beam_set_workgroup_shape`, beam_exact.hip).  The shape is a scheduling choice: every result must be the
full shape's, which the other suites pin to the reference.  Here the half shape is forced and checked against the
same oracles -- the sequential restatement of sort_token_no_order() (libjulius/src/beam.c:1342-1516) for the pruning
step, the compiled reference's golden trellises, `julius -1pass` on a reference-built 20 000-word lexicon, streaming
sessions -- and the automatic choice is checked on a batch large enough to trigger it."""
import numpy as np
import pytest

from beamutil import assert_trellis_equal, load_beam_golden, ref_task
from julius_amd import lexblob, lib, synth

pytestmark = pytest.mark.gpu


def _half(bm):
    bm.set_workgroup_shape("half")
    assert bm.workgroup_shape(1) == "half"
    return bm


def _scores(rng, n, levels):
    if levels == 0:
        return rng.permutation(n).astype(np.float32) * -0.37 - 100.0
    return (-rng.integers(0, levels, n).astype(np.float32) * 0.5 - 2000.0).astype(np.float32)


@pytest.mark.parametrize("mode", ["exact", "exact_serial"])
@pytest.mark.parametrize("beam", [1, 7, 33, 200, 800, 1000])
def test_prune_order_fuzz_half(engine, oracle, beam, mode):
    g = load_beam_golden("beam_rank.npz")
    lx = lib.Lexicon(engine, g["lex"])
    bm = _half(lib.Beam(engine, lx, beam, -1.0, max_utts=1).set_order_mode(mode))
    rng = np.random.default_rng(1000 + beam)
    sizes = sorted(set([1, 2, 3, beam, beam + 1, 2 * beam, 2 * beam + 1, 2 * beam + 2, 3 * beam + 5, 5 * beam + 17] +
                       [int(x) for x in rng.integers(1, max(8 * beam, 64), 20)]))
    for n in sizes:
        for levels in (0, 2, 5, 40, 1000):
            sc = _scores(rng, n, levels)
            assert np.array_equal(bm.prune_order(sc), oracle.sort_token_no_order(sc, beam)), (n, beam, levels, mode)


def test_prune_order_large_frames_half(engine, oracle):
    """Frames around and beyond what half a CU's LDS holds (the heap moves to global memory, the closed form stays)."""
    g = load_beam_golden("beam_rank.npz")
    lx = lib.Lexicon(engine, g["lex"])
    bm = _half(lib.Beam(engine, lx, 800, -1.0, max_utts=1))
    rng = np.random.default_rng(6)
    for n in (1700, 2900, 4200, 6000, 7700, 7800, 9000, 16000, 40000):
        for dup in (0.0, 0.08, 0.5):
            base = (-rng.random(n) * 300.0 - 5000.0).astype(np.float32)
            ndup = int(dup * n)
            if ndup:
                base[rng.integers(0, n, ndup)] = base[rng.integers(0, n, ndup)]
            assert np.array_equal(bm.prune_order(base), oracle.sort_token_no_order(base, 800)), (n, dup)


def test_half_shape_unavailable_is_reported(engine):
    """A beam whose typical frame does not fit half a CU's LDS: asking for the half shape is an error, auto stays full."""
    g = load_beam_golden("beam_rank.npz")
    lx = lib.Lexicon(engine, g["lex"])
    bm = lib.Beam(engine, lx, 4000, -1.0, max_utts=1)
    with pytest.raises(lib.JamdError):
        bm.set_workgroup_shape("half")
    assert bm.workgroup_shape(100000) == "full"


@pytest.mark.parametrize("name", ["beam_rank.npz", "beam_score.npz", "beam_isolated.npz", "beam_grammar.npz",
                                  "beam_grammar_free.npz"])
@pytest.mark.parametrize("mode", ["exact", "exact_serial"])
def test_golden_half(engine, oracle, name, mode):
    g = load_beam_golden(name)
    lx = lib.Lexicon(engine, g["lex"])
    bm = _half(lib.Beam(engine, lx, g["beam_width"], g["score_pruning_width"], max_utts=len(g["utts"])).set_order_mode(mode))
    res, tre = bm.pass1_host([oracle.gmm_outprob(g["am"], u["frames"]) for u in g["utts"]])
    for r, atoms, u in zip(res, tre, g["utts"]):
        assert r.status == 0
        assert_trellis_equal(atoms, u["trellis"])
        assert np.array_equal(np.array(r.wseq[:r.wnum]), u["wseq"]) and r.score == u["score"]


@pytest.mark.parametrize("chunks", [[1] * 40 + [10000], [7, 1, 50, 0, 3, 10000]])
def test_streaming_half(engine, oracle, chunks):
    g = load_beam_golden("beam_score.npz")
    scores = [oracle.gmm_outprob(g["am"], u["frames"]) for u in g["utts"]]
    S = scores[0].shape[1]
    lx = lib.Lexicon(engine, g["lex"])
    bm = _half(lib.Beam(engine, lx, g["beam_width"], g["score_pruning_width"], max_utts=len(scores)))
    bm.stream_begin(len(scores))
    pos = [0] * len(scores)
    for ci, c in enumerate(chunks):
        part, off = [], [0]
        for u, sc in enumerate(scores):
            n = min(len(sc) - pos[u], (c + u) if c else 0)
            part.append(sc[pos[u]:pos[u] + n]); pos[u] += n; off.append(off[-1] + n)
        rows = np.concatenate(part) if off[-1] else np.zeros((1, S), np.float32)
        d = lib.DevBuf(engine, rows.nbytes).upload(rows)
        bm.stream_push_dev(d.ptr, S, np.array(off, np.int32), final=ci == len(chunks) - 1)
        d.free()
    res = bm.results(len(scores))
    for u, r in enumerate(res):
        assert r.status == 0 and r.score == g["utts"][u]["score"]
        assert_trellis_equal(bm.trellis(u), g["utts"][u]["trellis"])


def test_full_size_half_vs_compiled_reference_and_auto_choice(engine, ref, tmp_path):
    """BASELINE configs[2] at full size (reference-built 20 000-word lexicon, beam 800): the half shape against
    `julius -1pass` entry by entry; then a batch of 1.5 x CU count + 1 utterances, where the automatic choice is the
    half shape, against the full shape on the same work area."""
    eng, lex, am, task = ref_task(ref, tmp_path, 0, 800, [], nphone=40, S=3000, M=16, nword=20000, nvar=25, maxlen=8,
                                  nbigram_per_word=10)
    utts = [synth.make_utterance(task, nwords=4 + 5 * u, seed=7100 + u)[0] for u in range(3)]
    gm = lib.Gmm(engine, am)
    scores = [gm.outprob_host(fr) for fr in utts]
    lx = lib.Lexicon(engine, lex)
    probe = lib.Beam(engine, lx, 800, -1.0, max_utts=1).set_workgroup_shape("auto")
    nbig = next(n for n in range(1, 1 << 14) if probe.workgroup_shape(n) == "half")   # 1.5 x CU count + 1
    probe.close()
    assert nbig > 64
    bm = lib.Beam(engine, lx, 800, -1.0, max_utts=nbig, atoms_per_utt=1 << 16).set_workgroup_shape("auto")
    assert bm.order_mode() == "exact"
    assert bm.workgroup_shape(nbig - 1) == "full" and bm.workgroup_shape(nbig) == "half"
    _half(bm)
    res, tre = bm.pass1_host(scores)
    for fr, r, atoms in zip(utts, res, tre):
        synth.write_htk_param(tmp_path / "u.mfc", fr)
        rtr, (rwseq, rscore) = eng.recognize(tmp_path / "u.mfc")
        assert r.status == 0 and len(rtr["wid"]) > 1000
        assert_trellis_equal(atoms, rtr)
        assert np.array_equal(np.array(r.wseq[:r.wnum]), rwseq) and r.score == rscore
    # the automatic choice on a large batch of short utterances (the first frames of the three, cut at different lengths)
    short = [scores[u % 3][:40 + (u * 7) % 50] for u in range(nbig)]
    bm.set_workgroup_shape("auto")
    res_a, tre_a = bm.pass1_host(short)
    bm.set_workgroup_shape("full")
    res_f, tre_f = bm.pass1_host(short)
    for u in range(nbig):
        a, f = res_a[u], res_f[u]
        assert (a.status, a.natom, a.wnum, a.score, a.frames) == (f.status, f.natom, f.wnum, f.score, f.frames)
        if u % 16 == 0:
            ca, cf = lexblob.canonical_trellis(tre_a[u]), lexblob.canonical_trellis(tre_f[u])
            assert all(np.array_equal(ca[k], cf[k]) for k in ca)


def test_streaming_session_chooses_its_shape_once(engine, oracle):
    """A streaming session of more utterances than 1.5 x the CU count opens in the half shape and keeps it for every
    push (the parked state is the layout's); the results equal the one-shot call in the full shape."""
    g = load_beam_golden("beam_score.npz")
    base = [oracle.gmm_outprob(g["am"], u["frames"]) for u in g["utts"]]
    S = base[0].shape[1]
    lx = lib.Lexicon(engine, g["lex"])
    probe = lib.Beam(engine, lx, g["beam_width"], g["score_pruning_width"], max_utts=1).set_workgroup_shape("auto")
    nbig = next(n for n in range(1, 1 << 14) if probe.workgroup_shape(n) == "half")
    probe.close()
    scores = [base[u % len(base)][:len(base[u % len(base)]) - (u * 3) % 17] for u in range(nbig)]
    bm = lib.Beam(engine, lx, g["beam_width"], g["score_pruning_width"], max_utts=nbig).set_workgroup_shape("auto")
    assert bm.workgroup_shape(nbig) == "half"
    bm.stream_begin(nbig)
    pos = [0] * nbig
    while True:
        take = [min(23 + u % 5, len(sc) - p) for u, (sc, p) in enumerate(zip(scores, pos))]
        final = all(p + k >= len(sc) for sc, p, k in zip(scores, pos, take))
        rows = np.concatenate([sc[p:p + k] for sc, p, k in zip(scores, pos, take)])
        off = np.zeros(nbig + 1, np.int32)
        off[1:] = np.cumsum(take)
        pos = [p + k for p, k in zip(pos, take)]
        d = lib.DevBuf(engine, rows.nbytes).upload(rows)
        bm.stream_push_dev(d.ptr, S, off, final=final)
        d.free()
        if final:
            break
    res_s = bm.results(nbig)
    tre_s = [bm.trellis(u) for u in range(0, nbig, 37)]
    bm.set_workgroup_shape("full")
    res_f, tre_f = bm.pass1_host(scores)
    for u in range(nbig):
        a, f = res_s[u], res_f[u]
        assert (a.status, a.natom, a.wnum, a.score, a.frames) == (f.status, f.natom, f.wnum, f.score, f.frames)
    for i, u in enumerate(range(0, nbig, 37)):
        assert_trellis_equal(tre_s[i], lexblob.canonical_trellis(tre_f[u]))


def test_wait_started(engine, oracle):
    """jamd_beam_wait_started(): returns at once before any launch, and after a launch once the kernel is next to run;
    the results are untouched by it."""
    g = load_beam_golden("beam_rank.npz")
    lx = lib.Lexicon(engine, g["lex"])
    bm = lib.Beam(engine, lx, g["beam_width"], g["score_pruning_width"], max_utts=len(g["utts"]))
    bm.wait_started()
    scores = [oracle.gmm_outprob(g["am"], u["frames"]) for u in g["utts"]]
    S = scores[0].shape[1]
    off = np.zeros(len(scores) + 1, np.int32)
    off[1:] = np.cumsum([len(x) for x in scores])
    allsc = np.ascontiguousarray(np.concatenate(scores), np.float32)
    d = lib.DevBuf(engine, allsc.nbytes).upload(allsc)
    bm.pass1_dev(d.ptr, S, off)
    bm.wait_started()
    res = bm.results()
    for u, r in enumerate(res):
        assert r.status == 0 and r.score == g["utts"][u]["score"]
        assert_trellis_equal(bm.trellis(u), g["utts"][u]["trellis"])
    d.free()


# ---------------------------------------------------------------- the multipath frame (csrc/beam_exact_mp.h) in the half shape
def test_multipath_golden_half(engine, oracle):
    """The compiled reference's golden multipath trellises, decoded two utterances per CU (round 5)."""
    g = load_beam_golden("beam_multipath.npz")
    lx = lib.Lexicon(engine, g["lex"])
    scores = [oracle.gmm_outprob(g["am"], u["frames"]) for u in g["utts"]]
    for mode in ("exact", "exact_serial"):
        bm = _half(lib.Beam(engine, lx, g["beam_width"], g["score_pruning_width"], max_utts=len(scores)).set_order_mode(mode))
        res, tre = bm.pass1_host(scores)
        for r, atoms, u in zip(res, tre, g["utts"]):
            assert r.status == 0
            assert_trellis_equal(atoms, u["trellis"])
            assert np.array_equal(np.array(r.wseq[:r.wnum]), u["wseq"]) and r.score == u["score"]
        bm.close()


@pytest.mark.parametrize("chunks", [[7] * 60, [0, 25, 0, 3, 1000]])
def test_multipath_streaming_half(engine, oracle, chunks):
    g = load_beam_golden("beam_multipath.npz")
    lx = lib.Lexicon(engine, g["lex"])
    scores = [oracle.gmm_outprob(g["am"], u["frames"]) for u in g["utts"]]
    S = scores[0].shape[1]
    bm = _half(lib.Beam(engine, lx, g["beam_width"], g["score_pruning_width"], max_utts=len(scores)))
    bm.stream_begin(len(scores))
    pos = [0] * len(scores)
    for ci, c in enumerate(chunks):
        part, off = [], [0]
        for u, sc in enumerate(scores):
            n = min(len(sc) - pos[u], (c + u) if c else 0)
            part.append(sc[pos[u]:pos[u] + n]); pos[u] += n; off.append(off[-1] + n)
        rows = np.concatenate(part) if off[-1] else np.zeros((1, S), np.float32)
        d = lib.DevBuf(engine, rows.nbytes).upload(rows)
        bm.stream_push_dev(d.ptr, S, np.array(off, np.int32), final=ci == len(chunks) - 1)
        bm.results(len(scores))
        d.free()
    for u, r in enumerate(bm.results(len(scores))):
        gu = g["utts"][u]
        assert r.status == 0 and r.score == gu["score"]
        assert_trellis_equal(bm.trellis(u), gu["trellis"])


@pytest.mark.parametrize("seed", range(10))
def test_multipath_tie_heavy_fuzz_half(engine, oracle, ref, tmp_path, seed):
    """The tie-heavy fuzz of tests/test_multipath_exact_gpu.py (reference-built multipath lexicons: plain, state skips, two
    entry arcs, -iwsp, grammar; quantised random scores) in the half shape: the mid-frame sort's whole array comes out of
    exact_prune<FULL> on 512 threads and half a CU's LDS, or of the extraction loop where that does not fit."""
    from beamutil import ref_grammar_task
    from test_multipath_exact_gpu import SKIP_TRANS, SPLIT_TRANS
    rng = np.random.default_rng(5200 + seed)
    kind = ["plain", "skip", "split", "iwsp", "grammar"][seed % 5]
    nword = int(rng.choice([80, 200, 400]))
    sep = str(int(rng.choice([0, 3, 20])))
    if kind == "grammar":
        eng, lex, am, task = ref_grammar_task(ref, tmp_path, 500 + seed, 50, ["-penalty1", "-1.5", "-multipath"], wrap=bool(seed & 1), nword=max(nword, 70))
    elif kind == "iwsp":
        eng, lex, am, task = ref_task(ref, tmp_path, 500 + seed, 50, ["-sepnum", sep, "-multipath", "-iwsp", "-spmodel", "sp"], sp=True, nword=nword)
    elif kind in ("skip", "split"):
        eng, lex, am, task = ref_task(ref, tmp_path, 500 + seed, 50, ["-sepnum", sep], nword=nword, trans=SKIP_TRANS if kind == "skip" else SPLIT_TRANS)
    else:
        eng, lex, am, task = ref_task(ref, tmp_path, 500 + seed, 50, ["-sepnum", sep, "-multipath"], nword=nword)
    assert lex["lm_type"] & 0x100
    S = len(am["st_off"]) - 1
    lx = lib.Lexicon(engine, lex)
    T = int(rng.integers(25, 70))
    step = float(rng.choice([0.5, 2.0, 8.0]))
    scores = [(-np.round(rng.random((T, S)) * 40.0 / step) * step - 20.0).astype(np.float32) for _ in range(3)]
    sorted_frames = 0
    for beam in (2, int(rng.integers(5, 40)), int(rng.integers(40, 300))):
        for width in (-1.0, float(rng.choice([30.0, 80.0]))):
            bm = _half(lib.Beam(engine, lx, beam, width, max_utts=len(scores), atoms_per_utt=1 << 16))
            res, tre = bm.pass1_host(scores)
            for sc, r, atoms in zip(scores, res, tre):
                oatoms, owseq, oscore, rc, died = oracle.beam_pass1(lex, sc, beam, width)
                assert r.status == rc, (seed, kind, beam, width)
                assert_trellis_equal(atoms, lexblob.canonical_trellis(oatoms))
                if rc == 0:
                    assert list(r.wseq[:r.wnum]) == list(owseq) and r.score == oscore
                sorted_frames += int(r.max_tokens > beam)
            bm.close()
    assert sorted_frames > 0


def test_multipath_larger_task_half_and_auto_choice(engine, oracle, ref, tmp_path):
    """A 400-word multipath lexicon whose every frame really sorts (tokens > beam), half shape against the CPU restatement;
    then a batch large enough for the automatic choice to pick the half shape, against the full shape on the same work area."""
    beam = 300
    eng, lex, am, task = ref_task(ref, tmp_path, 71, beam, ["-sepnum", "10", "-multipath"], nword=400)
    utts = [synth.make_utterance(task, nwords=4 + u, seed=7100 + u)[0] for u in range(3)]
    scores = [oracle.gmm_outprob(am, fr) for fr in utts]
    lx = lib.Lexicon(engine, lex)
    probe = lib.Beam(engine, lx, beam, -1.0, max_utts=1).set_workgroup_shape("auto")
    nbig = next(n for n in range(1, 1 << 14) if probe.workgroup_shape(n) == "half")
    probe.close()
    bm = lib.Beam(engine, lx, beam, -1.0, max_utts=nbig, atoms_per_utt=1 << 16)
    _half(bm)
    res, tre = bm.pass1_host(scores)
    for sc, r, atoms in zip(scores, res, tre):
        oatoms, wseq, score, rc, died = oracle.beam_pass1(lex, sc, beam, -1.0)
        assert_trellis_equal(atoms, lexblob.canonical_trellis(oatoms))
        assert (r.status == 0) == (rc == 0) and r.max_tokens > beam
        if rc == 0:
            assert list(r.wseq[:r.wnum]) == list(wseq) and r.score == score
    short = [scores[u % 3][:30 + (u * 7) % 40] for u in range(nbig)]
    bm.set_workgroup_shape("auto")
    assert bm.workgroup_shape(nbig) == "half"
    res_a, tre_a = bm.pass1_host(short)
    bm.set_workgroup_shape("full")
    res_f, tre_f = bm.pass1_host(short)
    for u in range(nbig):
        a, f = res_a[u], res_f[u]
        assert (a.status, a.natom, a.wnum, a.score, a.frames) == (f.status, f.natom, f.wnum, f.score, f.frames)
        if u % 16 == 0:
            assert tre_a[u].tobytes() == tre_f[u].tobytes()
    bm.close()
