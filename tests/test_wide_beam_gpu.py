"""Exact-order first pass at WIDE beams (the reference's DNN recipe runs `-b 4000`, /root/reference README.md:127)
against the compiled reference on the BASELINE-size lexicon: 20 000 words, tree built by libjulius/src/wchmm.c,
`julius -1pass -input outprob` over the same [T][S] score matrix the device search reads.

The score streams are chosen to drive every form of the pruning step with REAL rank pruning on a big lexicon:
  gmm    -- scores of a GMM along a real word sequence (peaked: the search follows a path);
  flat   -- nearly uniform scores (what a random-init DNN emits): three to five tokens per survivor, frames larger
            than the LDS heap at the widest beams (the heap of such a frame is built in global memory);
  ties   -- the flat stream quantised to 0.25: every frame is full of exactly equal scores, so the visiting order
            (first writer wins, libjulius/src/beam.c:1945-1980; heap order of sort_token_upward, :1342-1480) decides
            most of the trellis.
Beams 1500 (narrow layout, closed-form extraction), 2500 / 4000 / 4400 (wide layout: survivors in the utterance's
slice, the pruning step overlays the whole LDS image) and 6000 (wide layout, no room for the top lists: sequential
extraction).  The word trellis must be IDENTICAL, ties included."""
import numpy as np
import pytest

from beamutil import assert_trellis_equal
from julius_amd import lexblob, lib, synth
from oracle import pyoracle

pytestmark = pytest.mark.gpu

S = 3000


@pytest.fixture(scope="module")
def big_task(tmp_path_factory):
    wd = tmp_path_factory.mktemp("wide")
    task = synth.make_triphone_task(wd, nphone=40, S=S, M=1, nword=20000, nvar=25, seed=0, maxlen=8, nbigram_per_word=10)
    return wd, task


def _streams(task, am, oracle, beam):
    rng = np.random.default_rng(beam)
    fr = synth.make_utterance(task, nwords=3, seed=9000 + beam)[0]      # a whole (short) utterance: the pass ends in a sentence
    flat = rng.normal(-8.0, 0.33, (90, S)).astype(np.float32)
    return {"gmm": oracle.gmm_outprob(am, fr), "flat": flat, "ties": (np.round(flat * 4.0) / 4.0).astype(np.float32)}


@pytest.mark.parametrize("beam", [1500, 2500, 4000, 4400, 6000])
def test_wide_beam_exact_vs_compiled_reference(engine, oracle, ref, big_task, tmp_path, beam):
    wd, task = big_task
    eng = pyoracle.RefEngine(ref, ["-h", task["hmmdefs"], "-hlist", task["hmmlist"], "-v", task["dict"], "-nlr", task["arpa"],
                                   "-input", "outprob", "-1pass", "-b", str(beam)])
    eng.save_lexicon(tmp_path / "lex.blob")
    lex = lexblob.load(tmp_path / "lex.blob")
    assert lex["nnode"] > 200000 and eng.beam_width == beam
    am = ref.am_load(task["hmmdefs"], task["hmmlist"]).export()
    streams = _streams(task, am, oracle, beam)
    if beam == 6000:
        streams = {k: v[:40] for k, v in streams.items()}       # sequential extraction: 6000 pops a frame on one lane
    lx = lib.Lexicon(engine, lex)
    bm = lib.Beam(engine, lx, beam, -1.0, max_utts=len(streams), atoms_per_utt=1 << 18)
    assert bm.order_mode() == "exact"
    res, tre = bm.pass1_host(list(streams.values()))
    peak = 0
    for (kind, sc), r, atoms in zip(streams.items(), res, tre):
        synth.write_htk_param(tmp_path / "u.prob", sc, parmkind=synth.PARM_USER)
        rtr, (rwseq, rscore) = eng.recognize(tmp_path / "u.prob")
        assert r.status in (0, 1), (kind, r.status)
        assert len(rtr["wid"]) > 50, kind
        assert_trellis_equal(atoms, rtr)                          # exact, ties included
        if r.status == 0:
            assert np.array_equal(np.array(r.wseq[:r.wnum]), rwseq) and r.score == rscore, kind
        peak = max(peak, r.max_tokens)
    assert peak > beam                                            # rank pruning really happened
    bm.close()


def test_wide_beam_streaming_equals_one_shot(engine, ref, big_task, tmp_path):
    """The wide layout keeps its survivors in the utterance's slice between launches: input pushed in ragged chunks
    (jamd_beam_stream_*) gives the one-shot call's trellis, at a beam where rank pruning is live on the big lexicon."""
    wd, task = big_task
    beam = 2500
    eng = pyoracle.RefEngine(ref, ["-h", task["hmmdefs"], "-hlist", task["hmmlist"], "-v", task["dict"], "-nlr", task["arpa"],
                                   "-input", "outprob", "-1pass", "-b", str(beam)])
    eng.save_lexicon(tmp_path / "lex.blob")
    lex = lexblob.load(tmp_path / "lex.blob")
    rng = np.random.default_rng(3)
    flat = rng.normal(-8.0, 0.33, (70, S)).astype(np.float32)
    scores = [flat, (np.round(flat[:55] * 4.0) / 4.0).astype(np.float32)]
    lx = lib.Lexicon(engine, lex)
    one = lib.Beam(engine, lx, beam, -1.0, max_utts=2, atoms_per_utt=1 << 18)
    res1, tre1 = one.pass1_host(scores)
    bm = lib.Beam(engine, lx, beam, -1.0, max_utts=2, atoms_per_utt=1 << 18)
    bm.stream_begin(2)
    pos = [0, 0]
    chunks = [1, 16, 0, 9, 23, 10000]
    for ci, c in enumerate(chunks):
        part, off = [], [0]
        for u, sc in enumerate(scores):
            n = min(len(sc) - pos[u], c + 2 * u if c else 0)
            part.append(sc[pos[u]:pos[u] + n]); pos[u] += n; off.append(off[-1] + n)
        rows = np.concatenate(part) if off[-1] else np.zeros((1, S), np.float32)
        d = lib.DevBuf(engine, rows.nbytes).upload(rows)
        bm.stream_push_dev(d.ptr, S, np.array(off, np.int32), final=ci == len(chunks) - 1)
        bm.results(2)
        d.free()
    for u, r in enumerate(bm.results(2)):
        assert (r.status, r.natom, r.score, r.max_tokens) == (res1[u].status, res1[u].natom, res1[u].score, res1[u].max_tokens)
        assert r.max_tokens > beam
        a, b = lexblob.canonical_trellis(bm.trellis(u)), lexblob.canonical_trellis(tre1[u])
        assert all(np.array_equal(a[k], b[k]) for k in a)
    # and the one-shot result is the reference's
    synth.write_htk_param(tmp_path / "u.prob", scores[0], parmkind=synth.PARM_USER)
    rtr, _ = eng.recognize(tmp_path / "u.prob")
    assert_trellis_equal(tre1[0], rtr)
    one.close(); bm.close()


@pytest.mark.parametrize("mode", ["exact", "exact_serial"])
@pytest.mark.parametrize("beam", [2000, 3000, 4000, 4400, 5000])
def test_prune_order_wide(engine, oracle, beam, mode):
    """The pruning step alone at wide beams (wide LDS layout; beam 5000: no room for the top lists) on score vectors
    full of exact ties, against the sequential restatement of sort_token_no_order() (beam.c:1492)."""
    from beamutil import load_beam_golden
    g = load_beam_golden("beam_rank.npz")
    lx = lib.Lexicon(engine, g["lex"])
    bm = lib.Beam(engine, lx, beam, -1.0, max_utts=1).set_order_mode(mode)
    rng = np.random.default_rng(beam)
    sizes = sorted(set([beam, beam + 1, 2 * beam, 2 * beam + 1, 2 * beam + 2, 3 * beam + 5, 14000, 15800, 16500, 21000, 50000] +
                       [int(x) for x in rng.integers(beam, 6 * beam, 6)]))
    for n in sizes:
        for levels in (0, 3, 60, 3000):
            if levels == 0:
                sc = rng.permutation(n).astype(np.float32) * -0.37 - 100.0
            else:
                sc = (-rng.integers(0, levels, n).astype(np.float32) * 0.5 - 2000.0).astype(np.float32)
            got = bm.prune_order(sc)
            want = oracle.sort_token_no_order(sc, beam)
            assert np.array_equal(got, want), (n, beam, levels, mode)
    bm.close()


def test_c4_dnn_hmm_end_to_end_at_beam_4000(engine, ref, tmp_path):
    """BASELINE configs[3] at full size and at the reference recipe's beam (`-b 4000`, README.md:127): hmmdefs with
    4000 states + dnnconf (48 x 11 -> 6 x 2048 -> 4000, .npy weights) + 20 000-word dictionary + ARPA 2-gram, all
    loaded by the reference's own readers; the device runs MFMA DNN scores -> exact-order first pass on the lexicon
    wchmm.c built, and must give the word trellis of the compiled reference's `julius -1pass -dnnconf` (its own
    dnn_calc_outprob() + beam.c) entry by entry."""
    if b"FMA" not in ref.lib.jref_simd_string():
        pytest.skip("reference built without its FMA kernel")
    dnn = synth.make_dnn(seed=0)
    task = synth.make_triphone_task(tmp_path, nphone=40, S=4000, M=1, nword=20000, nvar=25, seed=0, maxlen=8, nbigram_per_word=10)
    conf = synth.write_dnnconf(tmp_path, dnn, context_len=11)
    eng = pyoracle.RefEngine(ref, ["-h", task["hmmdefs"], "-hlist", task["hmmlist"], "-v", task["dict"], "-nlr", task["arpa"],
                                   "-dnnconf", conf, "-notypecheck", "-input", "htkparam", "-1pass", "-b", "4000"])
    eng.save_lexicon(tmp_path / "lex.blob")
    lex = lexblob.load(tmp_path / "lex.blob")
    assert lex["nnode"] > 200000 and eng.beam_width == 4000 and eng.nstate == 4000
    net = lib.Dnn.from_dnnconf(engine, conf)
    lx = lib.Lexicon(engine, lex)
    rng = np.random.default_rng(77)
    utts = [rng.normal(0, 1, (T, 528)).astype(np.float32) for T in (130, 90)]
    bm = lib.Beam(engine, lx, 4000, -1.0, max_utts=len(utts), atoms_per_utt=1 << 18)
    assert bm.order_mode() == "exact"
    frames = np.concatenate(utts)
    off = np.array([0, 130, 220], np.int32)
    d_fr = lib.DevBuf(engine, frames.nbytes).upload(frames)
    d_sc = lib.DevBuf(engine, 4 * len(frames) * net.S)
    net.outprob_dev(d_fr.ptr, len(frames), d_sc.ptr)
    bm.pass1_dev(d_sc.ptr, net.S, off)
    res = bm.results()
    for u, fr in enumerate(utts):
        synth.write_htk_param(tmp_path / "u.mfc", fr, parmkind=synth.PARM_USER)
        rtr, (rwseq, rscore) = eng.recognize(tmp_path / "u.mfc")
        assert len(rtr["wid"]) > 5000 and res[u].max_tokens > 8000
        assert_trellis_equal(bm.trellis(u), rtr)                  # exact, ties included
        if len(rwseq):
            assert res[u].status == 0 and np.array_equal(np.array(res[u].wseq[:res[u].wnum]), rwseq) and res[u].score == rscore
        else:
            assert res[u].status == 1
    bm.close()
