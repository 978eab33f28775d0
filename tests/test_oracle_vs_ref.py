"""CPU: oracle restatement vs the compiled reference on fresh seeded inputs
(more shapes than the committed fixtures).  Needs oracle/_ref/libjref.so, which
the dev container builds from /root/reference (oracle/Makefile)."""
import numpy as np
import pytest

from julius_amd import synth
from oracle import pyoracle as po


@pytest.mark.parametrize("S,M,D,ragged,nullf", [(30, 16, 39, False, 0.0), (27, 5, 26, True, 0.0),
                                                (12, 9, 13, True, 0.15), (9, 1, 39, False, 0.0)])
@pytest.mark.parametrize("gprune,n", [("none", 0), ("safe", 2), ("safe", 64)])
def test_plain_gmm(ref, oracle, tmp_path, S, M, D, ragged, nullf, gprune, n):
    m = synth.make_gmm(S=S, M=M, D=D, seed=S * 7 + M, ragged=ragged, null_frac=nullf)
    kind = "MFCC_E_D_A" if D == 39 else "USER"
    synth.write_hmmdefs(tmp_path / "h", m, kind=kind)
    am = ref.am_load(tmp_path / "h", gprune=gprune, gprune_num=n)
    ex = am.export()
    # the product-side flattening reproduces the loader's arrays
    for k in ("mean", "ivar", "gconst"):
        ok = ex["ent_dens"] >= 0
        assert np.array_equal(ex[k][ex["ent_dens"][ok]], m[k][m["ent_dens"][m["ent_dens"] >= 0]])
    assert np.array_equal(ex["st_off"], m["st_off"])
    assert np.array_equal(ex["ent_logw"], m["ent_logw"])
    fr = synth.make_frames(m, T=37, seed=3)
    want = am.outprob(fr)
    got = oracle.gmm_outprob(ex, fr, po.GPRUNE_NONE if gprune == "none" else po.GPRUNE_SAFE, n)
    assert np.array_equal(got, want)
    am.close()


@pytest.mark.parametrize("gprune,n", [("none", 64), ("safe", 1), ("safe", 2), ("safe", 8)])
def test_tied_gmm(ref, oracle, tmp_path, gprune, n):
    m = synth.make_tied_gmm(S=21, nbook=4, K=64, D=39, seed=5)
    synth.write_hmmdefs(tmp_path / "h", m)
    am = ref.am_load(tmp_path / "h", gprune=gprune, gprune_num=n)
    assert am.is_tied and am.nbook == 4
    ex = am.export()
    fr = synth.make_frames(m, T=45, seed=9, noise=2.0)
    got = oracle.gmm_outprob(ex, fr, po.GPRUNE_NONE if gprune == "none" else po.GPRUNE_SAFE, n)
    assert np.array_equal(got, am.outprob(fr))
    am.close()


@pytest.mark.parametrize("gprune", ["heu", "beam"])
@pytest.mark.parametrize("n,nbook,K,noise", [(1, 3, 64, 2.0), (2, 4, 64, 2.0), (4, 2, 129, 1.0), (10, 1, 40, 3.0), (64, 2, 24, 2.0)])
def test_tied_gmm_history_pruning(ref, oracle, tmp_path, gprune, n, nbook, K, noise):
    """SURVEY 8a A7, the tied-mixture half: gprune_heu() / gprune_beam() with a live last_id (`gprune_heu.c:305-335`,
    `gprune_beam.c:301-336`): the thresholds of frame t come from the codebook's cached winners of frame t-1
    (`calc_tied_mix.c:203-215`).  Under eager scoring (the batch loop `outprob.c:230-242`: every state of every frame)
    that history is deterministic, and the restatement must give the compiled reference's numbers and cache."""
    m = synth.make_tied_gmm(S=21, nbook=nbook, K=K, D=39, seed=K + n)
    synth.write_hmmdefs(tmp_path / "h", m)
    am = ref.am_load(tmp_path / "h", gprune=gprune, gprune_num=n)
    assert am.is_tied and am.nbook == nbook
    ex = am.export()
    fr = synth.make_frames(m, T=60, seed=9 + n, noise=noise)
    code = po.GPRUNE_HEU if gprune == "heu" else po.GPRUNE_BEAM
    want = am.outprob(fr)
    assert np.array_equal(oracle.gmm_outprob(ex, fr, code, n), want)
    safe = oracle.gmm_outprob(ex, fr, po.GPRUNE_SAFE, n)
    if n < 10:
        assert not np.array_equal(safe, want)              # these methods really prune differently from safe
    cap = n                                                 # the reference's cache rows are OP_gprune_num wide
    for b in range(nbook):
        sc, ids, num = am.tmix_cache(fr, b, cap)
        osc, oids, onum = oracle.tmix_topn(ex, b, fr, code, n)
        assert np.array_equal(num, onum)
        for t in range(len(fr)):
            assert np.array_equal(ids[t, :num[t]], oids[t, :num[t]]) and np.array_equal(sc[t, :num[t]], osc[t, :num[t]])
    am.close()


def test_lazy_equals_eager(ref, tmp_path):
    """outprob.c:245-247 (lazy cache fill) and :230-242 (batch) give the same values."""
    m = synth.make_gmm(S=15, M=4, D=39, seed=2)
    synth.write_hmmdefs(tmp_path / "h", m)
    am = ref.am_load(tmp_path / "h")
    fr = synth.make_frames(m, T=20, seed=4)
    full = am.outprob(fr)
    rng = np.random.default_rng(0)
    tt = np.sort(rng.integers(0, 20, 100)).astype(np.int32)
    ss = rng.integers(0, 15, 100).astype(np.int32)
    assert np.array_equal(am.outprob_list(fr, tt, ss), full[tt, ss])
    am.close()


@pytest.mark.parametrize("dims", [(48, 64, 64, 40), (40, 128, 128, 128, 128, 61), (528, 256, 256, 100)])
def test_dnn(ref, oracle, tmp_path, dims):
    if "FMA" not in ref.lib.jref_simd_string().decode():
        pytest.skip("host without FMA: the reference picks another SIMD kernel")
    dnn = synth.make_dnn(dims=dims, seed=dims[0])
    r = ref.dnn_load(dnn, tmp_path)
    fr = np.random.default_rng(1).normal(0, 1.5, (15, dims[0])).astype(np.float32)
    assert np.array_equal(oracle.dnn_outprob(dnn, fr, po.DNN_FMA), r.outprob(fr))


def test_gmm_blob_roundtrip(ref, tmp_path):
    """jamd_gmm_save() (reference-side shim) -> lexblob.load_gmm(): the same arrays the
    in-memory export gives, for a plain and a tied-mixture model."""
    from julius_amd import lexblob
    for name, m in (("plain", synth.make_gmm(S=12, M=3, D=39, seed=5, ragged=True)),
                    ("tied", synth.make_tied_gmm(S=12, nbook=2, K=8, D=39, seed=6))):
        synth.write_hmmdefs(tmp_path / name, m)
        am = ref.am_load(tmp_path / name)
        am.save_blob(tmp_path / f"{name}.blob")
        a, b = am.export(), lexblob.load_gmm(tmp_path / f"{name}.blob")
        for k in ("mean", "ivar", "gconst", "st_off", "ent_dens", "ent_logw"):
            assert np.array_equal(a[k], b[k]), (name, k)
        assert (a["st_book"] is None) == (b["st_book"] is None) and a["nbook"] == b["nbook"]
        if a["st_book"] is not None:
            assert np.array_equal(a["st_book"], b["st_book"])


@pytest.mark.parametrize("nbest", [4, 8, 24])
def test_gaussian_mixture_selection(oracle, ref, tmp_path, nbest):
    """-gshmm / -gsnum: gms_state() (gms.c:394) returns the real score for states whose selection
    state is among the nbest of the frame and the selection state's score otherwise."""
    task = synth.make_triphone_task(tmp_path, seed=5, nword=60)
    gpath, _ = synth.make_gs_model(task, seed=5)
    am = ref.am_load(task["hmmdefs"], task["hmmlist"], gshmm=gpath, gms_num=nbest)
    gs = am.gms()
    assert gs["nbest"] == nbest and len(gs["model"]["st_off"]) - 1 == 78
    full_model = ref.am_load(task["hmmdefs"], task["hmmlist"]).export()
    for u in range(3):
        fr, _ = synth.make_utterance(task, nwords=2 + u, seed=50 + u)
        want = am.outprob(fr)
        got = oracle.gms_apply(gs, fr, oracle.gmm_outprob(full_model, fr))
        used = gs["state2gs"] >= 0              # states outside every model: the reference reads out of bounds
        assert np.array_equal(got[:, used], want[:, used])
        assert 0.0 < (got[:, used] != oracle.gmm_outprob(full_model, fr)[:, used]).mean() < 1.0


@pytest.mark.parametrize("num,null_frac", [(1, 0.0), (3, 0.2), (10, 0.0), (64, 0.1)])
def test_verification_gmm(oracle, ref, tmp_path, num, null_frac):
    """-gmm / -gmmnum: gmm_proceed() (libjulius/src/gmm.c:574-600) through the reference's own entry
    points, frame by frame, and gc->gmm_score[] / the winner after a whole input."""
    task = synth.make_triphone_task(tmp_path, seed=12, nword=60)
    gpath, _, names = synth.make_rejection_gmm(tmp_path, task["model"]["centre"], seed=12 + num, M=20, null_frac=null_frac)
    eng = po.RefEngine(ref, ["-h", task["hmmdefs"], "-hlist", task["hmmlist"], "-v", task["dict"], "-nlr", task["arpa"],
                             "-input", "htkparam", "-gprune", "none", "-b", "120", "-gmm", str(gpath), "-gmmnum", str(num)])
    info = eng.gmm_info()
    assert info["nmodel"] == len(names) and info["gprune_num"] == num
    for u in range(2):
        fr, _ = synth.make_utterance(task, nwords=2 + u, seed=70 + u)
        want = eng.gmm_frame_scores(fr)
        got = oracle.rejgmm_frame_scores(info, fr)
        assert np.array_equal(got, want)
        synth.write_htk_param(tmp_path / "u.mfc", fr)
        eng.recognize(tmp_path / "u.mfc")
        sums, winner, cm, valid, nframe = eng.gmm_result()
        assert nframe == len(fr) and valid
        assert np.array_equal(oracle.rejgmm_accumulate(got), sums) and int(np.argmax(sums)) == winner
