"""CPU: the oracle restatement (oracle/jamd_oracle_am.c) against the committed
golden fixtures, which were produced by the COMPILED REFERENCE
(tests/make_golden.py -> oracle/_ref/libjref.so).  Bit-exact."""
import numpy as np
import pytest

from conftest import GOLDEN
from oracle import pyoracle as po


def load(name):
    z = np.load(GOLDEN / name)
    m = {k: z[k] for k in z.files}
    m["nbook"] = int(m.get("nbook", 0))
    if "book_size" in m:
        m["book_size"] = int(m["book_size"])
    m.setdefault("st_book", None)
    return m


def test_tables(oracle):
    g = load("tables.npz")
    assert np.array_equal(oracle.log_tbl()[g["addlog_idx"]], g["addlog_val"])
    assert np.array_equal(oracle.logistic_tbl()[g["logistic_idx"]], g["logistic_val"])


def test_gmm_plain_none(oracle):
    g = load("gmm_plain_none.npz")
    assert np.array_equal(oracle.gmm_outprob(g, g["frames"], po.GPRUNE_NONE), g["out"])


def test_gmm_ragged_null_densities(oracle):
    g = load("gmm_ragged.npz")
    assert (g["ent_dens"] < 0).any(), "fixture must contain NULL densities"
    assert np.array_equal(oracle.gmm_outprob(g, g["frames"], po.GPRUNE_NONE), g["out"])
    assert np.array_equal(oracle.gmm_outprob(g, g["frames"], po.GPRUNE_SAFE, 3), g["out_safe3"])


@pytest.mark.parametrize("key,gp,n", [("out_none", po.GPRUNE_NONE, 32), ("out_safe2", po.GPRUNE_SAFE, 2),
                                      ("out_safe4", po.GPRUNE_SAFE, 4)])
def test_gmm_tied(oracle, key, gp, n):
    g = load("gmm_tied.npz")
    assert np.array_equal(oracle.gmm_outprob(g, g["frames"], gp, n), g[key])


def test_tied_codebook_cache(oracle):
    g = load("gmm_tied.npz")
    sc, ids, num = oracle.tmix_topn(g, 1, g["frames"], po.GPRUNE_SAFE, 2)
    assert np.array_equal(num, g["cache2_num"])
    assert np.array_equal(ids, g["cache2_id"])
    assert np.array_equal(sc, g["cache2_score"])


@pytest.mark.parametrize("meth,code", [("max", po.IWCD_MAX), ("avg", po.IWCD_AVG), ("nbest", po.IWCD_NBEST)])
def test_outprob_cd(oracle, meth, code):
    g = load("cdset.npz")
    got = oracle.outprob_cd(g["scores"], g["set_off"], g["states"], code, 3)
    assert np.array_equal(got, g["cd_" + meth])


def test_addlog_edge_cases(oracle):
    # empty list -> LOG_ZERO; single element; equal elements; far-apart elements
    assert oracle.addlog_array(np.zeros(0, np.float32)) == -1000000.0
    assert oracle.addlog_array(np.array([-3.5], np.float32)) == np.float32(-3.5)
    v = oracle.addlog_array(np.array([-2.0, -2.0], np.float32))
    assert abs(v - (-2.0 + np.log(2.0))) < 2e-5
    assert oracle.addlog_array(np.array([-100.0, 0.0], np.float32)) == 0.0


def load_dnn(name):
    z = np.load(GOLDEN / name)
    nl = len(z["dims"]) - 1
    return dict(dims=z["dims"], w=[z[f"w{l}"] for l in range(nl)], b=[z[f"b{l}"] for l in range(nl)],
                prior=z["prior"]), z["frames"], z["out"]


def test_dnn_fma_path(oracle):
    """calc_dnn.c:774 + calc_dnn_fma.c:19 (8 fused partial sums) -- bit-exact."""
    dnn, fr, want = load_dnn("dnn_small.npz")
    assert np.array_equal(oracle.dnn_outprob(dnn, fr, po.DNN_FMA), want)
    # the other reference kernels differ only by rounding: mixed tolerance of SURVEY.md 7.7
    for mode in (po.DNN_AVX, po.DNN_SSE, po.DNN_SCALAR):
        got = oracle.dnn_outprob(dnn, fr, mode)
        assert np.all(np.abs(got - want) <= 1e-4 * np.maximum(1.0, np.abs(want)))


@pytest.mark.parametrize("nbest", [4, 24])
def test_gaussian_mixture_selection(oracle, nbest):
    """gms_state() (gms.c:394-412) on the committed reference outputs, utterance by utterance."""
    z = np.load(GOLDEN / "gms.npz")
    sub = lambda p: {k[len(p):]: z[k] for k in z.files if k.startswith(p)}
    full, gs = sub("full_"), dict(model=sub("gs_"), state2gs=z["state2gs"], nbest=nbest)
    full.update(nbook=0, st_book=None)
    used = z["state2gs"] >= 0
    for a, b in zip(z["utt_off"][:-1], z["utt_off"][1:]):
        fr = z["frames"][a:b]
        got = oracle.gms_apply(gs, fr, oracle.gmm_outprob(full, fr, po.GPRUNE_NONE))
        assert np.array_equal(got[:, used], z["out_%d" % nbest][a:b][:, used])


@pytest.mark.parametrize("num", [5, 20])
def test_verification_gmm(oracle, num):
    """gmm.c's private safe pruning (gmm.c:177-370) on the committed reference outputs: per-frame model
    scores, and the running sums gmm_proceed() leaves in gc->gmm_score[] (float adds in frame order)."""
    z = np.load(GOLDEN / "rejgmm.npz")
    gm = dict(model={k: z[k] for k in ("mean", "ivar", "gconst", "st_off", "ent_dens", "ent_logw")},
              model_state=z["model_state"], gprune_num=num)
    got = oracle.rejgmm_frame_scores(gm, z["frames"])
    assert np.array_equal(got, z["frame_scores_%d" % num])
    for u, (a, b) in enumerate(zip(z["utt_off"][:-1], z["utt_off"][1:])):
        sums = oracle.rejgmm_accumulate(got[a:b])
        assert np.array_equal(sums, z["utt_scores_%d" % num][u])
        assert int(np.argmax(sums)) == int(z["winner_%d" % num][u])
    assert not np.array_equal(z["frame_scores_5"], z["frame_scores_20"])      # -gmmnum 5 really prunes
