"""GPU: the HIP GMM kernels (through the C ABI) against (1) the golden fixtures
produced by the compiled reference and (2) the oracle on fresh seeded inputs.
Bit-exact (fp32 scores are computed in the reference's operation order)."""
import numpy as np
import pytest

from conftest import GOLDEN
from julius_amd import lib, synth
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu


def load(name):
    z = np.load(GOLDEN / name)
    m = {k: z[k] for k in z.files}
    m["nbook"] = int(m.get("nbook", 0))
    if "book_size" in m:
        m["book_size"] = int(m["book_size"])
    m.setdefault("st_book", None)
    return m


def test_golden_plain(engine):
    """Both forms of K1 against the compiled reference's golden scores: the narrow form (calls of <= 256 frames: one lane per
    mixture entry, csrc/gmm_outprob.hip K1n) and the tile kernel (one lane per frame) on the same frames repeated."""
    g = load("gmm_plain_none.npz")
    gm = lib.Gmm(engine, g)
    T = len(g["frames"])
    out = gm.outprob_host(g["frames"])
    assert ("gmm_narrow<D=39" if T <= 256 else "gmm_tile<D=39") in gm.last_kernel()
    assert np.array_equal(out, g["out"])
    reps = 256 // T + 2
    big = gm.outprob_host(np.tile(g["frames"], (reps, 1)))
    assert "gmm_tile<D=39" in gm.last_kernel()
    for r in range(reps):
        assert np.array_equal(big[r * T:(r + 1) * T], g["out"])


@pytest.mark.parametrize("D,ragged,T", [(39, False, 1), (39, False, 25), (39, True, 2), (39, True, 255), (39, True, 256),
                                          (26, True, 33), (38, False, 100), (25, True, 7)])
def test_narrow_call_equals_tile_kernel(engine, oracle, D, ragged, T):
    """A call of a handful of frames (live input, a streaming chunk) takes the narrow form of K1; the same frames inside a
    long call take the tile kernel: the scores are the same floats, and the oracle's."""
    m = synth.make_gmm(S=70, M=11, D=D, seed=900 + T, ragged=ragged, null_frac=0.08 if ragged else 0.0)
    fr = synth.make_frames(m, T=600, seed=T)
    gm = lib.Gmm(engine, m)
    wide = gm.outprob_host(fr)
    assert "gmm_tile<D=" in gm.last_kernel()
    for start in (0, 300):
        narrow = gm.outprob_host(fr[start:start + T])
        assert "gmm_narrow<D=" in gm.last_kernel()
        assert np.array_equal(narrow, wide[start:start + T])
    assert np.array_equal(gm.outprob_host(fr[:T]), oracle.gmm_outprob(m, fr[:T]))


def test_golden_ragged_null_densities(engine):
    g = load("gmm_ragged.npz")
    gm = lib.Gmm(engine, g)
    assert np.array_equal(gm.outprob_host(g["frames"]), g["out"])


@pytest.mark.parametrize("S,M,D,T,ragged", [
    (48, 16, 39, 1, False),        # single frame (frame-synchronous call)
    (48, 16, 39, 513, False),      # one block + 1 frame
    (100, 16, 39, 1000, False),    # one utterance
    (33, 7, 26, 300, True),        # templated D=26, ragged mixtures
    (20, 3, 13, 129, True),        # generic-D kernel
    (17, 5, 60, 70, False),        # generic-D kernel, D > 39
    (3, 1, 39, 64, False),         # single Gaussian per state
])
def test_vs_oracle(engine, oracle, S, M, D, T, ragged):
    m = synth.make_gmm(S=S, M=M, D=D, seed=S + T, ragged=ragged, null_frac=0.05 if ragged else 0.0)
    fr = synth.make_frames(m, T=T, seed=T)
    gm = lib.Gmm(engine, m)
    got = gm.outprob_host(fr)
    want = oracle.gmm_outprob(m, fr)
    assert got.shape == want.shape
    assert np.array_equal(got, want), f"max abs diff {np.abs(got - want).max()}"


def test_far_frames_hit_log_zero_and_cutoff(engine, oracle):
    """Frames far from every Gaussian: scores fall through the LOG_ADDMIN cutoff
    (addlog.c:114) and stay finite; an exact-zero log-sum maps to LOG_ZERO
    (calc_mix.c:78)."""
    m = synth.make_gmm(S=12, M=4, D=39, seed=1)
    fr = synth.make_frames(m, T=64, seed=2)
    fr[::2] *= 50.0
    gm = lib.Gmm(engine, m)
    assert np.array_equal(gm.outprob_host(fr), oracle.gmm_outprob(m, fr))


def test_full_size_properties(engine, oracle):
    """BASELINE configs[1] size (S=3000 x M=16 x D=39): the oracle is too slow for
    the whole matrix, so check (a) a random sample of (t, s) entries against the
    oracle restricted to those states, (b) frame-permutation equivariance and
    (c) that a batch equals its frames scored one by one."""
    m = synth.make_gmm(S=3000, M=16, D=39, seed=7)
    T = 1024
    fr = synth.make_frames(m, T=T, seed=8)
    gm = lib.Gmm(engine, m)
    out = gm.outprob_host(fr)
    assert out.shape == (T, 3000) and np.isfinite(out).all()
    rng = np.random.default_rng(0)
    ss = np.sort(rng.choice(3000, 40, replace=False))
    sub = dict(m)
    sub["st_off"] = np.concatenate([[0], np.cumsum(m["st_off"][ss + 1] - m["st_off"][ss])]).astype(np.int32)
    idx = np.concatenate([np.arange(m["st_off"][s], m["st_off"][s + 1]) for s in ss])
    sub["ent_dens"], sub["ent_logw"] = m["ent_dens"][idx], m["ent_logw"][idx]
    tt = np.sort(rng.choice(T, 50, replace=False))
    assert np.array_equal(out[np.ix_(tt, ss)], oracle.gmm_outprob(sub, fr[tt]))
    perm = rng.permutation(T)
    assert np.array_equal(gm.outprob_host(fr[perm]), out[perm])
    for t in (0, 511, 512, T - 1):
        assert np.array_equal(gm.outprob_host(fr[t:t + 1])[0], out[t])


def test_full_size_rows_vs_compiled_reference(engine, ref):
    """BASELINE configs[1] size against the COMPILED REFERENCE (libjref.so: outprob_state batch loop ->
    calc_mix -> gprune_none -> addlog_array, `outprob.c:230-242`): 64 complete rows of 3000 states, bit for bit."""
    m = synth.make_gmm(S=3000, M=16, D=39, seed=7)
    T = 2048
    fr = synth.make_frames(m, T=T, seed=9)
    out = lib.Gmm(engine, m).outprob_host(fr)
    tt = np.sort(np.random.default_rng(1).choice(T, 64, replace=False))
    want = ref.am_from_flat(m).outprob(fr[tt], want_out=True)
    assert want.shape == (64, 3000)
    assert np.array_equal(out[tt], want)


@pytest.mark.parametrize("method,code", [("heu", lib.GPRUNE_HEU), ("beam", lib.GPRUNE_BEAM)])
@pytest.mark.parametrize("num", [2, 5])
def test_gprune_heu_beam_on_plain_states_vs_compiled_reference(engine, ref, method, code, num):
    """SURVEY 8a A7, the plain-mixture half: calc_mix() calls the pruning function with last_id == NULL
    (`calc_mix.c:63`), where gprune_heu() / gprune_beam() are safe pruning (`gprune_heu.c:337-350`,
    `gprune_beam.c:337-350`).  The device serves such models with these methods, bit for bit with the compiled
    reference running the real gprune_heu / gprune_beam."""
    m = synth.make_gmm(S=90, M=12, D=39, seed=21, ragged=True, null_frac=0.05)
    fr = synth.make_frames(m, T=300, seed=22)
    want = ref.am_from_flat(m, gprune=method, gprune_num=num).outprob(fr, want_out=True)
    got = lib.Gmm(engine, m, code, num).outprob_host(fr)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("method,code", [("heu", lib.GPRUNE_HEU), ("beam", lib.GPRUNE_BEAM)])
@pytest.mark.parametrize("n,nbook,K,D,noise", [(1, 3, 64, 39, 2.0), (2, 4, 64, 39, 2.0), (4, 2, 129, 39, 1.0),
                                               (10, 1, 40, 25, 3.0), (64, 2, 24, 13, 2.0), (2, 129, 16, 39, 2.0)])
def test_gprune_heu_beam_on_tied_mixture_vs_compiled_reference(engine, ref, tmp_path, method, code, n, nbook, K, D, noise):
    """SURVEY 8a A7, the tied-mixture half -- the case where these functions do something of their own, and the fast
    build's default for C1-type models: thresholds from the codebook's winners of frame t-1 (`gprune_heu.c:305-335`,
    `gprune_beam.c:301-336`, hand-over `calc_tied_mix.c:203-215`).  Parity is defined against the compiled reference
    under EAGER scoring (`outprob_set_batch_computation`, `outprob.c:230-242`: every state of every frame, so frame t-1's
    cache always exists): state scores and the MIXCACHE contents, bit for bit; a batch call restarts the history at
    every utterance boundary."""
    m = synth.make_tied_gmm(S=max(21, nbook), nbook=nbook, K=K, D=D, seed=K + n)
    synth.write_hmmdefs(tmp_path / "h", m, kind="MFCC_E_D_A" if D == 39 else "USER")
    am = ref.am_load(tmp_path / "h", gprune=method, gprune_num=n)
    assert am.is_tied and am.nbook == nbook
    ex = am.export()
    fr = synth.make_frames(m, T=75, seed=9 + n, noise=noise)
    gm = lib.Gmm(engine, ex, code, n)
    want = am.outprob(fr)
    got = gm.outprob_host(fr)
    assert np.array_equal(got, want), f"max |d| = {np.abs(got - want).max()}"
    sc, ids, num = gm.tmix_cache_host(fr)
    for b in range(min(nbook, 4)):
        rsc, rids, rnum = am.tmix_cache(fr, b, n)
        assert np.array_equal(num[:, b], rnum)
        for t in range(len(fr)):
            k = rnum[t]
            assert np.array_equal(ids[t, b, :k], rids[t, :k]) and np.array_equal(sc[t, b, :k], rsc[t, :k]), (b, t)
    # two utterances in one call: the second one starts without history, as outprob_prepare() leaves it
    fr2 = synth.make_frames(m, T=40, seed=77, noise=noise)
    both = np.concatenate([fr, fr2])
    d_fr = lib.DevBuf(engine, both.nbytes).upload(both)
    d_out = lib.DevBuf(engine, 4 * len(both) * gm.S)
    gm.outprob_utts_dev(d_fr.ptr, [0, len(fr), len(both)], d_out.ptr)
    engine.sync()
    out = d_out.download((len(both), gm.S), np.float32)
    assert np.array_equal(out[:len(fr)], want) and np.array_equal(out[len(fr):], am.outprob(fr2))
    am.close()


def test_tied_mixture_more_than_65535_frames(engine):
    """A launch over more frames than the grid's y dimension allows (a batch of utterances easily is): the
    per-frame kernels of the tied-mixture and state-set paths stride over the frames; row t of the long call
    equals the same frame scored alone."""
    tied = synth.make_tied_gmm(S=24, nbook=2, K=16, D=39, seed=5)
    base = synth.make_frames(tied, T=700, seed=6)
    T = 70000
    fr = np.tile(base, (T // len(base), 1))
    gm = lib.Gmm(engine, tied, lib.GPRUNE_SAFE, 2)
    out = gm.outprob_host(fr)
    small = gm.outprob_host(base)
    assert out.shape == (T, 24)
    for r in (0, 1, 99):
        assert np.array_equal(out[r * 700:(r + 1) * 700], small)
    got = lib.CdSet(engine, np.array([0, 3, 7], np.int32), np.arange(7, dtype=np.int32), lib.IWCD_MAX, 3).outprob_host(out)
    assert np.array_equal(got[:, 0], out[:, 0:3].max(1)) and np.array_equal(got[:, 1], out[:, 3:7].max(1))


def test_device_pointer_entry(engine, oracle):
    """jamd_gmm_outprob_dev with caller-owned device buffers and stream (torch is
    only the allocator here)."""
    import torch
    m = synth.make_gmm(S=40, M=8, D=39, seed=3)
    fr = synth.make_frames(m, T=200, seed=4)
    gm = lib.Gmm(engine, m)
    d_fr = torch.from_numpy(fr).cuda()
    d_out = torch.empty((200, 40), dtype=torch.float32, device="cuda")
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        gm.outprob_dev(d_fr.data_ptr(), 200, d_out.data_ptr(), st.cuda_stream)
    st.synchronize()
    assert np.array_equal(d_out.cpu().numpy(), oracle.gmm_outprob(m, fr))


def test_bad_arguments_fail_loudly(engine):
    m = synth.make_gmm(S=6, M=2, D=39, seed=1)
    m2 = dict(m)
    m2["nstream"] = 2
    with pytest.raises(lib.JamdError):
        lib.Gmm(engine, m2)
    with pytest.raises(lib.JamdError):
        lib.Gmm(engine, m, gprune=3)


# ----------------------------------------------------------- pruned / tied / cd
def test_golden_safe_plain(engine):
    g = load("gmm_ragged.npz")
    gm = lib.Gmm(engine, g, gprune=lib.GPRUNE_SAFE, gprune_num=3)
    assert np.array_equal(gm.outprob_host(g["frames"]), g["out_safe3"])


@pytest.mark.parametrize("key,gp,n", [("out_none", lib.GPRUNE_NONE, 32), ("out_safe2", lib.GPRUNE_SAFE, 2),
                                      ("out_safe4", lib.GPRUNE_SAFE, 4)])
def test_golden_tied(engine, key, gp, n):
    g = load("gmm_tied.npz")
    gm = lib.Gmm(engine, g, gprune=gp, gprune_num=n)
    assert np.array_equal(gm.outprob_host(g["frames"]), g[key])


def test_golden_tied_codebook_cache(engine):
    g = load("gmm_tied.npz")
    gm = lib.Gmm(engine, g, gprune=lib.GPRUNE_SAFE, gprune_num=2)
    sc, ids, num = gm.tmix_cache_host(g["frames"])
    assert np.array_equal(num[:, 1], g["cache2_num"])
    assert np.array_equal(ids[:, 1, :], g["cache2_id"])
    assert np.array_equal(sc[:, 1, :], g["cache2_score"])


@pytest.mark.parametrize("S,M,D,T,n", [(40, 16, 39, 300, 1), (40, 16, 39, 300, 5), (25, 9, 26, 130, 9),
                                       (12, 20, 13, 200, 12), (12, 20, 13, 200, 40)])
def test_safe_vs_oracle(engine, oracle, S, M, D, T, n):
    m = synth.make_gmm(S=S, M=M, D=D, seed=S + n, ragged=True, null_frac=0.05)
    fr = synth.make_frames(m, T=T, seed=n)
    gm = lib.Gmm(engine, m, gprune=lib.GPRUNE_SAFE, gprune_num=n)
    assert np.array_equal(gm.outprob_host(fr), oracle.gmm_outprob(m, fr, po.GPRUNE_SAFE, n))


@pytest.mark.parametrize("D", [16, 24, 32, 48, 60])
def test_generic_veclen_needs_more_than_64k_of_lds(engine, oracle, D):
    """Vector lengths without a specialised kernel keep their frames in dynamic LDS (2 KB per component) next to the
    34 KB output tile: from D = 16 on that passes the default 64 KB window (hipFuncAttributeMaxDynamicSharedMemorySize
    must be raised), safe pruning and plain scoring alike; beyond the CU's 160 KB the library refuses with a message."""
    m = synth.make_gmm(S=20, M=6, D=D, seed=D, ragged=True)
    fr = synth.make_frames(m, T=300, seed=D)
    assert np.array_equal(lib.Gmm(engine, m).outprob_host(fr), oracle.gmm_outprob(m, fr))
    gm = lib.Gmm(engine, m, gprune=lib.GPRUNE_SAFE, gprune_num=3)
    assert np.array_equal(gm.outprob_host(fr), oracle.gmm_outprob(m, fr, po.GPRUNE_SAFE, 3))
    t = synth.make_tied_gmm(S=12, nbook=2, K=24, D=D, seed=D)
    tm = lib.Gmm(engine, t, gprune=lib.GPRUNE_SAFE, gprune_num=4)
    assert np.array_equal(tm.outprob_host(fr), oracle.gmm_outprob(t, fr, po.GPRUNE_SAFE, 4))


def test_vector_length_beyond_the_lds_is_refused(engine):
    m = synth.make_gmm(S=4, M=2, D=96, seed=1)
    with pytest.raises(lib.JamdError, match="LDS"):
        lib.Gmm(engine, m).outprob_host(synth.make_frames(m, T=8, seed=1))


@pytest.mark.parametrize("nbook,K,D,T,gp,n", [(3, 64, 39, 260, "safe", 2), (1, 256, 39, 100, "safe", 4),
                                              (5, 33, 25, 140, "safe", 8), (2, 40, 13, 70, "safe", 3),
                                              (3, 64, 39, 130, "none", 64)])
def test_tied_vs_oracle(engine, oracle, nbook, K, D, T, gp, n):
    m = synth.make_tied_gmm(S=30, nbook=nbook, K=K, D=D, seed=K)
    fr = synth.make_frames(m, T=T, seed=T, noise=2.0)
    code = lib.GPRUNE_NONE if gp == "none" else lib.GPRUNE_SAFE
    gm = lib.Gmm(engine, m, gprune=code, gprune_num=n)
    assert np.array_equal(gm.outprob_host(fr), oracle.gmm_outprob(m, fr, code, n))


def test_compound_model(engine, oracle):
    """Mixed tied / plain states (calc_compound_mix, calc_tied_mix.c:258)."""
    m = synth.make_tied_gmm(S=18, nbook=2, K=32, D=39, seed=9)
    p = synth.make_gmm(S=12, M=4, D=39, seed=10)
    G0 = m["mean"].shape[0]
    mix = dict(
        mean=np.concatenate([m["mean"], p["mean"]]), ivar=np.concatenate([m["ivar"], p["ivar"]]),
        gconst=np.concatenate([m["gconst"], p["gconst"]]),
        st_off=np.concatenate([m["st_off"], m["st_off"][-1] + p["st_off"][1:]]).astype(np.int32),
        ent_dens=np.concatenate([m["ent_dens"], p["ent_dens"] + G0]).astype(np.int32),
        ent_logw=np.concatenate([m["ent_logw"], p["ent_logw"]]),
        st_book=np.concatenate([m["st_book"], -np.ones(12, np.int32)]).astype(np.int32), nbook=2, nstream=1)
    fr = synth.make_frames(m, T=150, seed=3, noise=2.0)
    for code, n in ((lib.GPRUNE_SAFE, 2), (lib.GPRUNE_NONE, 32)):
        gm = lib.Gmm(engine, mix, gprune=code, gprune_num=n)
        assert np.array_equal(gm.outprob_host(fr), oracle.gmm_outprob(mix, fr, code, n))


@pytest.mark.parametrize("meth,code", [("max", lib.IWCD_MAX), ("avg", lib.IWCD_AVG), ("nbest", lib.IWCD_NBEST)])
def test_golden_cdset(engine, meth, code):
    g = load("cdset.npz")
    cd = lib.CdSet(engine, g["set_off"], g["states"], code, 3)
    assert np.array_equal(cd.outprob_host(g["scores"]), g["cd_" + meth])


def test_cdset_vs_oracle_with_log_zero_members(engine, oracle):
    rng = np.random.default_rng(5)
    S, T = 200, 77
    scores = rng.normal(-40, 10, size=(T, S)).astype(np.float32)
    scores[rng.random((T, S)) < 0.2] = -1000000.0
    sizes = rng.integers(1, 40, size=50)
    set_off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    states = np.concatenate([rng.choice(S, size=n, replace=False) for n in sizes]).astype(np.int32)
    for code, nb in ((lib.IWCD_MAX, 3), (lib.IWCD_AVG, 3), (lib.IWCD_NBEST, 1), (lib.IWCD_NBEST, 3), (lib.IWCD_NBEST, 16)):
        got = lib.CdSet(engine, set_off, states, code, nb).outprob_host(scores)
        want = oracle.outprob_cd(scores, set_off, states, code, nb)
        assert np.array_equal(got, want, equal_nan=True), (code, nb)
