"""GPU: Gaussian mixture selection (-gshmm / -gsnum).  jamd_gms_apply_dev() stands where
gms_state() (libsent/src/phmm/gms.c:394-412) stands in the reference: the full score matrix goes in,
what outprob_state() returns under GMS comes out -- bit for bit, against the committed outputs of
the compiled reference and against the oracle on fresh inputs."""
import numpy as np
import pytest

from conftest import GOLDEN
from julius_amd import lib

pytestmark = pytest.mark.gpu


def _sub(z, prefix):
    return {k[len(prefix):]: z[k] for k in z.files if k.startswith(prefix)}


@pytest.fixture(scope="module")
def fixture():
    z = np.load(GOLDEN / "gms.npz")
    return dict(z=z, full=_sub(z, "full_"), gs=_sub(z, "gs_"))


@pytest.mark.parametrize("strict", [False, True])
@pytest.mark.parametrize("nbest", [4, 24])
def test_golden(engine, fixture, nbest, strict):
    z, used = fixture["z"], fixture["z"]["state2gs"] >= 0
    gm = lib.Gmm(engine, fixture["full"])
    stage = lib.Gms(engine, fixture["gs"], z["state2gs"], nbest).set_strict_order(strict)
    real = gm.outprob_host(z["frames"])
    got = stage.apply_host(z["frames"], real, z["utt_off"])
    want = z["out_%d" % nbest]
    assert np.array_equal(got[:, used], want[:, used])
    assert np.array_equal(got[:, ~used], real[:, ~used])          # unmapped states are left alone
    assert 0.0 < (got != real).mean() < 1.0


@pytest.mark.parametrize("strict", [False, True])
def test_utterance_boundaries(engine, oracle, fixture, strict):
    """The last-best Gaussian restarts at every utterance: a batch equals its utterances one by one,
    and each equals the oracle."""
    z = fixture["z"]
    gm = lib.Gmm(engine, fixture["full"])
    stage = lib.Gms(engine, fixture["gs"], z["state2gs"], 8).set_strict_order(strict)
    real = gm.outprob_host(z["frames"])
    batch = stage.apply_host(z["frames"], real, z["utt_off"])
    gs = dict(model=fixture["gs"], state2gs=z["state2gs"], nbest=8)
    for a, b in zip(z["utt_off"][:-1], z["utt_off"][1:]):
        one = stage.apply_host(z["frames"][a:b], real[a:b])
        assert np.array_equal(one, batch[a:b])
        assert np.array_equal(one, oracle.gms_apply(gs, z["frames"][a:b], real[a:b]))


def test_boundary_tie_goes_to_the_lower_state(engine, oracle, fixture):
    """Two selection states with identical Gaussians tie on every frame.  Whenever the pair straddles
    the nbest boundary the ranking form selects the lower id; the strict form follows the oracle's heap."""
    z, gs = fixture["z"], {k: np.array(v) for k, v in fixture["gs"].items()}
    ids, cnt = np.unique(z["state2gs"][z["state2gs"] >= 0], return_counts=True)
    lo, hi = sorted(int(x) for x in ids[np.argsort(-cnt)[:2]])          # two selection states real states map to
    a0, b0, n = gs["st_off"][lo], gs["st_off"][hi], gs["st_off"][lo + 1] - gs["st_off"][lo]
    assert gs["st_off"][hi + 1] - b0 == n
    for k in ("mean", "ivar", "gconst"):
        gs[k][gs["ent_dens"][b0:b0 + n]] = gs[k][gs["ent_dens"][a0:a0 + n]]
    gs["ent_logw"][b0:b0 + n] = gs["ent_logw"][a0:a0 + n]
    fr = z["frames"]                                    # as ONE utterance: the heap's history matters
    real = lib.Gmm(engine, fixture["full"]).outprob_host(fr)
    want = oracle.gms_apply(dict(model=gs, state2gs=z["state2gs"], nbest=4), fr, real)
    strict = lib.Gms(engine, gs, z["state2gs"], 4).set_strict_order(True).apply_host(fr, real)
    fast = lib.Gms(engine, gs, z["state2gs"], 4).apply_host(fr, real)
    assert np.array_equal(strict, want)                # the reference's choice, whichever of the two it was
    rest = ~np.isin(z["state2gs"], [lo, hi])           # every other selection state: same decision in both forms
    assert np.array_equal(fast[:, rest], want[:, rest])
    on_lo, on_hi = z["state2gs"] == lo, z["state2gs"] == hi
    sel_lo = (fast[:, on_lo] == real[:, on_lo]).all(axis=1)
    sel_hi = (fast[:, on_hi] == real[:, on_hi]).all(axis=1)
    assert not (sel_hi & ~sel_lo).any()                # the higher id is never selected without the lower
    assert (sel_lo & ~sel_hi).any()                    # and the pair did straddle the boundary
    w_lo = (want[:, on_lo] == real[:, on_lo]).all(axis=1)
    w_hi = (want[:, on_hi] == real[:, on_hi]).all(axis=1)
    assert (w_hi & ~w_lo).any()                        # where the reference's heap sometimes took the higher one


def test_more_selected_than_states(engine, oracle, fixture):
    """-gsnum above the number of selection states: every state is selected, nothing is replaced."""
    z = fixture["z"]
    gm = lib.Gmm(engine, fixture["full"])
    stage = lib.Gms(engine, fixture["gs"], z["state2gs"], 1000)
    real = gm.outprob_host(z["frames"][:50])
    assert np.array_equal(stage.apply_host(z["frames"][:50], real), real)


def test_bad_arguments(engine, fixture):
    z = fixture["z"]
    with pytest.raises(lib.JamdError):
        lib.Gms(engine, fixture["gs"], z["state2gs"], 0)
    bad = z["state2gs"].copy()
    bad[0] = 10000
    with pytest.raises(lib.JamdError):
        lib.Gms(engine, fixture["gs"], bad, 8)
    stage = lib.Gms(engine, fixture["gs"], z["state2gs"], 8)
    with pytest.raises(lib.JamdError):
        stage.apply_host(z["frames"][:10], np.zeros((10, stage.S), np.float32), utt_off=[0, 5])
