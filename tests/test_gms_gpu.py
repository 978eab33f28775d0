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


@pytest.mark.parametrize("nbest", [4, 24])
def test_golden(engine, fixture, nbest):
    z, used = fixture["z"], fixture["z"]["state2gs"] >= 0
    gm = lib.Gmm(engine, fixture["full"])
    stage = lib.Gms(engine, fixture["gs"], z["state2gs"], nbest)
    real = gm.outprob_host(z["frames"])
    got = stage.apply_host(z["frames"], real, z["utt_off"])
    want = z["out_%d" % nbest]
    assert np.array_equal(got[:, used], want[:, used])
    assert np.array_equal(got[:, ~used], real[:, ~used])          # unmapped states are left alone
    assert 0.0 < (got != real).mean() < 1.0


def test_utterance_boundaries(engine, oracle, fixture):
    """The last-best Gaussian restarts at every utterance: a batch equals its utterances one by one,
    and each equals the oracle."""
    z = fixture["z"]
    gm = lib.Gmm(engine, fixture["full"])
    stage = lib.Gms(engine, fixture["gs"], z["state2gs"], 8)
    real = gm.outprob_host(z["frames"])
    batch = stage.apply_host(z["frames"], real, z["utt_off"])
    gs = dict(model=fixture["gs"], state2gs=z["state2gs"], nbest=8)
    for a, b in zip(z["utt_off"][:-1], z["utt_off"][1:]):
        one = stage.apply_host(z["frames"][a:b], real[a:b])
        assert np.array_equal(one, batch[a:b])
        assert np.array_equal(one, oracle.gms_apply(gs, z["frames"][a:b], real[a:b]))


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
