"""GPU: GMM-based input verification / rejection (-gmm / -gmmnum / -gmmreject).
jamd_rejgmm_* stands where gmm_proceed()'s scoring stands (libjulius/src/gmm.c:574-600 with its
private pruning, gmm.c:177-370): per-frame model scores and the per-input sums, bit for bit against
the committed outputs of the compiled reference and against the oracle on fresh inputs."""
import numpy as np
import pytest

from conftest import GOLDEN
from julius_amd import lib, synth

pytestmark = pytest.mark.gpu

KEYS = ("mean", "ivar", "gconst", "st_off", "ent_dens", "ent_logw")


@pytest.fixture(scope="module")
def z():
    return np.load(GOLDEN / "rejgmm.npz")


@pytest.mark.parametrize("num", [5, 20])
def test_golden(engine, z, num):
    m = lib.RejGmm(engine, {k: z[k] for k in KEYS}, z["model_state"], num)
    fs, us = m.scores_host(z["frames"], z["utt_off"])
    assert np.array_equal(fs, z["frame_scores_%d" % num])
    assert np.array_equal(us, z["utt_scores_%d" % num])
    assert np.array_equal(np.argmax(us, axis=1), z["winner_%d" % num])


@pytest.mark.parametrize("num,M,D,null_frac", [(1, 6, 39, 0.0), (3, 20, 39, 0.2), (10, 40, 26, 0.0), (64, 70, 13, 0.1),
                                                (16, 16, 39, 0.0)])
def test_against_oracle(engine, oracle, num, M, D, null_frac):
    """Mixture counts below, at and above -gmmnum; NULL densities; other vector lengths."""
    model = synth.make_gmm(S=5, M=M, D=D, seed=num + M, ragged=True, null_frac=null_frac)
    order = np.array([4, 0, 3, 1, 2], np.int32)       # the model list is not in state order (gmm->start)
    fr = synth.make_frames(model, T=257, seed=3)
    gm = dict(model=model, model_state=order, gprune_num=num)
    want = oracle.rejgmm_frame_scores(gm, fr)
    off = np.array([0, 100, 101, 257], np.int32)
    fs, us = lib.RejGmm(engine, model, order, num).scores_host(fr, off)
    assert np.array_equal(fs, want)
    for u in range(3):
        assert np.array_equal(us[u], oracle.rejgmm_accumulate(want[off[u]:off[u + 1]]))


def test_bad_arguments(engine, z):
    model = {k: z[k] for k in KEYS}
    with pytest.raises(lib.JamdError):
        lib.RejGmm(engine, model, z["model_state"], 0)
    with pytest.raises(lib.JamdError):
        lib.RejGmm(engine, model, [0, 99], 5)
    m = lib.RejGmm(engine, model, z["model_state"], 5)
    with pytest.raises(lib.JamdError):
        m.scores_host(z["frames"][:10], [0, 5])
