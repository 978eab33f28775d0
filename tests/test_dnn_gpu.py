"""GPU: DNN-HMM scoring (MFMA layer GEMMs + table logistic + table log-softmax)
through the C ABI against the reference-generated golden fixture and the oracle.
Bit-exact with the reference's FMA kernel (8 partial sums per output)."""
import numpy as np
import pytest

from conftest import GOLDEN
from julius_amd import lib, synth
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu
TOL = 1e-4   # north_star: 1e-4 relative on float log-likelihoods (mixed abs/rel)


def close(got, want):
    return np.all(np.abs(got - want) <= TOL * np.maximum(1.0, np.abs(want)))


def test_golden_dnn_small(engine):
    z = np.load(GOLDEN / "dnn_small.npz")
    nl = len(z["dims"]) - 1
    dnn = dict(dims=z["dims"], w=[z[f"w{l}"] for l in range(nl)], b=[z[f"b{l}"] for l in range(nl)], prior=z["prior"])
    got = lib.Dnn(engine, dnn).outprob_host(z["frames"])
    assert close(got, z["out"])
    assert np.array_equal(got, z["out"]), f"not bit-exact: max |d| = {np.abs(got - z['out']).max()}"


@pytest.mark.parametrize("dims,T", [((48, 64, 64, 40), 1), ((48, 64, 64, 40), 65), ((40, 72, 72, 72, 33), 130),
                                    ((528, 256, 256, 100), 77), ((16, 8, 8, 5), 9),
                                    # tiles of the 128 x 128 layer kernel: widths / frame counts just past a tile, chains
                                    # with a short last slab (136 / 8 = 17 entries, 1032 / 8 = 129)
                                    ((24, 136, 200, 129), 257), ((64, 1032, 8, 3), 129)])
def test_vs_oracle_shapes(engine, oracle, dims, T):
    dnn = synth.make_dnn(dims=dims, seed=T)
    fr = np.random.default_rng(T).normal(0, 1.5, (T, dims[0])).astype(np.float32)
    got = lib.Dnn(engine, dnn).outprob_host(fr)
    want = oracle.dnn_outprob(dnn, fr, po.DNN_FMA)
    assert close(got, want)
    assert np.array_equal(got, want), f"max |d| = {np.abs(got - want).max()}"


def test_full_size_envr_shape(engine, oracle, ref, tmp_path):
    """BASELINE configs[3] shape: 528 -> 6 x 2048 sigmoid -> 4000 senones, against the COMPILED REFERENCE's
    dnn_calc_outprob() (libsent/src/phmm/calc_dnn.c:774, its FMA kernel calc_dnn_fma.c:19, loaded from the same
    .npy files by dnn_setup()), and against the C restatement."""
    dnn = synth.make_dnn(seed=3)
    T = 96
    fr = np.random.default_rng(5).normal(0, 1.0, (T, 528)).astype(np.float32)
    net = lib.Dnn(engine, dnn)
    got = net.outprob_host(fr)
    if b"FMA" in ref.lib.jref_simd_string():       # the reference picks the best SIMD kernel of the host CPU
        want_ref = ref.dnn_load(dnn, tmp_path, num_threads=1).outprob(fr)
        assert close(got, want_ref)
        assert np.array_equal(got, want_ref), f"vs compiled reference: max |d| = {np.abs(got - want_ref).max()}"
    want = oracle.dnn_outprob(dnn, fr[:40], po.DNN_FMA)
    assert close(got[:40], want)
    assert np.array_equal(got[:40], want), f"max |d| = {np.abs(got[:40] - want).max()}"
    # batch == frame-by-frame (the reference's mode), and softmax normalisation
    big = np.random.default_rng(6).normal(0, 1.0, (300, 528)).astype(np.float32)
    out = net.outprob_host(big)
    assert np.array_equal(net.outprob_host(big[17:18])[0], out[17])
    post = 10.0 ** (out.astype(np.float64) + dnn["prior"].astype(np.float64))
    assert np.allclose(post.sum(axis=1), 1.0, atol=2e-3)


def test_input_length_restriction(engine):
    dnn = synth.make_dnn(dims=(44, 16, 8), seed=1)   # 44 % 8 != 0: same restriction as calc_dnn.c:395
    with pytest.raises(lib.JamdError):
        lib.Dnn(engine, dnn)
