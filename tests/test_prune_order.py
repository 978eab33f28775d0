"""The rank-pruning step of the exact-order first-pass kernel (beam_exact.hip: level-parallel heapify +
closed-form extraction with replayed tail events) against the oracle's sequential restatement of
sort_token_no_order() (libjulius/src/beam.c:1342-1516), on score vectors full of exact ties."""
import numpy as np
import pytest

from beamutil import load_beam_golden
from julius_amd import lib

pytestmark = pytest.mark.gpu


def _scores(rng, n, levels):
    if levels == 0:                       # all distinct
        return rng.permutation(n).astype(np.float32) * -0.37 - 100.0
    return (-rng.integers(0, levels, n).astype(np.float32) * 0.5 - 2000.0).astype(np.float32)


@pytest.mark.parametrize("mode", ["exact", "exact_serial"])
@pytest.mark.parametrize("beam", [1, 2, 7, 33, 200, 800, 1500])
def test_prune_order_fuzz(engine, oracle, beam, mode):
    g = load_beam_golden("beam_rank.npz")
    lx = lib.Lexicon(engine, g["lex"])
    bm = lib.Beam(engine, lx, beam, -1.0, max_utts=1).set_order_mode(mode)
    rng = np.random.default_rng(beam)
    sizes = sorted(set([1, 2, 3, beam, beam + 1, 2 * beam, 2 * beam + 1, 2 * beam + 2, 3 * beam + 5, 5 * beam + 17] +
                       [int(x) for x in rng.integers(1, max(8 * beam, 64), 24)]))
    for n in sizes:
        for levels in (0, 2, 5, 40, 1000):
            sc = _scores(rng, n, levels)
            got = bm.prune_order(sc)
            want = oracle.sort_token_no_order(sc, beam)
            assert np.array_equal(got, want), (n, beam, levels, mode)


def test_prune_order_large_frames(engine, oracle):
    """Frames of the size the 20k-word task produces (thousands of tokens into a beam of 800), a few percent
    of the tokens in tie groups, plus frames larger than the LDS heap."""
    g = load_beam_golden("beam_rank.npz")
    lx = lib.Lexicon(engine, g["lex"])
    bm = lib.Beam(engine, lx, 800, -1.0, max_utts=1)
    rng = np.random.default_rng(5)
    for n in (1700, 2900, 4200, 6000, 9000, 16000, 40000):
        for dup in (0.0, 0.08, 0.5):
            base = (-rng.random(n) * 300.0 - 5000.0).astype(np.float32)
            ndup = int(dup * n)
            if ndup:
                base[rng.integers(0, n, ndup)] = base[rng.integers(0, n, ndup)]
            got = bm.prune_order(base)
            want = oracle.sort_token_no_order(base, 800)
            assert np.array_equal(got, want), (n, dup)
