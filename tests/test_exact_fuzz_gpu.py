"""Exact-order first-pass kernel against the CPU restatement (pinned to the compiled reference by
tests/test_beam_oracle.py) on a sweep of small random tasks chosen to be FULL of exact score ties: few tied states
(many lexicon branches score identically), quantised acoustic scores, beams from 1 to a few hundred so that the
rank-pruning step goes through all of its forms (nothing pruned, sort_token_downward, sort_token_upward with the
closed-form extraction, serial fallback), with and without a score beam, in both workgroup shapes.  The word trellis
must be identical."""
import numpy as np
import pytest

from beamutil import assert_trellis_equal
from julius_amd import lexblob, lib, synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", ["full", "half"])
@pytest.mark.parametrize("seed", range(12))
def test_exact_kernel_vs_oracle_on_tie_heavy_tasks(engine, oracle, seed, shape):
    rng = np.random.default_rng(1000 + seed)
    nphone = int(rng.choice([4, 6, 10]))
    S = nphone * int(rng.choice([3, 6, 9]))
    nword = int(rng.choice([40, 150, 500]))
    lex = synth.make_lexicon(nword=nword, nphone=nphone, S=S, seed=seed, minlen=1, maxlen=5, sepnum=int(rng.choice([0, 5, 30])),
                             nshort=3, nbigram_per_word=4)
    lx = lib.Lexicon(engine, lex)
    T = int(rng.integers(30, 140))
    # quantised scores: exact ties between different states as well
    step = float(rng.choice([0.5, 2.0, 8.0]))
    scores = [(-np.round(rng.random((T, S)) * 40.0 / step) * step - 20.0).astype(np.float32) for _ in range(3)]
    for beam in (1, 3, int(rng.integers(5, 40)), int(rng.integers(40, 400))):
        for width in (-1.0, float(rng.choice([30.0, 80.0]))):
            bm = lib.Beam(engine, lx, beam, width, max_utts=len(scores))
            assert bm.order_mode() == "exact"
            bm.set_workgroup_shape(shape)              # both workgroup shapes of the kernel (julius_amd.h, JAMD_SHAPE_*)
            assert bm.workgroup_shape(len(scores)) == shape
            res, tre = bm.pass1_host(scores)
            for sc, r, atoms in zip(scores, res, tre):
                oatoms, owseq, oscore, rc, died = oracle.beam_pass1(lex, sc, beam, width)
                assert r.status == rc, (seed, beam, width)
                assert_trellis_equal(atoms, lexblob.canonical_trellis(oatoms))
                if rc == 0:
                    assert list(r.wseq[:r.wnum]) == list(owseq) and r.score == oscore
                if rc == 2:
                    assert r.died_at == died
            bm.close()
