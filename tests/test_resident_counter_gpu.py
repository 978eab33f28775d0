"""GPU: the bookkeeping of the resident counter (csrc/beam.hip mark_started() / account_launch()) -- the device word the
first-pass workgroups bump when they start, on which the scoring stream of a pipelining host waits
(jamd_beam_stream_wait_resident(): hipStreamWaitValue32, no host involvement).  Two fixes of round 5 that had
no test: the counter is drained back to zero before its 32 bits could wrap, and a REFUSED launch
accounts nothing -- a phantom workgroup would make every later wait hang."""
import numpy as np
import pytest

from beamutil import assert_trellis_equal, load_beam_golden
from julius_amd import lib

pytestmark = pytest.mark.gpu


def _device_scores(engine, oracle, g):
    scores = [oracle.gmm_outprob(g["am"], u["frames"]) for u in g["utts"]]
    S = scores[0].shape[1]
    off = np.zeros(len(scores) + 1, np.int32)
    off[1:] = np.cumsum([len(s) for s in scores])
    allsc = np.concatenate(scores).astype(np.float32)
    return lib.DevBuf(engine, allsc.nbytes).upload(allsc), S, off


def _check_golden(bm, g):
    for i, (r, u) in enumerate(zip(bm.results(), g["utts"])):
        assert r.status == 0 and r.score == u["score"]
        assert np.array_equal(np.array(r.wseq[:r.wnum]), u["wseq"])
        assert_trellis_equal(bm.trellis(i), u["trellis"])


@pytest.mark.timeout(300)
def test_resident_counter_is_drained_before_it_wraps(engine, oracle):
    import torch
    g = load_beam_golden("beam_rank.npz")
    nu = len(g["utts"])
    lx = lib.Lexicon(engine, g["lex"])
    bm = lib.Beam(engine, lx, g["beam_width"], g["score_pruning_width"], max_utts=nu)
    d_sc, S, off = _device_scores(engine, oracle, g)
    s_score, s_beam = torch.cuda.Stream(), torch.cuda.Stream()
    x = torch.zeros(1, device="cuda")
    bm.debug_preset_resident(0x7FFFFFF0)                       # a few launches below the point where the counter is drained
    drained = 0
    for it in range(16 // nu + 6):
        before = bm.debug_resident()[0]
        bm.pass1_dev(d_sc.ptr, S, off, s_beam.cuda_stream)
        launched, target = bm.debug_resident()
        if launched < before:
            drained += 1
            assert launched == nu and 1 <= target <= nu            # counted again from zero
        else:
            assert launched == before + nu and before < target <= launched
        # the pipelining host's pattern: work queued on another stream behind the wait starts once this launch holds its CUs
        bm.stream_wait_resident(s_score.cuda_stream)
        with torch.cuda.stream(s_score):
            x += 1
        torch.cuda.synchronize()                                   # (a wait for a value the counter never reaches would hang here)
        _check_golden(bm, g)
    assert drained == 1
    assert int(x.item()) == 16 // nu + 6


@pytest.mark.timeout(300)
def test_refused_launch_accounts_nothing(engine, oracle):
    import torch
    g = load_beam_golden("beam_multipath.npz")
    nu = len(g["utts"])
    lx = lib.Lexicon(engine, g["lex"])
    bm = lib.Beam(engine, lx, g["beam_width"], g["score_pruning_width"], max_utts=nu)
    d_sc, S, off = _device_scores(engine, oracle, g)
    s_score, s_beam = torch.cuda.Stream(), torch.cuda.Stream()
    x = torch.zeros(1, device="cuda")
    bm.pass1_dev(d_sc.ptr, S, off, s_beam.cuda_stream)             # an accepted launch first: the wait below has an event to refer to
    torch.cuda.synchronize()
    base = bm.debug_resident()
    assert base[0] == nu
    bm.set_order_mode("fast")                                      # the canonical-tie kernel does not take multipath lexicons:
    with pytest.raises(lib.JamdError):                              # JAMD_ESTATE before anything is enqueued or accounted
        bm.pass1_dev(d_sc.ptr, S, off, s_beam.cuda_stream)
    assert bm.debug_resident() == base
    bm.stream_wait_resident(s_score.cuda_stream)                   # still satisfiable: it refers to the accepted launch
    with torch.cuda.stream(s_score):
        x += 1
    torch.cuda.synchronize()
    bm.set_order_mode("exact")
    bm.pass1_dev(d_sc.ptr, S, off, s_beam.cuda_stream)             # accepted again on the same work area, the scoring stream waiting on the counter
    bm.stream_wait_resident(s_score.cuda_stream)
    with torch.cuda.stream(s_score):
        x += 1
    torch.cuda.synchronize()
    launched, target = bm.debug_resident()
    assert launched == 2 * nu and nu < target <= 2 * nu
    assert int(x.item()) == 2
    _check_golden(bm, g)
