"""The sweep replay of the rank-pruning step (csrc/beam_sweep.h): hundreds of tail candidates resolved together --
iterated sweeps over the closed form's move table instead of the extraction loop of sort_token_upward()
(libjulius/src/beam.c:1368-1383) -- against the oracle's sequential restatement, on real frames of the C4 task (beam
4000, tests/golden/prune_frames_c4.npz: token scores in creation order, dumped from the oracle's first pass by
tools/dump_prune_inputs.py) and on synthetic frames of that shape with sprinkled exact ties."""
from pathlib import Path

import numpy as np
import pytest

from beamutil import load_beam_golden
from julius_amd import lib

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).parent / "golden"


def _beam(engine, beam, mode="exact"):
    g = load_beam_golden("beam_rank.npz")
    lx = lib.Lexicon(engine, g["lex"])
    return lib.Beam(engine, lx, beam, -1.0, max_utts=1).set_order_mode(mode)


def test_real_frames_of_the_c4_task(engine, oracle):
    z = np.load(GOLD / "prune_frames_c4.npz")
    beam = int(z["beam"])
    bm = _beam(engine, beam)
    swept = 0
    for name in sorted(k for k in z.files if k.startswith("f")):
        sc = z[name]
        got = bm.prune_order(sc)
        info = bm.prune_info()
        want = oracle.sort_token_no_order(sc, beam)
        assert np.array_equal(got, want), (name, len(sc), info)
        swept += info > 0
    assert swept >= 6, swept            # frames with a tied element on a tail position, and every downward frame
    bm.close()


@pytest.mark.parametrize("beam", [2000, 4000])
def test_sweep_on_synthetic_frames(engine, oracle, beam):
    """Frames 2.05 - 4.5 beams large (the upward sort), continuous scores with a fraction of exact duplicates: hundreds of
    tail candidates, some of them tied -- the sweep has to run, converge and give the sequential loop's order."""
    bm = _beam(engine, beam)
    rng = np.random.default_rng(beam + 1)
    stats = []
    for mult in (2.05, 2.3, 2.6, 3.1, 3.7, 4.5):
        n = int(mult * beam) + int(rng.integers(0, 50))
        for dup in (0.002, 0.01, 0.05, 0.3):
            sc = (-rng.random(n) * 300.0 - 5000.0).astype(np.float32)
            nd = max(2, int(dup * n))
            sc[rng.integers(0, n, nd)] = sc[rng.integers(0, n, nd)]
            got = bm.prune_order(sc)
            info = bm.prune_info()
            want = oracle.sort_token_no_order(sc, beam)
            assert np.array_equal(got, want), (n, beam, dup, info)
            stats.append(info)
    assert sum(1 for s in stats if s > 0) >= len(stats) // 2, stats
    bm.close()


def test_sweep_with_heavy_ties(engine, oracle):
    """Few score levels: large tie groups with landed members (release-time order inside a group), groups too large for
    the sweep (it hands the frame to the extraction loop) -- the result is the sequential loop's either way."""
    beam = 3000
    bm = _beam(engine, beam)
    rng = np.random.default_rng(11)
    seen = set()
    for n in (6200, 7500, 9000, 12000):
        for levels in (40, 400, 4000, 40000):
            sc = (-rng.integers(0, levels, n).astype(np.float32) * 0.5 - 2000.0).astype(np.float32)
            got = bm.prune_order(sc)
            info = bm.prune_info()
            want = oracle.sort_token_no_order(sc, beam)
            assert np.array_equal(got, want), (n, levels, info)
            seen.add(1 if info > 0 else info)
    assert 1 in seen, seen
    bm.close()


@pytest.mark.parametrize("beam", [1500, 4000])
def test_downward_sort_through_the_sweep(engine, oracle, beam):
    """beam < tokens <= 2 beam: sort_token_downward() (beam.c:1414-1457) -- the survivors are the residual MIN-heap after
    n - beam extractions, in heap layout.  Device: closed form over the extracted elements + replay of the sifts below
    them (down_finish()); frames with few and with many events, ties on the cut, ties everywhere."""
    bm = _beam(engine, beam)
    rng = np.random.default_rng(beam + 7)
    stats = []
    for mult in (1.02, 1.2, 1.45, 1.7, 1.9, 1.98, 2.0):
        n = min(2 * beam, int(mult * beam) + int(rng.integers(0, 20)))
        for kind in ("distinct", "dup", "levels400", "levels8"):
            if kind == "distinct":
                sc = (rng.permutation(n).astype(np.float32) * -0.37 - 100.0).astype(np.float32)
            elif kind == "dup":
                sc = (-rng.random(n) * 300.0 - 5000.0).astype(np.float32)
                nd = max(2, int(0.03 * n))
                sc[rng.integers(0, n, nd)] = sc[rng.integers(0, n, nd)]
            else:
                levels = 400 if kind == "levels400" else 8
                sc = (-rng.integers(0, levels, n).astype(np.float32) * 0.5 - 2000.0).astype(np.float32)
            got = bm.prune_order(sc)
            info = bm.prune_info()
            want = oracle.sort_token_no_order(sc, beam)
            assert np.array_equal(got, want), (n, beam, kind, info)
            stats.append(info)
    assert sum(1 for s in stats if s > 0) >= len(stats) // 2, stats
    bm.close()


@pytest.mark.parametrize("beam", [2500, 4000])
def test_randomised_sizes_and_tie_densities(engine, oracle, beam):
    """A seeded sweep over frame sizes from just above the beam to five beams (both sort directions) and over tie
    densities from none to three score levels: whatever path the device takes -- closed form, wave-serial replay,
    sweep replay, downward closed form, extraction loop -- the order is the sequential loop's."""
    bm = _beam(engine, beam)
    rng = np.random.default_rng(1000 + beam)
    paths = {}
    for it in range(70):
        n = int(rng.integers(beam + 1, 5 * beam))
        kind = int(rng.integers(0, 6))
        if kind == 0:
            sc = (rng.permutation(n).astype(np.float32) * -0.37 - 100.0).astype(np.float32)
        elif kind <= 2:
            sc = (-rng.random(n) * float(rng.choice([3.0, 300.0, 30000.0])) - 5000.0).astype(np.float32)
            nd = max(2, int(float(rng.choice([0.001, 0.01, 0.1, 0.6])) * n))
            sc[rng.integers(0, n, nd)] = sc[rng.integers(0, n, nd)]
        else:
            levels = int(rng.choice([3, 17, 150, 2000, 30000]))
            sc = (-rng.integers(0, levels, n).astype(np.float32) * 0.5 - 2000.0).astype(np.float32)
        got = bm.prune_order(sc)
        info = bm.prune_info()
        want = oracle.sort_token_no_order(sc, beam)
        assert np.array_equal(got, want), (it, n, beam, kind, info)
        paths[1 if info > 0 else info] = paths.get(1 if info > 0 else info, 0) + 1
    assert paths.get(1, 0) >= 10, paths
    bm.close()


@pytest.mark.parametrize("beam", [800, 2500, 4000])
def test_whole_array_of_the_sort(engine, oracle, beam):
    """jamd_beam_prune_arrange(): tindex[] WHOLE after sort_token_no_order() -- residual heap and extracted part -- as the
    multipath frame's mid-frame sort needs it (csrc/beam_exact_mp.h): sweep replay + sift replay in both directions (beam
    800: the narrow layout; wider: the wide one), against the sequential code.  Sizes on both sides of 2 x beam, tie
    densities from none to heavy."""
    bm = _beam(engine, beam)
    rng = np.random.default_rng(7 * beam + 3)
    closed = 0
    for mult in (1.02, 1.3, 1.7, 1.97, 2.05, 2.4, 3.0, 4.2, 6.0):
        n = int(mult * beam) + int(rng.integers(0, 40))
        for dup in (0.0, 0.01, 0.2):
            sc = (-rng.random(n) * 300.0 - 5000.0).astype(np.float32)
            nd = int(dup * n)
            if nd:
                sc[rng.integers(0, n, nd)] = sc[rng.integers(0, n, nd)]
            order, arr = bm.prune_arrange(sc)
            info = bm.prune_info()
            worder, warr = oracle.sort_token_arrange(sc, beam)
            assert np.array_equal(order, worder), (n, beam, dup, info)
            assert np.array_equal(arr, warr), (n, beam, dup, info, int((arr != warr).sum()))
            closed += info > 0
    assert closed > 0
    # real frames of the C4 task and few score levels (everything ties)
    if beam == 4000:
        z = np.load(GOLD / "prune_frames_c4.npz")
        for name in sorted(k for k in z.files if k.startswith("f")):
            order, arr = bm.prune_arrange(z[name])
            worder, warr = oracle.sort_token_arrange(z[name], beam)
            assert np.array_equal(order, worder) and np.array_equal(arr, warr), name
    for levels in (3, 40):
        sc = (-rng.integers(0, levels, 3 * beam).astype(np.float32) - 100.0)
        order, arr = bm.prune_arrange(sc)
        worder, warr = oracle.sort_token_arrange(sc, beam)
        assert np.array_equal(order, worder) and np.array_equal(arr, warr), levels
    bm.close()


def test_frames_the_sweep_does_not_hold(engine, oracle):
    """The sweep's entries carry their positions as 16-bit routes (round 6): a frame of 2^16 tokens or more is not held and
    goes to the extraction loop (the heap in global memory), as is a frame on either side of that size with many events --
    the order stays the sequential loop's, and jamd_beam_prune_info() says which way a frame went."""
    beam = 4000
    bm = _beam(engine, beam)
    rng = np.random.default_rng(65536)
    for n, swept_wanted in ((65535, True), (65536, False), (70001, False)):
        sc = (-rng.random(n) * 300.0 - 5000.0).astype(np.float32)
        nd = n // 20
        sc[rng.integers(0, n, nd)] = sc[rng.integers(0, n, nd)]
        got = bm.prune_order(sc)
        info = bm.prune_info()
        want = oracle.sort_token_no_order(sc, beam)
        assert np.array_equal(got, want), (n, info)
        if not swept_wanted:
            assert info <= 0, (n, info)
    bm.close()
