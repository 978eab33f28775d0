"""The closed form of sort_token_upward()'s extraction loop that the exact-order first-pass kernel uses
(julius_amd/csrc/beam_exact.hip, DESIGN.md section 3 "K6x"), restated in plain Python and checked against the
oracle's sequential restatement of libjulius/src/beam.c:1342-1516 on tie-heavy random inputs.  CPU only.

Model.  After heapify, let B = the elements with score >= the k-th largest, each with its heap position ("virtual
position").  While the element taken from the tail of the array is not in B, an extraction is a hole running down
the path of larger children (left on ties): every heap position then emits its own element followed by the stable
merge of its children's streams, so the extraction order is (score descending, PRE-ORDER index of the position
ascending).  An extraction whose tail element IS in B ("event") re-inserts that element: it runs down the current max
path and stops at the first hole whose larger child is not greater -- from then on it counts as the element of that
position.  Who sits where at a given time follows from the order alone: a position holds the best remaining element
of its subtree that is not sitting further up, and because the subtrees along a root-to-leaf chain are nested, the
occupants of a chain come out of ONE scan over the remaining elements in order.

Which events matter.  An event changes the place of ONE element (the re-inserted one); an element whose score is unique in
B is ranked by its score wherever it sits.  The elements that can ever be re-inserted are those that start on a tail
position.  So the replay may stop behind the last turn at which a TIED element can sit on the tail position (`i_last`,
moved back when a tied element is re-inserted on a later tail position); earlier events of untied elements are still
replayed, because they decide who sits where when that turn comes."""
import numpy as np
import pytest

MAXL = 20


def prekey(p):
    L = p.bit_length() - 1
    return (((p - (1 << L)) << (MAXL - L)) << 5) | L


def insub(p, c):
    d = p.bit_length() - c.bit_length()
    return d >= 0 and (p >> d) == c


def heapify_upward(score):
    """First loop of sort_token_upward() (beam.c:1354-1367); returns the index array (1-based heap in a 0-based list)."""
    n = len(score)
    ti = list(range(n))
    for root in range(n // 2, 0, -1):
        s, parent = ti[root - 1], root
        while True:
            child = parent * 2
            if child > n:
                break
            if child < n and score[ti[child - 1]] < score[ti[child]]:
                child += 1
            if score[s] >= score[ti[child - 1]]:
                break
            ti[parent - 1] = ti[child - 1]
            parent = child
        ti[parent - 1] = s
    return ti


def closed_form_order(score, k, skip=True, stats=None):
    """tindex[n-k .. n-1] after sort_token_upward(k, n), without running its second loop.  skip: stop the replay of
    the events behind the last turn that can change the order."""
    n = len(score)
    heap = heapify_upward(score)
    vk = sorted(score, reverse=True)[k - 1]
    vpos = {idx: pos for pos, idx in enumerate(heap, 1) if score[idx] >= vk}
    key = lambda x: (-score[x], prekey(vpos[x]))
    order = sorted(vpos, key=key)
    mult = {}
    for e in vpos:
        mult[score[e]] = mult.get(score[e], 0) + 1
    i_last = max([n - p + 1 for e, p in vpos.items() if p >= n - k + 1 and mult[score[e]] > 1], default=0)
    out = []
    for i in range(1, k + 1):
        out.append(order[i - 1])
        if skip and i > i_last:
            continue
        q = n - i + 1
        if not any(vpos[e] == q for e in order[i:]):
            continue
        # occupants of the chain root -> q: one scan over the remaining elements in order
        Lq, d, occq = q.bit_length() - 1, 0, None
        for x in order[i - 1:]:
            if insub(vpos[x], q >> (Lq - d)):
                if d == Lq:
                    occq = x
                    break
                d += 1
        if occq is None:
            continue                       # the element moved up (or out) before its tail turn
        s, hs, hole = occq, n - i, 1        # event: s runs down the max path of the heap of size n - i
        for x in order[i:]:
            if x == s:
                continue
            if 2 * hole > hs:
                break
            v = vpos[x]
            if insub(v, hole) and v != hole:
                c = v >> (v.bit_length() - hole.bit_length() - 1)
                if c > hs:
                    continue
                if score[s] >= score[x]:
                    break
                hole = c
        vpos[s] = hole
        if stats is not None:
            stats[0] += 1
        if hole >= n - k + 1 and any(score[x] == score[s] for x in order[i:] if x != s):
            i_last = max(i_last, n - hole + 1)     # a tied element on a later tail position
        order[i:] = sorted(order[i:], key=key)
    return out[::-1]


@pytest.mark.parametrize("seed", range(6))
def test_closed_form_equals_sequential_heap(oracle, seed):
    rng = np.random.default_rng(seed)
    checked = events = 0
    for _ in range(120):
        n = int(rng.integers(5, 160))
        k = int(rng.integers(1, max(2, (n - 1) // 2)))
        if not k < n - k:
            continue
        levels = int(rng.choice([2, 3, 5, 10, 30, 1000]))
        score = [float(x) for x in -rng.integers(0, levels, n) * 0.5 - 100.0]
        want = list(oracle.sort_token_no_order(np.array(score, np.float32), k))
        full, cut = [0], [0]
        assert closed_form_order(score, k, skip=False, stats=full) == want, (n, k, levels)
        assert closed_form_order(score, k, skip=True, stats=cut) == want, (n, k, levels)
        assert cut[0] <= full[0]
        checked += 1
        events += full[0]
    assert checked > 60 and events > 0


def heapify_overlapped(score, split=3):
    """The schedule of heapify_overlapped() in beam_exact.hip, lane by lane: the sift-downs of ALL levels of a subtree
    run together, the sift of depth L starting one step after the sifts of depth L + 1; within a step every active
    sift first READS the two children of its hole (all reads see the state left by the previous step), then all
    WRITE.  Subtrees rooted at depth `split` first (independent of each other), then depths split-1 .. 0."""
    n = len(score)
    H = [None] + list(range(n))                       # 1-based heap of token ids
    top = n // 2
    if top < 1:
        return H[1:]
    val = lambda idx: score[idx]

    def run(roots, ldeep):
        # roots: list of (position, depth); every sift starts at step ldeep - depth
        sifts = [dict(parent=p, s=H[p], t0=ldeep - d, live=True) for p, d in roots]
        g = 0
        while any(x["live"] for x in sifts):
            reads = []
            for x in sifts:                            # reads of this step
                if not x["live"] or g < x["t0"]:
                    reads.append(None)
                    continue
                c = 2 * x["parent"]
                reads.append((H[c] if c <= n else None, H[c + 1] if c + 1 <= n else None))
            for x, rd in zip(sifts, reads):            # decisions and writes
                if rd is None:
                    continue
                left, right = rd
                if left is None:
                    H[x["parent"]] = x["s"]; x["live"] = False
                    continue
                child, cid = 2 * x["parent"], left
                if right is not None and val(left) < val(right):
                    child, cid = child + 1, right
                if val(x["s"]) >= val(cid):
                    H[x["parent"]] = x["s"]; x["live"] = False
                else:
                    H[x["parent"]] = cid; x["parent"] = child
            g += 1

    ltop = top.bit_length() - 1
    if ltop >= split:
        roots = [(p, p.bit_length() - 1) for p in range(1 << split, top + 1)]
        run(roots, ltop)                               # (the device runs four subtrees per wave; they do not interact)
    lt = min(ltop, split - 1)
    run([(p, p.bit_length() - 1) for p in range(1, min(top, (2 << lt) - 1) + 1)], lt)
    return H[1:]


@pytest.mark.parametrize("seed", range(4))
def test_overlapped_heapify_equals_sequential(seed):
    rng = np.random.default_rng(100 + seed)
    for _ in range(150):
        n = int(rng.integers(1, 700))
        levels = int(rng.choice([2, 3, 10, 1000, 10 ** 6]))
        score = [float(x) for x in -rng.integers(0, levels, n) * 0.25 - 50.0]
        for split in (1, 3, 6):
            assert heapify_overlapped(score, split) == heapify_upward(score), (n, levels, split)


@pytest.mark.parametrize("seed", range(6))
def test_whole_array_model_equals_sequential_heap(seed):
    """The model behind exact_prune<FULL> (csrc/beam_exact.hip, beam_sweep.h; the multipath frame's mid-frame sort needs
    tindex[] WHOLE): the extracted part = the sweep replay's extraction order, the residual heap = every non-event turn's
    tail element sifted down from where the hole left the extracted region -- in pure Python (tools/prune_lab2.py) against
    the sequential loop of sort_token_upward() / _downward() (beam.c:1342-1480), both directions, from distinct scores to
    three score levels."""
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tools"))
    from prune_lab2 import check_full
    rng = np.random.default_rng(900 + seed)
    for _ in range(40):
        k = int(rng.integers(1, 90))
        n = int(rng.integers(k + 1, 5 * k + 2))
        nlev = int(rng.choice([3, 8, 30, 1000, 10 ** 6]))
        sc = rng.integers(0, nlev, n).astype(np.float32)
        ok, rounds, nev = check_full(sc, k)
        assert ok, (seed, n, k, nlev, rounds, nev)
