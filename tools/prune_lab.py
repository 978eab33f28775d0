#!/usr/bin/env python3
"""Lab for the rank pruning step (sort_token_upward, beam.c:1342-1385) on real inputs (tools/dump_prune_inputs.py).
Sequential ground truth with event bookkeeping, and the static closed-form model with landed incarnations."""
import sys, numpy as np

def load(path):
    raw = np.fromfile(path, dtype=np.int32); i = 0; out = []
    while i < len(raw):
        n, k = int(raw[i]), int(raw[i + 1])
        out.append((k, raw[i + 2:i + 2 + n].view(np.float32).copy())); i += 2 + n
    return out

def heapify_up(sc):
    """first loop of sort_token_upward: H[1..n] token ids (max-heap by score, reference's comparisons)"""
    n = len(sc); H = [0] + list(range(n))
    for root in range(n // 2, 0, -1):
        s = H[root]; parent = root
        while 2 * parent <= n:
            child = 2 * parent
            if child < n and sc[H[child]] < sc[H[child + 1]]: child += 1
            if sc[s] >= sc[H[child]]: break
            H[parent] = H[child]; parent = child
        H[parent] = s
    return H

def extract_up(sc, H, k, E):
    """second loop; returns extraction order (ids) and the events [(turn, id, q, h)] for tail elements in set E"""
    n = len(H) - 1; H = list(H); out = []; ev = []
    m = n
    for i in range(1, k + 1):
        s = H[m]; out.append(H[1]); q = m; H[m] = H[1]; m -= 1; parent = 1
        while 2 * parent <= m:
            child = 2 * parent
            if child < m and sc[H[child]] < sc[H[child + 1]]: child += 1
            if sc[s] >= sc[H[child]]: break
            H[parent] = H[child]; parent = child
        H[parent] = s
        if s in E: ev.append((i, s, q, parent))
    return out, ev

def depth(p): return int(p).bit_length() - 1
def prekey(p, maxl=20):
    L = depth(p); return (((p - (1 << L)) << (maxl - L)) << 5) | L

if __name__ == "__main__":
    recs = load(sys.argv[1] if len(sys.argv) > 1 else "/tmp/lab/prune_inputs.bin")
    import collections
    for fi, (k, sc) in enumerate(recs[::max(1, len(recs) // 12)]):
        n = len(sc)
        if not (k < n - k): continue
        H = heapify_up(sc)
        vk = np.sort(sc)[::-1][k - 1]
        E = set(int(t) for t in np.nonzero(sc >= vk)[0])
        out, ev = extract_up(sc, H, k, E)
        pos = {H[p]: p for p in range(1, n + 1)}
        cand = [t for t in E if pos[t] >= n - k + 1]
        ties = len(E) - len(set(sc[list(E)]))
        print(f"frame n={n} k={k} n/k={n / k:.2f} |E*|={len(E)} tied_extra={ties} candidates={len(cand)} events={len(ev)} "
              f"re-events={len(ev) - len(set(e[1] for e in ev))}")


# ------------------------------------------------------------------ the static model
def model_sweep(score, vpos, maxl=20):
    """Elements e (arrays score[e], vpos[e] = virtual heap position) in the event-free closed form: rank order =
    (score descending, pre-order of vpos), T_d(e) = the turn at which e leaves the depth-d ancestor of vpos[e]
    (T_0 = rank + 1; T_{d+1}(e) = T_d(predecessor of e among the elements below the same depth-d ancestor)).
    Returns rank[e], TD[e] = T at e's own depth, moves[(turn, depth)] = element that moves up from that depth at that turn."""
    m = len(score)
    pk = np.array([prekey(int(p), maxl) for p in vpos], dtype=np.int64)
    order = np.lexsort((pk, -score.astype(np.float64)))
    rank = np.empty(m, np.int64); rank[order] = np.arange(m)
    dep = np.array([depth(int(p)) for p in vpos])
    A = order.copy(); T = np.arange(1, m + 1)
    TD = np.zeros(m, np.int64); moves = {}
    d = 0
    while len(A):
        own = dep[A] == d
        TD[A[own]] = T[own]
        if d >= 1:
            for e, t in zip(A, T): moves[(int(t), d)] = int(e)
        desc = np.nonzero(~own)[0]
        if not len(desc): break
        assert desc[0] > 0
        child = (vpos[A[desc]] >> (dep[A[desc]] - d - 1))
        # the predecessor sits just before e in A and belongs to the same depth-d group
        anc_prev = vpos[A[desc - 1]] >> (dep[A[desc - 1]] - d)
        assert np.all(anc_prev == (child >> 1)), "first of a group descends"
        newT = T[desc - 1]
        o = np.argsort(child, kind="stable")
        A = A[desc][o]; T = newT[o]; d += 1
    return rank, TD, moves

def evaluate(score, vpos, n, k):
    """status and landing of every element sitting on a tail position, in the model where everything is real"""
    rank, TD, moves = model_sweep(score, vpos)
    res = {}
    for e in np.nonzero(vpos >= n - k + 1)[0]:
        q = int(vpos[e]); i = n - q + 1
        if TD[e] >= i:
            t = 0
            while True:
                x = moves.get((i, t + 1))
                if x is None or x == e or not (score[x] > score[e]): break
                t += 1
            if t == 0: h = 1
            else:
                x = moves[(i, t)]; h = int(vpos[x]) >> (depth(int(vpos[x])) - t)
            res[int(e)] = (i, h)
    return res, rank

def lab_round1(recs, every):
    for fi, (k, sc) in enumerate(recs[::every]):
        n = len(sc)
        if not (k < n - k): continue
        H = heapify_up(sc)
        vk = np.sort(sc)[::-1][k - 1]
        ids = np.nonzero(sc >= vk)[0]
        E = set(int(t) for t in ids)
        out, ev = extract_up(sc, H, k, E)
        pos = {H[p]: p for p in range(1, n + 1)}
        score = sc[ids]; vpos = np.array([pos[int(t)] for t in ids], dtype=np.int64)
        loc = {int(t): j for j, t in enumerate(ids)}
        truth = {}
        for (i, s, q, h) in ev: truth.setdefault(loc[s], []).append((i, q, h))
        # Jacobi rounds: vpos of an event := its landing
        cur = vpos.copy(); rounds = 0; hist = []
        fixed = {}         # element -> list of (turn, q, h) decided so far
        for rounds in range(1, 30):
            res, rank = evaluate(score, cur, n, k)
            # agreement with the truth of the events found this round
            good = sum(1 for e, (i, h) in res.items() if any(t[0] == i and t[2] == h for t in truth.get(e, [])))
            hist.append((len(res), good))
            if not res: break
            for e, (i, h) in res.items(): cur[e] = h
        # final order
        pk = np.array([prekey(int(p)) for p in cur], dtype=np.int64)
        order = np.lexsort((pk, -score.astype(np.float64)))[:k]
        ok = [int(ids[j]) for j in order] == out
        print(f"n={n} events={len(ev)} rounds={rounds} per round (found, agree with truth)={hist} final order exact={ok}")

if __name__ == "__main__" and len(sys.argv) > 2 and sys.argv[2] == "jacobi":
    lab_round1(load(sys.argv[1]), 25)
