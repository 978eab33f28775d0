#!/usr/bin/env python3
"""Lab, part 2: the PARALLEL form of the rank pruning step's event replay, as the device code will run it.

Model: every top element has a chain of incarnations (virtual heap positions): its original position, and one more per
event (the element sat on the tail position of turn i when that turn came and was re-inserted at h).  One round =
one sweep over ALL incarnations at once:
  * entries sorted by (score desc, pre-order of the position): the last incarnation of an element is REAL, the earlier
    ones are PROBES (they ride along to learn when they would have left their leaf, but delay nobody);
  * T_0 = 1 + number of real entries before; level by level (depth d -> d+1) the entries that lie deeper than d are
    partitioned stably by their next path bit (all left-goers first: a wavelet matrix, the groups stay contiguous) and
    take T_{d+1} = T_d of the entry just before them (probes are looked through; a landed entry born at turn b is looked
    through when the value behind it is < b);
  * an entry on a tail position (turn i) is an event iff T at its own depth >= i; its landing = walk over the elements
    that move at turn i (depth 1, 2, ...) while they are strictly better.
Rounds repeat until no belief changes.  Checked here against the sequential loop."""
import sys, numpy as np
sys.path.insert(0, __import__("os").path.dirname(__file__))
from prune_lab import load, heapify_up, extract_up, depth, prekey

def heapify(sc, up=True):
    n = len(sc); H = [0] + list(range(n))
    better = (lambda a, b: a > b) if up else (lambda a, b: a < b)
    for root in range(n // 2, 0, -1):
        s = H[root]; parent = root
        while 2 * parent <= n:
            child = 2 * parent
            if child < n and better(sc[H[child + 1]], sc[H[child]]): child += 1
            if not better(sc[H[child]], sc[s]): break
            H[parent] = H[child]; parent = child
        H[parent] = s
    return H

def extract(sc, H, cnt, up=True):
    n = len(H) - 1; H = list(H); out = []
    better = (lambda a, b: a > b) if up else (lambda a, b: a < b)
    m = n
    for i in range(1, cnt + 1):
        s = H[m]; out.append(H[1]); H[m] = H[1]; m -= 1; parent = 1
        while 2 * parent <= m:
            child = 2 * parent
            if child < m and better(sc[H[child + 1]], sc[H[child]]): child += 1
            if not better(sc[H[child]], sc[s]): break
            H[parent] = H[child]; parent = child
        H[parent] = s
    return out, H

USE_LIMIT = True

class Sweep:
    def __init__(self, gkey, n, cnt):
        """gkey[e]: order key of element e, SMALLER = extracted earlier (ties equal); n = heap size, cnt = turns"""
        self.g = gkey; self.n = n; self.cnt = cnt

    def run(self, vpos0, limit, max_rounds=40, verbose=False, tied=None):
        g, n = self.g, self.n
        m = len(g)
        chains = [[] for _ in range(m)]          # per element: [(q, turn, h), ...]
        for rnd in range(1, max_rounds + 1):
            # ---- entries
            ex, evp, eprobe, ebirth, eref = [], [], [], [], []
            for e in range(m):
                vp = int(vpos0[e]); b = 0
                for ci, (q, turn, h) in enumerate(chains[e]):
                    ex.append(e); evp.append(vp); eprobe.append(True); ebirth.append(b); eref.append(ci)
                    vp = h; b = turn
                ex.append(e); evp.append(vp); eprobe.append(False); ebirth.append(b); eref.append(len(chains[e]))
            ex = np.array(ex); evp = np.array(evp, np.int64); eprobe = np.array(eprobe); ebirth = np.array(ebirth)
            M = len(ex)
            pk = np.array([prekey(int(p)) for p in evp], np.int64)
            order = np.lexsort((np.arange(M), pk, g[ex]))
            dep = np.array([depth(int(p)) for p in evp])
            # Tie groups holding a landed (born > 0) real entry: the members come out by pre-order among those PRESENT, a
            # landed one is present from the turn after its birth.  Re-order such groups by their true turns; a probe
            # stays in front of the real member that follows it in pre-order.
            order = list(order); i0 = 0; turn = 0
            fixed = []
            while i0 < M:
                i1 = i0
                while i1 < M and g[ex[order[i1]]] == g[ex[order[i0]]]: i1 += 1
                grp = order[i0:i1]
                reals = [x for x in grp if not eprobe[x]]
                if any(ebirth[x] > turn for x in reals):
                    attach = {}; nxt = None
                    for x in reversed(grp):
                        if eprobe[x]: attach.setdefault(nxt, []).insert(0, x)
                        else: nxt = x
                    pend = list(reals); sched = []; t = turn
                    while pend:
                        t += 1
                        c = next((x for x in pend if ebirth[x] < t), None)
                        if c is None: c = min(pend, key=lambda x: ebirth[x]); t = ebirth[c] + 1   # (cannot happen in a consistent state)
                        pend.remove(c); sched.append(c)
                    newg = []
                    for x in sched: newg += attach.get(x, []) + [x]
                    newg += attach.get(None, [])
                    grp = newg
                fixed += grp; turn += len(reals); i0 = i1
            order = np.array(fixed)
            real_before = np.cumsum(~eprobe[order]) - (~eprobe[order])
            A = order; T = real_before + 1
            TD = np.zeros(M, np.int64); moves = {}
            posend = np.zeros(M, np.int64)       # where an entry that is still in the heap after cnt turns sits then
            d = 0
            while len(A):
                for x, t in zip(A, T):
                    if t > self.cnt: posend[x] = int(evp[x]) >> (dep[x] - d)
                # moves of real entries at this level
                if d >= 1:
                    for x, t in zip(A, T):
                        if not eprobe[x]: moves[(int(t), d)] = int(x)
                own = dep[A] == d
                TD[A[own]] = T[own]
                L = len(A)
                newA0, newT0, newA1, newT1 = [], [], [], []
                for idx in range(L):
                    x = A[idx]
                    if own[idx]: continue
                    # resolve the value of the entry before idx
                    j = idx - 1
                    while j >= 0 and (eprobe[A[j]] or ebirth[A[j]] > 0): j -= 1
                    v = T[j] if j >= 0 else 0
                    for s in range(j + 1, idx):
                        y = A[s]
                        if eprobe[y]: continue
                        if v < ebirth[y]: continue            # moved before y was born
                        v = max(T[s], ebirth[y])
                    bit = (int(evp[x]) >> (dep[x] - d - 1)) & 1
                    (newA1 if bit else newA0).append(x); (newT1 if bit else newT0).append(v)
                A = np.array(newA0 + newA1, np.int64); T = np.array(newT0 + newT1, np.int64); d += 1
            # ---- evaluation: new chains
            changed = False; nev = 0
            first = {}      # (elem, chain index) -> entry
            for x in range(M): first[(int(ex[x]), int(eref[x]))] = x
            for e in range(m):
                old = chains[e]; new = []
                vp = int(vpos0[e]); ci = 0
                while True:
                    x = first.get((e, ci))
                    if x is None or int(evp[x]) != vp: break           # this incarnation was not in the sweep
                    turn = n - vp + 1
                    if not (vp >= n - self.cnt + 1 and turn <= limit): break
                    if TD[x] < turn: break                              # moved up before its turn
                    t = 0
                    while True:
                        y = moves.get((turn, t + 1))
                        if y is None or int(ex[y]) == e or not (g[ex[y]] < g[e]): break
                        t += 1
                    if t == 0: h = 1
                    else:
                        y = moves[(turn, t)]; h = int(evp[y]) >> (dep[y] - t)
                    new.append((vp, turn, h)); vp = h; ci += 1
                if new != old: changed = True
                if tied is not None and tied[e]:
                    for (q, turn, h) in new:
                        if h >= n - self.cnt + 1 and n - h + 1 > limit: limit = n - h + 1; changed = True
                chains[e] = new; nev += len(new)
            if verbose: print("  round", rnd, "entries", M, "events", nev, "changed", changed)
            if not changed:
                self.last = dict(ex=ex, evp=evp, eprobe=eprobe, dep=dep, moves=moves, TD=TD, posend=posend)
                return chains, rnd
        return None, max_rounds

def final_order(gkey, vpos0, chains):
    m = len(gkey)
    vp = np.array([c[-1][2] if c else int(vpos0[e]) for e, c in enumerate(chains)], np.int64)
    birth = np.array([c[-1][1] if c else 0 for c in chains], np.int64)
    pk = np.array([prekey(int(p)) for p in vp], np.int64)
    order = list(np.lexsort((pk, gkey)))
    out = []; i0 = 0; turn = 0
    while i0 < m:
        i1 = i0
        while i1 < m and gkey[order[i1]] == gkey[order[i0]]: i1 += 1
        grp = order[i0:i1]
        if any(birth[x] > turn for x in grp):
            pend = list(grp); sched = []; t = turn
            while pend:
                t += 1
                c = next((x for x in pend if birth[x] < t), None)
                if c is None: c = min(pend, key=lambda x: birth[x]); t = birth[c] + 1
                pend.remove(c); sched.append(c)
            grp = sched
        out += grp; turn += len(grp); i0 = i1
    return np.array(out)

def check_up(sc, k, verbose=False):
    n = len(sc)
    H = heapify(sc, True)
    out, _ = extract(sc, H, k, True)
    vk = np.sort(sc)[::-1][k - 1]
    ids = np.nonzero(sc >= vk)[0]
    pos = np.zeros(n, np.int64); pos[np.array(H[1:])] = np.arange(1, n + 1)
    score = sc[ids]; vpos0 = pos[ids]
    # order key: dense rank of the score, descending
    u = np.unique(score)[::-1]; gk = np.searchsorted(-u, -score)
    sw = Sweep(gk, n, k)
    cnts = np.bincount(gk); tied = cnts[gk] > 1
    if USE_LIMIT:
        tt = [n - int(p) + 1 for p, t in zip(vpos0, tied) if t and p >= n - k + 1]
        limit = max(tt) if tt else 0
        chains, rounds = sw.run(vpos0, limit, verbose=verbose, tied=tied)
    else:
        chains, rounds = sw.run(vpos0, k, verbose=verbose)
    if chains is None: return False, rounds, 0
    o = final_order(gk, vpos0, chains)[:k]
    return [int(ids[j]) for j in o] == out, rounds, sum(len(c) for c in chains)

def check_down(sc, k, verbose=False):
    """sort_token_downward(): min-heap, R = n - k extractions; the survivors are the residual heap H[1..k]."""
    n = len(sc); R = n - k
    H = heapify(sc, False)
    out, Hf = extract(sc, H, R, False)
    want = Hf[1:k + 1]
    vR = np.sort(sc)[R - 1]
    ids = np.nonzero(sc <= vR)[0]
    pos = np.zeros(n, np.int64); pos[np.array(H[1:])] = np.arange(1, n + 1)
    score = sc[ids]; vpos0 = pos[ids]
    u = np.unique(score); gk = np.searchsorted(u, score)
    sw = Sweep(gk, n, R)
    chains, rounds = sw.run(vpos0, R, verbose=verbose)
    if chains is None: return False, rounds, 0
    L = sw.last
    ex, evp, eprobe, dep, moves = L["ex"], L["evp"], L["eprobe"], L["dep"], L["moves"]
    event_turn = set(t for c in chains for (q, t, h) in c)
    P = list(H)
    for i in range(1, R + 1):
        if i in event_turn: continue
        q = n - i + 1; m = n - i
        f = 1; d = 1
        while (i, d) in moves:
            x = moves[(i, d)]; f = int(evp[x]) >> (dep[x] - d); d += 1
        s = P[q]; p = f
        while 2 * p <= m:
            c = 2 * p
            if c < m and sc[P[c]] > sc[P[c + 1]]: c += 1
            if sc[s] <= sc[P[c]]: break
            P[p] = P[c]; p = c
        P[p] = s
    res = P[1:k + 1]
    for x in range(len(ex)):
        if not eprobe[x] and L["posend"][x] > 0:
            assert L["posend"][x] <= k
            res[int(L["posend"][x]) - 1] = int(ids[ex[x]])
    return res == want, rounds, sum(len(c) for c in chains)

def check_full(sc, k):
    """sort_token_no_order() with the WHOLE array out -- what the multipath frame's mid-frame sort needs (csrc/beam_exact_mp.h,
    exact_prune<FULL>): the extracted part from the sweep's extraction order, the residual heap from the sift replay
    below the extracted region, either direction.  Returns (equal to the sequential loop, rounds, events)."""
    n = len(sc); up = k < n - k; cnt = k if up else n - k
    H = heapify(sc, up)
    out, Hf = extract(sc, H, cnt, up)
    if up:
        v = np.sort(sc)[::-1][cnt - 1]; ids = np.nonzero(sc >= v)[0]
        u = np.unique(sc[ids])[::-1]; gk = np.searchsorted(-u, -sc[ids])
    else:
        v = np.sort(sc)[cnt - 1]; ids = np.nonzero(sc <= v)[0]
        u = np.unique(sc[ids]); gk = np.searchsorted(u, sc[ids])
    pos = np.zeros(n, np.int64); pos[np.array(H[1:])] = np.arange(1, n + 1)
    vpos0 = pos[ids]
    sw = Sweep(gk, n, cnt)
    chains, rounds = sw.run(vpos0, cnt)
    if chains is None: return False, rounds, 0
    L = sw.last
    ex, evp, eprobe, dep, moves = L["ex"], L["evp"], L["eprobe"], L["dep"], L["moves"]
    event_turn = set(t for c in chains for (q, t, h) in c)
    better = (lambda a, b: a > b) if up else (lambda a, b: a < b)
    P = list(H)
    for i in range(1, cnt + 1):
        if i in event_turn: continue
        q = n - i + 1; m = n - i
        f = 1; d = 1
        while (i, d) in moves:                                    # where the hole leaves the extracted region
            x = moves[(i, d)]; f = int(evp[x]) >> (dep[x] - d); d += 1
        s = P[q]; p = f
        while 2 * p <= m:
            c = 2 * p
            if c < m and better(sc[P[c + 1]], sc[P[c]]): c += 1
            if not better(sc[P[c]], sc[s]): break
            P[p] = P[c]; p = c
        P[p] = s
    R = n - cnt
    res = P[1:R + 1]
    for x in range(len(ex)):
        if not eprobe[x] and L["posend"][x] > 0:
            if L["posend"][x] > R: return False, rounds, 0
            res[int(L["posend"][x]) - 1] = int(ids[ex[x]])
    order = [int(ids[j]) for j in final_order(gk, vpos0, chains)[:cnt]]      # i-th extracted -> array position n - i + 1
    full = res + order[::-1]
    return full == list(Hf[1:]), rounds, sum(len(c) for c in chains)

if __name__ == "__main__":
    if sys.argv[1] == "realdown":
        recs = load(sys.argv[2]); step = int(sys.argv[3]) if len(sys.argv) > 3 else 10
        dn = [(k, sc) for k, sc in recs if not k < len(sc) - k and len(sc) > k]
        for k, sc in dn[::step]:
            ok, rounds, nev = check_down(sc, k)
            print(f"n={len(sc)} k={k} exact={ok} rounds={rounds} events={nev}", flush=True)
    elif sys.argv[1] == "fuzzdown":
        rng = np.random.default_rng(int(sys.argv[2])); N = int(sys.argv[3]); bad = 0; rmax = 0
        for it in range(N):
            k = int(rng.integers(1, 200)); n = int(rng.integers(k + 1, 2 * k + 1))
            nlev = int(rng.choice([3, 8, 30, 1000]))
            sc = rng.integers(0, nlev, n).astype(np.float32)
            ok, rounds, nev = check_down(sc, k)
            rmax = max(rmax, rounds)
            if not ok:
                bad += 1; print("MISMATCH", it, n, k, nlev, rounds, nev); np.save(f"/tmp/lab/badd_{it}.npy", np.append(sc, k))
        print("fuzz down done: bad", bad, "of", N, "max rounds", rmax)
    elif sys.argv[1] == "real":
        recs = load(sys.argv[2]); step = int(sys.argv[3]) if len(sys.argv) > 3 else 40
        for k, sc in recs[::step]:
            n = len(sc)
            if not k < n - k: continue
            ok, rounds, nev = check_up(sc, k)
            print(f"n={n} k={k} exact={ok} rounds={rounds} events={nev}", flush=True)
    else:
        rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
        bad = 0; N = int(sys.argv[3]) if len(sys.argv) > 3 else 300; rmax = 0
        for it in range(N):
            n = int(rng.integers(8, 400)); k = int(rng.integers(1, max(2, (n - 1) // 2)))
            if not k < n - k: continue
            nlev = int(rng.choice([3, 8, 30, 1000]))
            sc = rng.integers(0, nlev, n).astype(np.float32)
            ok, rounds, nev = check_up(sc, k)
            rmax = max(rmax, rounds)
            if not ok:
                bad += 1; print("MISMATCH", it, n, k, nlev, rounds, nev); np.save(f"/tmp/lab/bad_{it}.npy", sc)
        print("fuzz done: bad", bad, "of", N, "max rounds", rmax)
