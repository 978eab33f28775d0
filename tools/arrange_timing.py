#!/usr/bin/env python3
"""development: device time of the whole-array form of the rank pruning step (exact_prune<FULL>: sweep replay + sift replay)
on the real C4 frames of tests/golden/prune_frames_c4.npz and on C3-sized synthetic frames (beam 800)."""
import json, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from julius_amd import lib
from beamutil import load_beam_golden
eng = lib.Engine(0)
g = load_beam_golden("beam_rank.npz")
lx = lib.Lexicon(eng, g["lex"])
z = np.load(ROOT / "tests/golden/prune_frames_c4.npz")
rows = []
for beam, frames in ((int(z["beam"]), [(k, z[k]) for k in sorted(z.files) if k.startswith("f")][:6]),
                     (800, [(f"s{i}", (-np.random.default_rng(i).random(5000) * 300 - 5000).astype(np.float32)) for i in range(3)])):
    bm = lib.Beam(eng, lx, beam, -1.0, max_utts=1)
    for name, sc in frames:
        bm.prune_arrange(sc)
        t0 = time.perf_counter(); bm.prune_arrange(sc); host = time.perf_counter() - t0
        r = bm.prune_info()
        t0 = time.perf_counter(); bm.prune_order(sc); host2 = time.perf_counter() - t0
        rows.append({"beam": beam, "frame": name, "n": len(sc), "rounds": r, "sweep_us": bm.last_sweep_us, "events": bm.last_sweep_events,
                     "arrange_call_ms": round(host * 1e3, 2), "order_call_ms": round(host2 * 1e3, 2)})
    bm.close()
print(json.dumps(rows))
