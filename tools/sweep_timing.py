#!/usr/bin/env python3
"""Device time of the sweep replay (csrc/beam_sweep.h) on the real C4 frames of tests/golden/prune_frames_c4.npz."""
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
bm = lib.Beam(eng, lx, int(z["beam"]), -1.0, max_utts=1)
rows = []
for name in sorted(k for k in z.files if k.startswith("f")):
    sc = z[name]
    bm.prune_order(sc)
    t0 = time.perf_counter(); bm.prune_order(sc); host = time.perf_counter() - t0
    r = bm.prune_info()
    rows.append({"frame": name, "n": len(sc), "rounds": r, "sweep_us": bm.last_sweep_us, "events": bm.last_sweep_events, "host_call_ms": round(host * 1e3, 2)})
print(json.dumps(rows))
