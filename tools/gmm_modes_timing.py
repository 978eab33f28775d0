#!/usr/bin/env python3
"""Device timing of the pruned / tied-mixture scoring kernels (K1s, K2) next to K1
(development aid; numbers quoted in DESIGN.md)."""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from julius_amd import lib, synth

T = int(sys.argv[1]) if len(sys.argv) > 1 else 16000
eng = lib.Engine(0)
def run(name, model, gprune, n):
    gm = lib.Gmm(eng, model, gprune, n)
    fr = synth.make_frames(model, T=T, seed=3) if "centre" in model else np.random.default_rng(3).normal(0, 1.5, (T, model["mean"].shape[1])).astype(np.float32)
    d_fr = torch.from_numpy(fr).cuda(); d_out = torch.empty((T, gm.S), dtype=torch.float32, device="cuda")
    st = torch.cuda.Stream()
    for _ in range(2): gm.outprob_dev(d_fr.data_ptr(), T, d_out.data_ptr(), st.cuda_stream)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): gm.outprob_dev(d_fr.data_ptr(), T, d_out.data_ptr(), st.cuda_stream)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f"{name}: S={gm.S} T={T}: {dt*1e3:.2f} ms -> {T/dt:.3e} frames/s, {T*gm.S/dt:.3e} frame*states/s [{gm.last_kernel()}]")
m = synth.make_gmm(S=3000, M=16, D=39, seed=0)
run("plain none", m, lib.GPRUNE_NONE, 0)
run("plain safe N=2", m, lib.GPRUNE_SAFE, 2)
run("plain safe N=8", m, lib.GPRUNE_SAFE, 8)
tm = synth.make_tied_gmm(S=3000, nbook=129, K=64, D=39, seed=1)
run("tied-mixture 129 books x 64, none", tm, lib.GPRUNE_NONE, 0)
run("tied-mixture 129 books x 64, safe N=2", tm, lib.GPRUNE_SAFE, 2)
