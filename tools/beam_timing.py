#!/usr/bin/env python3
"""Quick device timing of the first-pass kernel on the synthetic 20k-word task
(development aid; bench.py --workload e2e is the reported number)."""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from julius_amd import lib, synth
import os
if os.environ.get("JAMD_LIB"):
    lib.LIB_PATH = Path(os.environ["JAMD_LIB"]).resolve()

nutt = int(sys.argv[1]) if len(sys.argv) > 1 else 64
beam = int(sys.argv[2]) if len(sys.argv) > 2 else 800
nword = int(sys.argv[3]) if len(sys.argv) > 3 else 20000
S, M, D = 3000, 16, 39
lex = synth.make_lexicon(nword=nword, nphone=40, S=S, seed=0)
model = synth.make_gmm(S=S, M=M, D=D, seed=0)
utts = [synth.make_lexicon_utterance(lex, model, nwords=12, seed=u)[0] for u in range(min(nutt, 16))]
utts = [utts[u % len(utts)] for u in range(nutt)]
off = np.zeros(nutt + 1, np.int32); off[1:] = np.cumsum([len(x) for x in utts])
frames = np.concatenate(utts)
eng = lib.Engine(0)
gm = lib.Gmm(eng, model); lx = lib.Lexicon(eng, lex)
bm = lib.Beam(eng, lx, beam, -1.0, max_utts=nutt, atoms_per_utt=1 << 17)
if len(sys.argv) > 4 and sys.argv[4] == "strict":
    bm.set_strict_order(True)
d_fr = torch.from_numpy(frames).cuda()
d_sc = torch.empty((len(frames), S), dtype=torch.float32, device="cuda")
st = torch.cuda.Stream()
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    gm.outprob_dev(d_fr.data_ptr(), len(frames), d_sc.data_ptr(), st.cuda_stream)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    bm.pass1_dev(d_sc.data_ptr(), S, off, st.cuda_stream)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    res = bm.results()
    print(f"utts {nutt} frames {len(frames)} beam {beam}: gmm {1e3*(t1-t0):.2f} ms, beam {1e3*(t2-t1):.2f} ms "
          f"-> {len(frames)/(t2-t1):.3e} frames/s beam, {1e6*(t2-t1)/max(len(x) for x in utts):.1f} us/frame/utt; "
          f"phases us {list(res[0].phase_us)} status {[r.status for r in res[:4]]} ties(node,we,cut) {[(r.ties_node, r.ties_wordend, r.ties_cut) for r in res[:2]]} maxtok {res[0].max_tokens} natom {res[0].natom}")
