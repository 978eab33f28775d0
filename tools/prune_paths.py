#!/usr/bin/env python3
"""Dev: how the pruning steps of one C4 utterance are resolved (jamd_beam_prune_stats), raw counters."""
import sys, tempfile, json
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench
from julius_amd import lib, synth, lexblob
flat = "--flat" in sys.argv
eng = lib.Engine(0)
wd = Path(tempfile.mkdtemp())
dnn = synth.make_dnn(seed=0) if flat else synth.make_decodable_dnn(seed=0)
task, jargs, prefix = bench.build_reference_task(wd, 20000, 4000, dnn)
lx = lib.Lexicon.from_file(eng, str(prefix) + ".lex")
net = lib.Dnn.from_dnnconf(eng, task["dnnconf"])
fr = (np.random.default_rng(1000).normal(0, 1, (1000, 528)).astype(np.float32) if flat
      else synth.make_dnn_utterance(task, dnn, nwords=30, seed=0)[0])
bm = lib.Beam(eng, lx, 4000, -1.0, max_utts=1, atoms_per_utt=1 << 18)
d_fr = lib.DevBuf(eng, fr.nbytes).upload(fr)
d_sc = lib.DevBuf(eng, 4 * len(fr) * net.S)
net.outprob_dev(d_fr.ptr, len(fr), d_sc.ptr)
bm.pass1_dev(d_sc.ptr, net.S, np.array([0, len(fr)], np.int32))
bm.results()
st = bm.prune_stats(0)
print(json.dumps({"frames": len(fr), "stats": st, "reasons_hex": hex(st[2] & 0xffffffff)}))
