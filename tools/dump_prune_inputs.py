#!/usr/bin/env python3
"""Test tooling: real inputs of the rank pruning step (sort_token_no_order, beam.c:1492) taken from the CPU oracle's first
pass over the C4 task that decodes (or, --flat, round 3's flat-score stream): one record (n, beam, scores in creation
order) per frame, written by oracle/jamd_oracle_beam.c under JAMD_ORACLE_DUMP_SORT.  Used by tools/prune_lab.py."""
import argparse, os, sys, tempfile, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

ap = argparse.ArgumentParser()
ap.add_argument("--out", default="/tmp/lab/prune_inputs.bin")
ap.add_argument("--flat", action="store_true")
ap.add_argument("--gmm", action="store_true", help="the C3 task at beam 800 instead")
ap.add_argument("--nwords", type=int, default=6)
ap.add_argument("--beam", type=int, default=None)
args = ap.parse_args()
if os.path.exists(args.out):
    os.remove(args.out)
os.environ["JAMD_ORACLE_DUMP_SORT"] = args.out
import bench
from julius_amd import synth, lexblob
from oracle import pyoracle
wd = Path(tempfile.mkdtemp())
orc = pyoracle.Oracle()
beam = args.beam or (800 if args.gmm else 4000)
if args.gmm:
    task, jargs, prefix = bench.build_reference_task(wd, 20000, beam, None)
    fr = synth.make_utterance(task, nwords=args.nwords, seed=0)[0]
    sc = orc.gmm_outprob(task["model"], fr)
else:
    dnn = synth.make_dnn(seed=0) if args.flat else synth.make_decodable_dnn(seed=0)
    task, jargs, prefix = bench.build_reference_task(wd, 20000, beam, dnn)
    if args.flat:
        fr = np.random.default_rng(1000).normal(0, 1, (120, 528)).astype(np.float32)
    else:
        fr = synth.make_dnn_utterance(task, dnn, nwords=args.nwords, seed=0)[0]
    t0 = time.time()
    sc = orc.dnn_outprob(dnn, fr)
    print("dnn scores", sc.shape, round(time.time() - t0, 1), "s")
lex = lexblob.load(str(prefix) + ".lex")
t0 = time.time()
atoms, wseq, score, rc, died = orc.beam_pass1(lex, sc, beam)
print("first pass", len(fr), "frames", round(time.time() - t0, 1), "s; rc", rc, "words", list(wseq), "score", score)
raw = np.fromfile(args.out, dtype=np.int32)
i = 0; ns = []
while i < len(raw):
    n, k = int(raw[i]), int(raw[i + 1]); ns.append((n, k)); i += 2 + n
ns = np.array(ns)
print(len(ns), "pruned frames; n/k quantiles", np.quantile(ns[:, 0] / ns[:, 1], [0, .1, .25, .5, .75, .9, 1]).round(2),
      "downward frac", float((ns[:, 0] <= 2 * ns[:, 1]).mean()))
