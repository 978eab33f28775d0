"""development: the shim multipath test's task through the direct API: exact (multipath frame) vs strict vs oracle."""
import sys, tempfile
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from julius_amd import lexblob, lib, synth
from oracle import pyoracle

tmp = Path(tempfile.mkdtemp())
task = synth.make_triphone_task(tmp, seed=87, nword=120, nphone=10, S=160)
args = ["-h", task["hmmdefs"], "-hlist", task["hmmlist"], "-v", task["dict"], "-nlr", task["arpa"],
        "-input", "htkparam", "-gprune", "none", "-b", "200", "-b2", "30", "-n", "1", "-s", "500", "-sepnum", "5", "-multipath"]
ref = pyoracle.Ref()
eng = pyoracle.RefEngine(ref, [str(a) for a in args])
eng.save_lexicon(tmp / "lex.blob")
lex = lexblob.load(tmp / "lex.blob")
am = ref.am_load(task["hmmdefs"], task["hmmlist"]).export()
orc = pyoracle.Oracle()
e = lib.Engine(0)
lx = lib.Lexicon(e, lex)
print("nnode", lex["nnode"], "startnum", lex["startnum"], "iso", lex["isolatenum"], "maxarc", np.diff(lex["ac_off"]).max())
for u in range(3):
    fr = synth.make_utterance(task, nwords=3 + u, seed=8700 + u)[0]
    sc = orc.gmm_outprob(am, fr)
    bm = lib.Beam(e, lx, 200, -1.0, max_utts=1, atoms_per_utt=1 << 17)
    res, tre = bm.pass1_host([sc])
    bm.set_strict_order(True)
    sres, stre = bm.pass1_host([sc])
    oat, ow, os_, rc, died = orc.beam_pass1(lex, sc, 200, -1.0)
    a, s = tre[0], stre[0]
    print("utt", u, "T", len(fr), "natom exact", len(a), "strict", len(s), "oracle", len(oat), "status", res[0].status, sres[0].status, rc, "maxtok", res[0].max_tokens, sres[0].max_tokens)
    n = min(len(a), len(s))
    for i in range(n):
        if tuple(a[i]) != tuple(s[i]):
            print("  first diff at atom", i, "exact", a[i], "strict", s[i])
            break
