"""CPU: julius_amd/shim/jamd_export.c -- the exporter a Julius maintainer runs once per
configuration: Julius' own loaders -> "JAMDGMM1" / "JAMDLEX1" files for workers that never link
Julius.  Built against the unmodified reference by oracle/Makefile (julius_amd/jamd_export);
its files must equal what the tap driver's in-process flattening produces."""
import subprocess

import numpy as np
import pytest

from julius_amd import lexblob, synth
from oracle import pyoracle

EXPORT = pyoracle.HERE.parent / "julius_amd" / "jamd_export"


@pytest.mark.parametrize("lm", ["ngram", "grammar", "grammar_forward_dfa"])
def test_export_program_matches_in_process_flattening(ref, tmp_path, lm):
    if not EXPORT.exists():
        pytest.skip("julius_amd/jamd_export not built")
    task = synth.make_triphone_task(tmp_path, seed=91, nword=60)
    if lm == "ngram":
        args = ["-h", task["hmmdefs"], "-hlist", task["hmmlist"], "-v", task["dict"], "-nlr", task["arpa"],
                "-input", "htkparam", "-gprune", "none", "-b", "120", "-sepnum", "4"]
    else:
        task = synth.make_triphone_grammar(task, ncat=3, seed=91) if lm == "grammar" else synth.make_forward_grammar(task, ncat=3, maxwords=3, seed=91)
        args = ["-h", task["hmmdefs"], "-hlist", task["hmmlist"], "-dfa", task["dfa"], "-v", task["gdict"],
                "-input", "htkparam", "-gprune", "none", "-b", "120", "-penalty1", "-2.0"]
    args = [str(a) for a in args]
    out = subprocess.run([str(EXPORT)] + args + ["-jamdout", str(tmp_path / "m")], check=True, capture_output=True, text=True)
    assert "wrote" in out.stdout
    eng = pyoracle.RefEngine(ref, args)
    eng.save_lexicon(tmp_path / "ref.lex")
    a, b = lexblob.load(tmp_path / "m.lex"), lexblob.load(tmp_path / "ref.lex")
    assert a.keys() == b.keys()
    for k in a:
        assert np.array_equal(np.asarray(a[k]), np.asarray(b[k])), k
    if lm == "grammar_forward_dfa":
        # the forward automaton of g.dfa.forward as CSR (arcs of a state in the reference's list order = reverse file order),
        # the initial tokens' states (beam.c:1739-1747), and the python writer reproduces the file byte for byte
        arcs, accept = task["fwd"]
        assert a["nfwd"] == 1 + max(max(s for s, _ in arcs), max(arcs.values()))
        got = {(s, int(a["fwd_label"][e])): int(a["fwd_to"][e]) for s in range(a["nfwd"]) for e in range(a["fwd_off"][s], a["fwd_off"][s + 1])}
        assert got == {k: v for k, v in arcs.items()}
        assert list(a["init_to_state"]) == [arcs[(0, 1)]] * a["ninit"]      # every sentence starts with <s> (category 1)
        lexblob.save(a, tmp_path / "again.lex")
        assert (tmp_path / "again.lex").read_bytes() == (tmp_path / "m.lex").read_bytes()
    else:
        assert a.get("nfwd", 0) == 0
    am = ref.am_load(task["hmmdefs"], task["hmmlist"])
    want, got = am.export(), lexblob.load_gmm(tmp_path / "m.am")
    for k in ("mean", "ivar", "gconst", "st_off", "ent_dens", "ent_logw"):
        assert np.array_equal(got[k], want[k]), k


def test_export_selection_model(ref, tmp_path):
    """-gshmm: PREFIX.gms holds the flattened selection model, the state map and -gsnum, equal to what
    gms_init() (gms.c:275-317) built inside the reference."""
    if not EXPORT.exists():
        pytest.skip("julius_amd/jamd_export not built")
    task = synth.make_triphone_task(tmp_path, seed=92, nword=60)
    gpath, _ = synth.make_gs_model(task, seed=92)
    args = [str(a) for a in ["-h", task["hmmdefs"], "-hlist", task["hmmlist"], "-v", task["dict"], "-nlr", task["arpa"],
                             "-input", "htkparam", "-gprune", "none", "-b", "120", "-gshmm", gpath, "-gsnum", "9"]]
    out = subprocess.run([str(EXPORT)] + args + ["-jamdout", str(tmp_path / "m")], check=True, capture_output=True, text=True)
    assert "m.gms" in out.stdout
    got = lexblob.load_gmm(tmp_path / "m.gms")
    want = ref.am_load(task["hmmdefs"], task["hmmlist"], gshmm=gpath, gms_num=9).gms()
    assert got["nbest"] == want["nbest"] == 9
    assert np.array_equal(got["state2gs"], want["state2gs"])
    for k in ("mean", "ivar", "gconst", "st_off", "ent_dens", "ent_logw"):
        assert np.array_equal(got[k], want["model"][k]), k


def test_export_verification_gmms(ref, tmp_path):
    """-gmm / -gmmnum / -gmmreject: PREFIX.rej holds the flattened GMM definitions, the output state of
    every model in recog->gmm->start order, -gmmnum, the names and gc->is_voice[]."""
    if not EXPORT.exists():
        pytest.skip("julius_amd/jamd_export not built")
    task = synth.make_triphone_task(tmp_path, seed=94, nword=60)
    gpath, _, names = synth.make_rejection_gmm(tmp_path, task["model"]["centre"], seed=94, null_frac=0.1)
    args = [str(a) for a in ["-h", task["hmmdefs"], "-hlist", task["hmmlist"], "-v", task["dict"], "-nlr", task["arpa"],
                             "-input", "htkparam", "-gprune", "none", "-b", "120", "-gmm", gpath, "-gmmnum", "7",
                             "-gmmreject", "noise,cough"]]
    out = subprocess.run([str(EXPORT)] + args + ["-jamdout", str(tmp_path / "m")], check=True, capture_output=True, text=True)
    assert "m.rej" in out.stdout
    got = lexblob.load_gmm(tmp_path / "m.rej")
    want = pyoracle.RefEngine(ref, args).gmm_info()
    assert got["gprune_num"] == want["gprune_num"] == 7
    assert np.array_equal(got["model_state"], want["model_state"])
    for k in ("mean", "ivar", "gconst", "st_off", "ent_dens", "ent_logw"):
        assert np.array_equal(got[k], want["model"][k]), k
    assert got["model_names"] == names[::-1]                       # the loader prepends: reverse file order
    assert [bool(v) for v in got["is_voice"]] == [n not in ("noise", "cough") for n in got["model_names"]]


def test_product_shim_makefile_builds_the_bindings(tmp_path):
    """julius_amd/shim/Makefile -- the product's own recipe for the reference-side half of the boundary -- builds the
    flatten objects and the three bindings against a Julius tree without touching oracle/ (only the config headers of
    an unconfigured tree come from oracle/refcfg here); the pass-1 shim exports exactly beam.o's five symbols
    (libjulius/include/julius/extern.h:57-61)."""
    import os, subprocess
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    ref = Path(os.environ.get("JULIUS_REF", "/root/reference"))
    if not (ref / "libsent/include/sent/stddefs.h").exists():
        import pytest
        pytest.skip("no Julius source tree on this box")
    subprocess.run(["make", "-s", "-C", str(root / "julius_amd/shim"), "objs", f"JULIUS_SRC={ref}", f"JULIUS_CFG={root / 'oracle/refcfg'}",
                    f"OBJDIR={tmp_path}"], check=True)
    for o in ("jamd_flatten.o", "jamd_flatten_lex.o", "jamd_pass1_shim.o", "jamd_outprob_wrap.o", "jamd_gmm_wrap.o"):
        assert (tmp_path / o).stat().st_size > 1000
    syms = subprocess.run(["nm", "-g", "--defined-only", str(tmp_path / "jamd_pass1_shim.o")], check=True, capture_output=True, text=True).stdout
    have = {ln.split()[-1] for ln in syms.splitlines() if " T " in ln}
    want = {"get_back_trellis_init", "get_back_trellis_proceed", "get_back_trellis_end", "fsbeam_free", "finalize_1st_pass"}
    assert want <= have, have
    wrap = subprocess.run(["make", "-s", "-C", str(root / "julius_amd/shim"), "print-wrap", f"JULIUS_SRC={ref}"], check=True, capture_output=True, text=True).stdout
    assert "--wrap=outprob_state" in wrap and "--wrap=gmm_proceed" in wrap
