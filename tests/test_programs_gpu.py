"""The reference's real PROGRAMS over the product's boundaries (north_star: "julius / julius-simple link
unchanged").  oracle/Makefile compiles julius/main.c ... and julius-simple/julius-simple.c from the reference
sources as they lie and links them (a) complete and unmodified, (b) with libjulius' beam.o replaced by
julius_amd/shim/jamd_pass1_shim.o (boundary B), (c) with the outprob entry points wrapped (boundary O); the
unmodified binary also loads the calcmix plugin (boundary P).  Every test execs the binaries on a file list and
compares what they PRINT: pass1_best / sentence1 / score lines must be identical."""
import re
import shlex
import subprocess

import pytest

from julius_amd import synth
from oracle import pyoracle

pytestmark = pytest.mark.gpu

BIN = pyoracle.HERE / "_ref" / "bin"
KEEP = ("pass1_best", "sentence1", "wseq1", "phseq1", "cmscore1", "score1", "input MFCC", "<search failed>", "<input rejected")


def _need(*names):
    for n in names:
        if not (BIN / n).exists():
            pytest.skip(f"oracle/_ref/bin/{n} not built")


def _task(tmp_path, seed=31, nutt=4):
    task = synth.make_triphone_task(tmp_path, seed=seed, nword=120, nphone=10, S=160)
    files = []
    for u in range(nutt):
        fr, _ = synth.make_utterance(task, nwords=3 + 2 * u, seed=100 * seed + u)
        synth.write_htk_param(tmp_path / f"u{u}.mfc", fr)
        files.append(str(tmp_path / f"u{u}.mfc"))
    (tmp_path / "list").write_text("\n".join(files) + "\n")
    args = ["-h", task["hmmdefs"], "-hlist", task["hmmlist"], "-v", task["dict"], "-nlr", task["arpa"], "-input", "htkparam",
            "-b", "200", "-b2", "30", "-n", "1", "-s", "500", "-gprune", "none", "-sepnum", "5"]
    return task, files, [str(a) for a in args]


def _run(exe, args, stdin=None, env=None):
    import os
    e = dict(os.environ)
    e.pop("JAMD_ORDER_MODE", None)
    e.pop("JAMD_STRICT_ORDER", None)
    e.update(env or {})
    out = subprocess.run([str(BIN / exe)] + args, input=stdin, capture_output=True, text=True, env=e, timeout=600)
    assert out.returncode == 0, (exe, out.stdout[-2000:], out.stderr[-2000:])
    return out.stdout


def _results(stdout):
    lines = [ln.rstrip() for ln in stdout.replace("\r", "\n").splitlines()]
    return [ln for ln in lines if ln.startswith(KEEP)]


def test_julius_over_device_first_pass(tmp_path):
    """bin/julius_amd (beam.o -> jamd_pass1_shim.o: HIP scoring + exact-order first pass, then the reference's own 2nd
    pass) prints what bin/julius prints."""
    _need("julius", "julius_amd")
    task, files, args = _task(tmp_path)
    a = args + ["-filelist", str(tmp_path / "list")]
    want = _results(_run("julius", a))
    got = _results(_run("julius_amd", a))
    assert len(want) >= 4 * 5 and any(ln.startswith("sentence1") for ln in want)
    assert got == want


def test_julius_simple_over_device_first_pass(tmp_path):
    """julius-simple/julius-simple.c (the JuliusLib sample: callbacks print the result) linked over boundary B."""
    _need("julius-simple", "julius-simple_amd")
    task, files, args = _task(tmp_path, seed=33, nutt=3)
    (tmp_path / "t.jconf").write_text(" ".join(shlex.quote(a) for a in args) + "\n")
    stdin = "\n".join(files) + "\n"
    pick = lambda s: [ln.rstrip() for ln in s.splitlines() if re.match(r"^(sentence|wseq|phseq|cmscore|score)\d*:", ln)
                      or ln.startswith("pass1_best")]
    want = pick(_run("julius-simple", ["-C", str(tmp_path / "t.jconf")], stdin=stdin))
    got = pick(_run("julius-simple_amd", ["-C", str(tmp_path / "t.jconf")], stdin=stdin))
    assert len(want) >= 3 * 3
    assert got == want


def test_julius_over_wrapped_scoring(tmp_path):
    """bin/julius_o: the complete reference (its own beam.c included) with outprob_state/outprob_cd/outprob/
    outprob_prepare/outprob_free wrapped: the CPU search consumes device scores."""
    _need("julius", "julius_o")
    task, files, args = _task(tmp_path, seed=35)
    a = args + ["-filelist", str(tmp_path / "list")]
    assert _results(_run("julius_o", a)) == _results(_run("julius", a))


def test_unmodified_julius_with_calcmix_plugin(tmp_path):
    """Boundary P with the real program: bin/julius -plugindir oracle/_ref/plugin -gprune jamd."""
    _need("julius")
    if not (pyoracle.PLUGIN_DIR / "jamd_calcmix.jpi").exists():
        pytest.skip("oracle/_ref/plugin/jamd_calcmix.jpi not built")
    task, files, args = _task(tmp_path, seed=37)
    a = args + ["-filelist", str(tmp_path / "list")]
    want = _results(_run("julius", a))
    i = a.index("-gprune")
    plug = ["-plugindir", str(pyoracle.PLUGIN_DIR)] + a[:i] + ["-gprune", "jamd"] + a[i + 2:]
    got = _results(_run("julius", plug))
    assert got == want


@pytest.mark.parametrize("interval", ["300", "100"])
def test_progout_interim_results(tmp_path, interval):
    """-progout: the shim sets r->have_interim every -proginterval msec (beam.c:2983-2992) and fills the result record
    as bt_current_max() does; the program prints the same progressive lines."""
    _need("julius", "julius_amd")
    task, files, args = _task(tmp_path, seed=39, nutt=3)
    a = args + ["-filelist", str(tmp_path / "list"), "-progout", "-proginterval", interval]
    w = _run("julius", a).replace("\r", "\n")
    g = _run("julius_amd", a).replace("\r", "\n")
    pick = lambda s: [ln.rstrip() for ln in s.splitlines() if ln.startswith(("pass1_best", "sentence1", "score1", "input MFCC"))]
    want, got = pick(w), pick(g)
    assert sum(1 for ln in want if ln.startswith("pass1_best:")) > 3 * 3      # several interim lines per input
    assert got == want


def test_julius_over_device_first_pass_grammar_with_forward_dfa(tmp_path):
    """The real program on a grammar compiled WITH a forward DFA (`g.dfa.forward` next to `g.dfa`, what recent mkdfa.pl
    writes): bin/julius_amd -- device scoring + device first pass carrying the automaton's state, then the reference's
    own second pass -- prints what the plain bin/julius prints, on sentences of the language and on chains too long for it."""
    _need("julius", "julius_amd")
    task = synth.make_forward_grammar(synth.make_triphone_task(tmp_path, seed=41, nword=120, nphone=10, S=160), ncat=3, maxwords=3, seed=41)
    files = []
    for u in range(5):
        fr, _ = synth.make_forward_grammar_utterance(task, seed=4100 + u, nwords=None if u < 3 else 3 + u)
        synth.write_htk_param(tmp_path / f"u{u}.mfc", fr)
        files.append(str(tmp_path / f"u{u}.mfc"))
    (tmp_path / "list").write_text("\n".join(files) + "\n")
    args = [str(a) for a in ["-h", task["hmmdefs"], "-hlist", task["hmmlist"], "-dfa", task["dfa"], "-v", task["gdict"], "-input", "htkparam",
                             "-b", "200", "-b2", "30", "-n", "1", "-s", "500", "-gprune", "none", "-filelist", tmp_path / "list"]]
    plain = _run("julius", args)
    assert "reading additional forward dfa" in plain
    want = _results(plain)
    got = _results(_run("julius_amd", args))
    assert len([x for x in want if x.startswith("sentence1")]) >= 3
    assert got == want
