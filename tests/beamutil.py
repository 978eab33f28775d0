"""Helpers shared by the first-pass tests (golden fixture access, trellis compare)."""
import numpy as np

from conftest import GOLDEN
from julius_amd import lexblob, synth

TR_KEYS = ("wid", "begintime", "endtime", "pwid", "pendtime", "backscore", "lscore")


def load_beam_golden(name):
    z = np.load(GOLDEN / name)
    lex = {}
    for k in z.files:
        if k.startswith("lex_"):
            v = z[k]
            lex[k[4:]] = v.item() if v.ndim == 0 else v
    am = {k[3:]: z[k] for k in z.files if k.startswith("am_")}
    am.update(st_book=None, nbook=0, nstream=1)
    utts = []
    for u in range(int(z["nutt"])):
        utts.append(dict(frames=z[f"u{u}_frames"], wseq=z[f"u{u}_wseq"], score=float(z[f"u{u}_score"]),
                         trellis={k: z[f"u{u}_tr_{k}"] for k in TR_KEYS}))
    return dict(lex=lex, am=am, utts=utts, beam_width=int(z["beam_width"]),
                score_pruning_width=float(z["score_pruning_width"]))


def assert_trellis_equal(atoms, want):
    """atoms: structured array in emission order; want: canonical dict (reference order)."""
    got = lexblob.canonical_trellis(atoms)
    assert len(got["wid"]) == len(want["wid"]), (len(got["wid"]), len(want["wid"]))
    for k in TR_KEYS:
        assert np.array_equal(got[k], want[k]), f"trellis field {k} differs"


def ref_task(ref, tmpdir, seed, beam, extra=(), **task_kw):
    """Synthetic triphone task loaded by the compiled reference: (RefEngine, lex dict, flat AM, task)."""
    from oracle import pyoracle
    task = synth.make_triphone_task(tmpdir, seed=seed, **task_kw)
    lm = ["-nlr", task["arpa"]] + (["-nrl", task["arpa_rl"]] if task.get("arpa_rl") else [])
    eng = pyoracle.RefEngine(ref, ["-h", task["hmmdefs"], "-hlist", task["hmmlist"], "-v", task["dict"]] + lm +
                             ["-input", "htkparam", "-1pass", "-gprune", "none", "-b", str(beam)] + list(extra))
    eng.save_lexicon(tmpdir / "lex.blob")
    lex = lexblob.load(tmpdir / "lex.blob")
    am = ref.am_load(task["hmmdefs"], task["hmmlist"]).export()
    return eng, lex, am, task


def ref_grammar_task(ref, tmpdir, seed, beam, extra=(), ncat=3, wrap=True, **task_kw):
    """Cross-word triphone task under a DFA grammar (synth.make_triphone_grammar) loaded by the
    compiled reference: (RefEngine, lex dict, flat AM, task)."""
    from oracle import pyoracle
    task = synth.make_triphone_grammar(synth.make_triphone_task(tmpdir, seed=seed, **task_kw), ncat=ncat, seed=seed, wrap=wrap)
    eng = pyoracle.RefEngine(ref, ["-h", task["hmmdefs"], "-hlist", task["hmmlist"], "-dfa", task["dfa"], "-v", task["gdict"],
                                   "-input", "htkparam", "-1pass", "-gprune", "none", "-b", str(beam)] + list(extra))
    eng.save_lexicon(tmpdir / "lex.blob")
    lex = lexblob.load(tmpdir / "lex.blob")
    am = ref.am_load(task["hmmdefs"], task["hmmlist"]).export()
    return eng, lex, am, task


def assert_trellis_equal_modulo_ties(atoms, want, ties):
    """Exact when the engine met no score tie.  With ties > 0 the engine's canonical
    rule (larger source id) and the reference's visiting order may keep different --
    equally scored -- histories, so a handful of atoms may differ: require the two
    trellises to agree on all but at most 4*ties (endtime, wid) entries and, on the
    common entries, on the begin frame and both scores for at least 99.9%."""
    if ties == 0:
        return assert_trellis_equal(atoms, want)
    got = lexblob.canonical_trellis(atoms)
    kg = got["endtime"].astype(np.int64) * (1 << 32) + got["wid"]
    kw = want["endtime"].astype(np.int64) * (1 << 32) + want["wid"]
    common, ig, iw = np.intersect1d(kg, kw, return_indices=True)
    assert len(kg) - len(common) <= 4 * ties and len(kw) - len(common) <= 4 * ties, (len(kg), len(kw), len(common))
    same = ((got["backscore"][ig] == want["backscore"][iw]) & (got["begintime"][ig] == want["begintime"][iw]) &
            (got["lscore"][ig] == want["lscore"][iw]))
    assert same.mean() >= 0.999, same.mean()


def assert_canonical_close(got, want, max_diff):
    """Two canonical trellises (dicts in reference order): identical, or -- when exact
    score ties made the engine and the reference keep different tokens at the rank cut
    (DESIGN.md section 4 "Ties") -- differing in at most max_diff (endtime, wid) entries
    with every common entry identical in begin frame, predecessor and LM score."""
    kg = got["endtime"].astype(np.int64) * (1 << 32) + got["wid"]
    kw = want["endtime"].astype(np.int64) * (1 << 32) + want["wid"]
    common, ig, iw = np.intersect1d(kg, kw, return_indices=True)
    assert len(kg) - len(common) <= max_diff and len(kw) - len(common) <= max_diff, (len(kg), len(kw), len(common))
    for k in ("begintime", "pwid", "pendtime", "lscore"):
        assert np.array_equal(got[k][ig], want[k][iw]), k
