"""GPU: the reference's OWN recogniser running on top of the device first pass.

oracle/_ref/libjref_amd.so is the unmodified reference with libjulius/src/beam.c
left out and julius_amd/shim/jamd_pass1_shim.c (which exports beam.o's five
symbols over the C ABI) linked in its place.  Front end, model loaders, 2nd pass
(stack decoding over the word trellis) and result handling are the reference's.
Checked against the plain reference (libjref.so) on the same inputs:
  * the word trellis handed to the 2nd pass is identical (up to exact-tie cases),
  * the pass-1 sentence and score are identical,
  * the FINAL sentence (after the 2nd pass) and its score are identical."""
import numpy as np
import pytest

from beamutil import assert_canonical_close, assert_canonical_scores_close
from julius_amd import synth
from oracle import pyoracle

pytestmark = pytest.mark.gpu


def _order_env(monkeypatch, mode):
    """mode: 'fast' (canonical tie breaks), 'strict' (sequential kernel), anything else = the shim's default,
    the exact-order kernel.  Returns whether the trellis must equal the reference's exactly."""
    monkeypatch.setenv("JAMD_STRICT_ORDER", "1" if mode == "strict" else "0")
    if mode == "fast":
        monkeypatch.setenv("JAMD_ORDER_MODE", "fast")
    else:
        monkeypatch.delenv("JAMD_ORDER_MODE", raising=False)
    return mode != "fast"


@pytest.mark.parametrize("strict", ["fast", "strict", "exact", "stream"])
@pytest.mark.parametrize("seed,beam,extra", [
    (31, 200, ["-sepnum", "5"]),
    (32, 100, ["-sepnum", "3", "-gprune", "safe", "-tmix", "3"]),
    (33, 150, ["-sepnum", "8", "-bs", "70", "-lmp", "6.0", "-2.0"]),
    (34, 150, ["-sepnum", "5", "-rl3"]),       # forward 2-gram for pass 1 + backward 3-gram for the reference's pass 2
])
def test_reference_two_pass_over_device_first_pass(ref, tmp_path, monkeypatch, seed, beam, extra, strict):
    exact = _order_env(monkeypatch, strict)
    # "stream": the shim advances the device search every 25 frames from get_back_trellis_proceed()
    monkeypatch.setenv("JAMD_STREAM_CHUNK", "25" if strict == "stream" else "0")
    if not pyoracle.REF_AMD_SO.exists():
        pytest.skip("oracle/_ref/libjref_amd.so not built")
    rl3 = "-rl3" in extra
    extra = [x for x in extra if x != "-rl3"]
    task = synth.make_triphone_task(tmp_path, seed=seed, nword=120, nphone=10, S=160, with_rl3=rl3)
    args = ["-h", task["hmmdefs"], "-hlist", task["hmmlist"], "-v", task["dict"], "-nlr", task["arpa"]]
    if rl3:
        args += ["-nrl", task["arpa_rl"]]
    args += ["-input", "htkparam", "-b", str(beam), "-b2", "30", "-n", "1", "-s", "500"]
    if "-gprune" not in extra:
        args += ["-gprune", "none"]
    args += list(extra)
    plain = pyoracle.RefEngine(ref, args)
    amd = pyoracle.RefEngine(pyoracle.Ref(so=pyoracle.REF_AMD_SO), args)
    for u in range(3):
        fr, _ = synth.make_utterance(task, nwords=3 + 2 * u, seed=100 * seed + u)
        synth.write_htk_param(tmp_path / "u.mfc", fr)
        tr0, (w0, s0) = plain.recognize(tmp_path / "u.mfc")
        st0, f0, fs0 = plain.final_result()
        tr1, (w1, s1) = amd.recognize(tmp_path / "u.mfc")
        st1, f1, fs1 = amd.final_result()
        # the 2nd pass of the shimmed recogniser read device scores: its cache is completely filled,
        # the plain reference's only where the search went
        d1, n1 = amd.cache_fill()
        d0, n0 = plain.cache_fill()
        assert d1 == n1 == n0 and d0 < n0
        assert st1 == st0
        assert np.array_equal(w1, w0) and s1 == s0                     # pass-1 best
        assert np.array_equal(f1, f0) and fs1 == fs0                   # final sentence after pass 2
        # the trellis the 2nd pass consumed: identical -- exactly in strict-order mode, up to the
        # exact-score ties of DESIGN.md section 4 with the frame-parallel kernel
        if exact:
            for k in tr0:
                assert np.array_equal(tr1[k], tr0[k]), k
        else:
            assert_canonical_close(tr1, tr0, max_diff=8)


def test_full_size_two_pass_with_backward_trigram(ref, tmp_path, monkeypatch):
    """SURVEY 8f N2 at the size of BASELINE configs[1-2]: 20 000-word dictionary, 3 000 x 16 tied-state triphones,
    forward 2-gram (-nlr) for the first pass AND backward 3-gram (-nrl) for the second, the reference's complete two-pass
    recogniser (recogmain.c:1292-1345) on top of the device first pass (boundary B, exact tie order) against the plain
    reference: pass-1 best, the trellis the second pass consumes, and the final sentence1 / score1 are identical on
    utterances the model really recognises (frames drawn along the words' own triphone states)."""
    _order_env(monkeypatch, "exact")
    monkeypatch.setenv("JAMD_STREAM_CHUNK", "0")
    if not pyoracle.REF_AMD_SO.exists():
        pytest.skip("oracle/_ref/libjref_amd.so not built")
    task = synth.make_triphone_task(tmp_path, nphone=40, S=3000, M=16, nword=20000, nvar=25, seed=0, maxlen=8,
                                    nbigram_per_word=10, with_rl3=True)
    args = ["-h", task["hmmdefs"], "-hlist", task["hmmlist"], "-v", task["dict"], "-nlr", task["arpa"], "-nrl", task["arpa_rl"],
            "-input", "htkparam", "-b", "800", "-b2", "30", "-n", "1", "-s", "500", "-gprune", "none"]
    plain = pyoracle.RefEngine(ref, args)
    amd = pyoracle.RefEngine(pyoracle.Ref(so=pyoracle.REF_AMD_SO), args)
    assert plain.beam_width == 800
    found = 0
    for u in range(4):
        fr, words = synth.make_path_utterance(task, nwords=6 + 2 * u, seed=700 + u)
        synth.write_htk_param(tmp_path / "u.mfc", fr)
        tr0, (w0, s0) = plain.recognize(tmp_path / "u.mfc")
        st0, f0, fs0 = plain.final_result()
        tr1, (w1, s1) = amd.recognize(tmp_path / "u.mfc")
        st1, f1, fs1 = amd.final_result()
        d1, n1 = amd.cache_fill()
        assert d1 == n1                                                # the second pass read device scores only
        assert st1 == st0 and np.array_equal(w1, w0) and s1 == s0
        assert np.array_equal(f1, f0) and fs1 == fs0
        for k in tr0:
            assert np.array_equal(tr1[k], tr0[k]), k
        found += int(st0 == 0 and len(f0) >= len(words))
    assert found >= 3, found                                           # real sentences, not failed searches on both sides


# ------------------------------------------------------------------ boundary O (scoring only)
def _compare_exact(plain, wrapped, mfc):
    tr0, (w0, s0) = plain.recognize(mfc)
    st0, f0, fs0 = plain.final_result()
    d0, n0 = plain.cache_fill()
    tr1, (w1, s1) = wrapped.recognize(mfc)
    st1, f1, fs1 = wrapped.final_result()
    d1, n1 = wrapped.cache_fill()
    assert d1 == n1 == n0 and d0 < n0            # every (t, s) came from the device
    assert st1 == st0 and np.array_equal(w1, w0) and s1 == s0
    assert np.array_equal(f1, f0) and fs1 == fs0
    for k in tr0:                                # the reference's own beam: no tie caveat, exact
        assert np.array_equal(tr1[k], tr0[k]), k


@pytest.mark.parametrize("seed,beam,extra", [
    (41, 200, ["-sepnum", "5", "-gprune", "none"]),
    (42, 100, ["-sepnum", "3", "-gprune", "safe", "-tmix", "3", "-iwcd1", "max"]),
])
def test_reference_search_over_device_scores(ref, tmp_path, seed, beam, extra):
    """libjref_o.so: the complete reference (CPU beam, 2nd pass) with outprob_state/_cd/outprob/
    outprob_prepare/outprob_free wrapped by julius_amd/shim/jamd_outprob_wrap.c.  Identical
    trellis, pass-1 and final results; the outprob cache is filled entirely by the device."""
    if not pyoracle.REF_O_SO.exists():
        pytest.skip("oracle/_ref/libjref_o.so not built")
    task = synth.make_triphone_task(tmp_path, seed=seed, nword=120, nphone=10, S=160)
    args = ["-h", task["hmmdefs"], "-hlist", task["hmmlist"], "-v", task["dict"], "-nlr", task["arpa"],
            "-input", "htkparam", "-b", str(beam), "-b2", "30", "-n", "1", "-s", "500"] + list(extra)
    plain = pyoracle.RefEngine(ref, args)
    wrapped = pyoracle.RefEngine(pyoracle.Ref(so=pyoracle.REF_O_SO), args)
    for u in range(3):
        fr, _ = synth.make_utterance(task, nwords=3 + 2 * u, seed=100 * seed + u)
        synth.write_htk_param(tmp_path / "u.mfc", fr)
        _compare_exact(plain, wrapped, tmp_path / "u.mfc")


@pytest.mark.parametrize("gprune", [["-gprune", "none"], ["-gprune", "safe", "-tmix", "3"]])
def test_reference_search_over_selected_device_scores(ref, tmp_path, gprune):
    """Boundary O with -gshmm: the wrapper applies the selection stage to the device scores before
    they go into the reference's cache, so the CPU beam and the 2nd pass read exactly what
    gms_state() would have returned."""
    if not pyoracle.REF_O_SO.exists():
        pytest.skip("oracle/_ref/libjref_o.so not built")
    task = synth.make_triphone_task(tmp_path, seed=43, nword=120, nphone=10, S=160)
    gpath, _ = synth.make_gs_model(task, seed=43)
    args = ["-h", task["hmmdefs"], "-hlist", task["hmmlist"], "-v", task["dict"], "-nlr", task["arpa"],
            "-input", "htkparam", "-b", "150", "-b2", "30", "-n", "1", "-s", "500", "-sepnum", "5",
            "-gshmm", str(gpath), "-gsnum", "5"] + gprune
    plain = pyoracle.RefEngine(ref, args)
    wrapped = pyoracle.RefEngine(pyoracle.Ref(so=pyoracle.REF_O_SO), args)
    for u in range(3):
        fr, _ = synth.make_utterance(task, nwords=3 + 2 * u, seed=4300 + u)
        synth.write_htk_param(tmp_path / "u.mfc", fr)
        _compare_exact(plain, wrapped, tmp_path / "u.mfc")


def test_c1_tied_mixture_grammar_over_device_scores(ref, tmp_path):
    """BASELINE configs[0] shape on the GPU: tied-mixture monophone GMM-HMM + a 100-word DFA
    grammar.  The grammar search is the reference's; the tied-mixture scoring (codebook
    top-N cache + per-state re-weighting, calc_tied_mix.c:162) runs on the device."""
    if not pyoracle.REF_O_SO.exists():
        pytest.skip("oracle/_ref/libjref_o.so not built")
    task = synth.make_grammar_task(tmp_path, seed=7)
    args = ["-h", task["hmmdefs"], "-dfa", task["dfa"], "-v", task["dict"], "-input", "htkparam",
            "-gprune", "safe", "-tmix", "2", "-b", "200"]
    plain = pyoracle.RefEngine(ref, args)
    wrapped = pyoracle.RefEngine(pyoracle.Ref(so=pyoracle.REF_O_SO), args)
    for u in range(3):
        fr, _ = synth.make_grammar_utterance(task, nwords=3 + u, seed=u)
        synth.write_htk_param(tmp_path / "u.mfc", fr)
        _compare_exact(plain, wrapped, tmp_path / "u.mfc")


@pytest.mark.parametrize("so", ["o", "amd"])
@pytest.mark.parametrize("gprune", ["beam", "heuristic"])
def test_c1_history_pruning_over_device_scores(ref, tmp_path, monkeypatch, so, gprune):
    """BASELINE configs[0] shape with the pruning the reference's fast build defaults to for tied-mixture models
    (`-gprune beam`, and `heu`): thresholds from the codebook's winners of frame t-1 (`calc_tied_mix.c:203-215`).  The
    device scores every frame, so parity is defined against the reference under EAGER scoring
    (`outprob_set_batch_computation`): through boundary O (reference search over device scores) and boundary B (device
    first pass, exact order) the trellis, pass-1 and final results are the eager reference's, bit for bit."""
    monkeypatch.setenv("JAMD_STRICT_ORDER", "0")
    monkeypatch.delenv("JAMD_ORDER_MODE", raising=False)
    monkeypatch.setenv("JAMD_STREAM_CHUNK", "25")           # ignored: this scoring needs the input in one piece
    lib_so = pyoracle.REF_O_SO if so == "o" else pyoracle.REF_AMD_SO
    if not lib_so.exists():
        pytest.skip(f"{lib_so} not built")
    task = synth.make_grammar_task(tmp_path, seed=11)
    args = ["-h", task["hmmdefs"], "-dfa", task["dfa"], "-v", task["dict"], "-input", "htkparam",
            "-gprune", gprune, "-tmix", "2", "-b", "200"]
    eager = pyoracle.RefEngine(ref, args).set_eager()
    wrapped = pyoracle.RefEngine(pyoracle.Ref(so=lib_so), args)
    for u in range(3):
        fr, _ = synth.make_grammar_utterance(task, nwords=3 + u, seed=20 + u)
        synth.write_htk_param(tmp_path / "u.mfc", fr)
        tr0, (w0, s0) = eager.recognize(tmp_path / "u.mfc")
        st0, f0, fs0 = eager.final_result()
        tr1, (w1, s1) = wrapped.recognize(tmp_path / "u.mfc")
        st1, f1, fs1 = wrapped.final_result()
        d1, n1 = wrapped.cache_fill()
        assert d1 == n1                                      # every (t, s) came from the device
        assert st1 == st0 and np.array_equal(w1, w0) and s1 == s0
        assert np.array_equal(f1, f0) and fs1 == fs0
        for k in tr0:
            assert np.array_equal(tr1[k], tr0[k]), k


@pytest.mark.parametrize("so,strict", [("o", False), ("amd", True)])
def test_c4_dnn_over_device_scores(ref, tmp_path, monkeypatch, so, strict):
    """DNN-HMM (-dnnconf) through the scoring wrapper: dnn_calc_outprob()'s work is done by the
    MFMA kernels for the whole utterance, Julius' own first and second pass consume the cache.
    Bit-identical trellis and results require the reference to run its FMA kernel (it picks
    the best SIMD path of the host CPU)."""
    monkeypatch.setenv("JAMD_STRICT_ORDER", "1" if strict else "0")   # "amd": device first pass too (exact order)
    lib_so = pyoracle.REF_O_SO if so == "o" else pyoracle.REF_AMD_SO
    if not lib_so.exists():
        pytest.skip(f"{lib_so} not built")
    if b"FMA" not in ref.lib.jref_simd_string():
        pytest.skip("reference built without its FMA kernel")
    task = synth.make_triphone_task(tmp_path, seed=51, nword=80, nphone=8, S=120)
    S, IN, H = 120, 48, 64
    dnn = synth.make_dnn(dims=(IN, H, H, S), seed=51)
    for l, (w, b) in enumerate(zip(dnn["w"], dnn["b"])):
        synth.write_npy(tmp_path / f"W{l}.npy", w)
        synth.write_npy(tmp_path / f"b{l}.npy", np.asarray(b).reshape(-1, 1))
    with open(tmp_path / "prior", "w") as f:
        for i, v in enumerate(dnn["prior_lin"]):
            f.write(f"{i} {float(v):.9e}\n")
    (tmp_path / "dnn.conf").write_text(
        f"feature_type USER\nfeature_len {IN}\ncontext_len 1\ninput_nodes {IN}\noutput_nodes {S}\n"
        f"hidden_nodes {H}\nhidden_layers 2\nW1 W0.npy\nW2 W1.npy\nB1 b0.npy\nB2 b1.npy\noutput_W W2.npy\n"
        f"output_B b2.npy\nstate_prior prior\nstate_prior_factor 1.0\nstate_prior_log10nize yes\nnum_threads 1\n")
    args = ["-h", task["hmmdefs"], "-hlist", task["hmmlist"], "-v", task["dict"], "-nlr", task["arpa"],
            "-dnnconf", tmp_path / "dnn.conf", "-input", "htkparam", "-notypecheck", "-b", "150", "-b2", "30",
            "-n", "1", "-s", "500", "-sepnum", "4"]
    plain = pyoracle.RefEngine(ref, args)
    wrapped = pyoracle.RefEngine(pyoracle.Ref(so=lib_so), args)
    rng = np.random.default_rng(51)
    for u in range(3):
        fr = rng.normal(0, 1, (60 + 40 * u, IN)).astype(np.float32)
        synth.write_htk_param(tmp_path / "u.mfc", fr, parmkind=synth.PARM_USER)
        tr0, (w0, s0) = plain.recognize(tmp_path / "u.mfc")
        st0, f0, fs0 = plain.final_result()
        tr1, (w1, s1) = wrapped.recognize(tmp_path / "u.mfc")
        st1, f1, fs1 = wrapped.final_result()
        d1, n1 = wrapped.cache_fill()
        assert d1 == n1
        assert st1 == st0 and np.array_equal(w1, w0) and s1 == s0 and np.array_equal(f1, f0) and fs1 == fs0
        for k in tr0:
            assert np.array_equal(tr1[k], tr0[k]), k


@pytest.mark.parametrize("mode", ["fast", "strict", "exact", "stream"])
@pytest.mark.parametrize("kind", ["c1", "triphone", "free"])
def test_grammar_two_pass_over_device_first_pass(ref, tmp_path, monkeypatch, kind, mode):
    """Grammar recognition (-dfa) with the FIRST PASS ON THE DEVICE: per-category lexicon trees,
    category-pair constraint, then the reference's own DFA-driven 2nd pass over the trellis the
    shim rebuilt.  c1 = BASELINE configs[0] shape (tied-mixture monophones, 100-word loop
    grammar); triphone = cross-word triphones with category-aware state sets; free = any word
    may start a sentence (dozens of initial tokens)."""
    exact = _order_env(monkeypatch, mode)
    monkeypatch.setenv("JAMD_STREAM_CHUNK", "20" if mode == "stream" else "0")
    if not pyoracle.REF_AMD_SO.exists():
        pytest.skip("oracle/_ref/libjref_amd.so not built")
    if kind == "c1":
        task = synth.make_grammar_task(tmp_path, seed=9)
        args = ["-h", task["hmmdefs"], "-dfa", task["dfa"], "-v", task["dict"], "-input", "htkparam",
                "-gprune", "safe", "-tmix", "2", "-b", "200", "-penalty1", "-1.0"]
        utts = [synth.make_grammar_utterance(task, nwords=3 + u, seed=u)[0] for u in range(3)]
    else:
        task = synth.make_triphone_grammar(synth.make_triphone_task(tmp_path, seed=61, nword=90, nphone=10, S=160),
                                           ncat=3, seed=61, wrap=(kind == "triphone"))
        args = ["-h", task["hmmdefs"], "-hlist", task["hmmlist"], "-dfa", task["dfa"], "-v", task["gdict"],
                "-input", "htkparam", "-gprune", "none", "-b", "150", "-penalty1", "-2.0", "-b2", "30", "-n", "1", "-s", "500"]
        utts = [synth.make_triphone_grammar_utterance(task, nwords=2 + 2 * u, seed=6100 + u)[0] for u in range(3)]
    plain = pyoracle.RefEngine(ref, args)
    amd = pyoracle.RefEngine(pyoracle.Ref(so=pyoracle.REF_AMD_SO), args)
    for fr in utts:
        synth.write_htk_param(tmp_path / "u.mfc", fr)
        tr0, (w0, s0) = plain.recognize(tmp_path / "u.mfc")
        st0, f0, fs0 = plain.final_result()
        tr1, (w1, s1) = amd.recognize(tmp_path / "u.mfc")
        st1, f1, fs1 = amd.final_result()
        d1, n1 = amd.cache_fill()
        assert d1 == n1                                                # 2nd pass = cache hits on device scores
        assert st1 == st0
        assert s1 == s0 and fs1 == fs0                                 # pass-1 and final scores
        if exact:                                                      # the reference's visiting order: exact
            assert np.array_equal(w1, w0) and np.array_equal(f1, f0)
            for k in tr0:
                assert np.array_equal(tr1[k], tr0[k]), k
        else:                                                          # equally scored alternatives may differ
            assert len(w1) > 0 and len(f1) > 0
            assert_canonical_scores_close(tr1, tr0)


@pytest.mark.parametrize("mode", ["fast", "strict", "exact"])
def test_wordlist_recognition_over_device_first_pass(ref, tmp_path, monkeypatch, mode):
    """Isolated word recognition (-w) through the shimmed recogniser: the first pass IS the
    recognition, the shim derives the final N-best result from the trellis it rebuilt."""
    exact = _order_env(monkeypatch, mode)
    monkeypatch.setenv("JAMD_STREAM_CHUNK", "0")
    if not pyoracle.REF_AMD_SO.exists():
        pytest.skip("oracle/_ref/libjref_amd.so not built")
    task = synth.make_wordlist_task(tmp_path, seed=11, triphone=True, nword=80)
    args = ["-h", task["hmmdefs"], "-hlist", task["hmmlist"], "-w", task["wordlist"], "-wsil", "silB", "silE", "silB",
            "-input", "htkparam", "-gprune", "none", "-b", "100", "-output", "3"]
    plain = pyoracle.RefEngine(ref, args)
    amd = pyoracle.RefEngine(pyoracle.Ref(so=pyoracle.REF_AMD_SO), args)
    for u in range(4):
        fr, _ = synth.make_wordlist_utterance(task, seed=50 + u)
        synth.write_htk_param(tmp_path / "u.mfc", fr)
        tr0, _ = plain.recognize(tmp_path / "u.mfc")
        st0, f0, fs0 = plain.final_result()
        tr1, _ = amd.recognize(tmp_path / "u.mfc")
        st1, f1, fs1 = amd.final_result()
        assert st1 == st0 and len(f1) == 1
        if exact:
            assert np.array_equal(f1, f0) and fs1 == fs0
            for k in tr0:
                assert np.array_equal(tr1[k], tr0[k]), k
        else:
            # tree branches whose logical triphones share a physical model carry exactly equal scores;
            # without LM scores to separate them they tie on the rank cut in most frames, and the two
            # searches keep different (equally good) ones: the best word's score must still agree closely
            assert abs(fs1 - fs0) <= 0.01 * abs(fs0)
            assert_canonical_scores_close(tr1, tr0, min_common=0.8, min_same=0.9)


@pytest.mark.parametrize("lm", ["ngram", "grammar"])
def test_batch_driver_equals_per_utterance(ref, tmp_path, monkeypatch, lm):
    """a file list decoded in ONE device launch (jamd_pass1_prefetch_add/_run), then the
    reference's unchanged per-input loop, which finds every first pass done.  Same trellis, pass-1
    and final results as the one-launch-per-utterance path of the same build, and the same final
    sentences as the plain reference; inputs that were not queued still work."""
    monkeypatch.setenv("JAMD_STRICT_ORDER", "0")
    monkeypatch.setenv("JAMD_STREAM_CHUNK", "0")
    if not pyoracle.REF_AMD_SO.exists():
        pytest.skip("oracle/_ref/libjref_amd.so not built")
    if lm == "ngram":
        task = synth.make_triphone_task(tmp_path, seed=71, nword=120, nphone=10, S=160)
        args = ["-h", task["hmmdefs"], "-hlist", task["hmmlist"], "-v", task["dict"], "-nlr", task["arpa"],
                "-input", "htkparam", "-gprune", "none", "-b", "200", "-b2", "30", "-n", "1", "-s", "500", "-sepnum", "5"]
        utts = [synth.make_utterance(task, nwords=2 + u % 5, seed=7100 + u)[0] for u in range(9)]
    else:
        task = synth.make_triphone_grammar(synth.make_triphone_task(tmp_path, seed=72, nword=90, nphone=10, S=160), ncat=3, seed=72)
        args = ["-h", task["hmmdefs"], "-hlist", task["hmmlist"], "-dfa", task["dfa"], "-v", task["gdict"],
                "-input", "htkparam", "-gprune", "none", "-b", "200", "-penalty1", "-2.0", "-b2", "30", "-n", "1", "-s", "500"]
        utts = [synth.make_triphone_grammar_utterance(task, nwords=2 + u % 4, seed=7200 + u)[0] for u in range(9)]
    files = []
    for u, fr in enumerate(utts):
        files.append(tmp_path / f"u{u}.mfc")
        synth.write_htk_param(files[-1], fr)
    plain = pyoracle.RefEngine(ref, args)
    amd = pyoracle.RefEngine(pyoracle.Ref(so=pyoracle.REF_AMD_SO), args)
    single = []
    for f in files:                                   # one launch per utterance
        tr, p1 = amd.recognize(f)
        single.append((tr, p1, amd.final_result(), amd.cache_fill()))
    amd.prefetch(files[:7])                           # the last two inputs are deliberately not queued
    for f, (tr0, p10, fin0, cf0) in zip(files, single):
        tr1, p11 = amd.recognize(f)
        fin1 = amd.final_result()
        for k in tr0:
            assert np.array_equal(tr1[k], tr0[k]), k  # the same kernel decoded it: identical, atom for atom
        assert np.array_equal(p11[0], p10[0]) and p11[1] == p10[1]
        assert fin1[0] == fin0[0] and np.array_equal(fin1[1], fin0[1]) and fin1[2] == fin0[2]
        assert amd.cache_fill() == cf0
        plain.recognize(f)
        st, fw, fs = plain.final_result()
        assert st == fin1[0] and fs == fin1[2]
        if lm == "ngram":
            assert np.array_equal(fw, fin1[1])
    assert amd.prefetch_served() == 7                 # the queued inputs really came from the batch launch


@pytest.mark.parametrize("mode", ["strict", "fast", "batch"])
def test_gaussian_mixture_selection_two_pass(ref, tmp_path, monkeypatch, mode):
    """-gshmm / -gsnum.  The plain reference scores the selection model, keeps the
    nbest states per frame and answers every other state with its selection state's score
    (gms_state(), gms.c:394-412); the shim runs that stage on the device after the full scoring, so
    the search and the 2nd pass see the same numbers: same trellis, same pass-1 and final results."""
    monkeypatch.setenv("JAMD_STRICT_ORDER", "1" if mode == "strict" else "0")
    monkeypatch.setenv("JAMD_STREAM_CHUNK", "25")     # ignored under GMS: the selection carries state across frames
    if not pyoracle.REF_AMD_SO.exists():
        pytest.skip("oracle/_ref/libjref_amd.so not built")
    task = synth.make_triphone_task(tmp_path, seed=81, nword=120, nphone=10, S=160)
    gpath, _ = synth.make_gs_model(task, seed=81)
    args = ["-h", task["hmmdefs"], "-hlist", task["hmmlist"], "-v", task["dict"], "-nlr", task["arpa"],
            "-input", "htkparam", "-gprune", "none", "-b", "200", "-b2", "30", "-n", "1", "-s", "500", "-sepnum", "5",
            "-gshmm", str(gpath), "-gsnum", "6"]
    nogms = pyoracle.RefEngine(ref, args[:-4])
    plain = pyoracle.RefEngine(ref, args)
    amd = pyoracle.RefEngine(pyoracle.Ref(so=pyoracle.REF_AMD_SO), args)
    files = []
    for u in range(4):
        fr, _ = synth.make_utterance(task, nwords=3 + u, seed=8100 + u)
        files.append(tmp_path / f"u{u}.mfc")
        synth.write_htk_param(files[-1], fr)
    if mode == "batch":
        amd.prefetch(files)
    changed = 0
    for f in files:
        tr0, (w0, s0) = plain.recognize(f)
        fin0 = plain.final_result()
        tr1, (w1, s1) = amd.recognize(f)
        fin1 = amd.final_result()
        assert np.array_equal(w1, w0) and s1 == s0
        assert fin1[0] == fin0[0] and np.array_equal(fin1[1], fin0[1]) and fin1[2] == fin0[2]
        if mode == "strict":
            for k in tr0:
                assert np.array_equal(tr1[k], tr0[k]), k
        else:
            assert_canonical_close(tr1, tr0, max_diff=8)
        _, (_, s2) = nogms.recognize(f)
        changed += s2 != s0
    assert changed > 0                                # the selection really changed what the search saw
    if mode == "batch":
        assert amd.prefetch_served() == len(files)


@pytest.mark.parametrize("so", ["amd", "o"])
def test_verification_gmm_on_device(ref, tmp_path, monkeypatch, so):
    """-gmm / -gmmnum / -gmmreject.  Both shimmed builds also wrap gmm.c's entry points
    (julius_amd/shim/jamd_gmm_wrap.c): the frames gmm_proceed() would score one by one are scored on
    the device at gmm_end().  gc->gmm_score[], the winner, its confidence, gmm_valid_input() and the
    final status -- including a rejected input, which never reaches the 2nd pass -- equal the plain
    reference's."""
    monkeypatch.setenv("JAMD_STRICT_ORDER", "0")
    monkeypatch.setenv("JAMD_STREAM_CHUNK", "0")
    lib_so = pyoracle.REF_AMD_SO if so == "amd" else pyoracle.REF_O_SO
    if not lib_so.exists():
        pytest.skip(f"{lib_so} not built")
    task = synth.make_triphone_task(tmp_path, seed=85, nword=120, nphone=10, S=160)
    gpath, _, names = synth.make_rejection_gmm(tmp_path, task["model"]["centre"], seed=85, M=24, null_frac=0.05)
    base = ["-h", task["hmmdefs"], "-hlist", task["hmmlist"], "-v", task["dict"], "-nlr", task["arpa"],
            "-input", "htkparam", "-gprune", "none", "-b", "150", "-b2", "30", "-n", "1", "-s", "500", "-sepnum", "5",
            "-gmm", str(gpath), "-gmmnum", "6"]
    files = []
    for u in range(3):
        fr, _ = synth.make_utterance(task, nwords=3 + u, seed=8500 + u)
        files.append((tmp_path / f"u{u}.mfc", len(fr)))
        synth.write_htk_param(files[-1][0], fr)
    probe = pyoracle.RefEngine(ref, base)
    probe.recognize(files[0][0])
    winner_name = names[::-1][probe.gmm_result()[1]]           # gmm->start lists the models in reverse file order
    other = [n for n in names if n != winner_name][0]
    total = 0
    for reject in (other, winner_name):                        # accepted inputs, then rejected ones
        args = base + ["-gmmreject", reject]
        plain = pyoracle.RefEngine(ref, args)
        dev = pyoracle.RefEngine(pyoracle.Ref(so=lib_so), args)
        assert plain.gmm_device_frames() == -1
        total = 0
        for f, n in files:
            plain.recognize(f)
            want, fin0 = plain.gmm_result(), plain.final_result()
            dev.recognize(f)
            got, fin1 = dev.gmm_result(), dev.final_result()
            total += n
            assert np.array_equal(got[0], want[0]) and got[1:] == want[1:]
            assert fin1[0] == fin0[0] and np.array_equal(fin1[1], fin0[1]) and fin1[2] == fin0[2]
            assert dev.gmm_device_frames() == total            # every frame was scored on the device
        if reject == winner_name:
            assert fin0[0] < 0 and not want[3]                 # J_RESULT_STATUS_REJECT_GMM


@pytest.mark.parametrize("lm,order", [("ngram", "exact"), ("grammar", "exact"), ("ngram", "chunks"), ("grammar", "strict")])
def test_multipath_two_pass_over_device_first_pass(ref, tmp_path, monkeypatch, lm, order):
    """-multipath: the shim flattens the multipath lexicon and decodes it with the exact-order kernel's multipath frame
    (csrc/beam_exact_mp.h; in one piece and in 16-frame pieces) or, asked to, in strict order; trellis, pass-1 and final
    results as the plain reference."""
    monkeypatch.setenv("JAMD_STRICT_ORDER", "1" if order == "strict" else "0")
    monkeypatch.setenv("JAMD_STREAM_CHUNK", "16" if order == "chunks" else "0")
    if not pyoracle.REF_AMD_SO.exists():
        pytest.skip("oracle/_ref/libjref_amd.so not built")
    if lm == "ngram":
        task = synth.make_triphone_task(tmp_path, seed=87, nword=120, nphone=10, S=160)
        args = ["-h", task["hmmdefs"], "-hlist", task["hmmlist"], "-v", task["dict"], "-nlr", task["arpa"],
                "-input", "htkparam", "-gprune", "none", "-b", "200", "-b2", "30", "-n", "1", "-s", "500", "-sepnum", "5", "-multipath"]
        utts = [synth.make_utterance(task, nwords=3 + u, seed=8700 + u)[0] for u in range(3)]
    else:
        task = synth.make_triphone_grammar(synth.make_triphone_task(tmp_path, seed=88, nword=90, nphone=10, S=160), ncat=3, seed=88)
        args = ["-h", task["hmmdefs"], "-hlist", task["hmmlist"], "-dfa", task["dfa"], "-v", task["gdict"],
                "-input", "htkparam", "-gprune", "none", "-b", "200", "-penalty1", "-2.0", "-b2", "30", "-n", "1", "-s", "500", "-multipath"]
        utts = [synth.make_triphone_grammar_utterance(task, nwords=2 + u, seed=8800 + u)[0] for u in range(3)]
    plain = pyoracle.RefEngine(ref, args)
    amd = pyoracle.RefEngine(pyoracle.Ref(so=pyoracle.REF_AMD_SO), args)
    assert plain.multipath == 1
    for fr in utts:
        synth.write_htk_param(tmp_path / "u.mfc", fr)
        tr0, (w0, s0) = plain.recognize(tmp_path / "u.mfc")
        fin0 = plain.final_result()
        tr1, (w1, s1) = amd.recognize(tmp_path / "u.mfc")
        fin1 = amd.final_result()
        for k in tr0:
            assert np.array_equal(tr1[k], tr0[k]), k
        assert np.array_equal(w1, w0) and s1 == s0
        assert fin1[0] == fin0[0] and np.array_equal(fin1[1], fin0[1]) and fin1[2] == fin0[2]


def test_outprob_vector_input_over_device_first_pass(ref, oracle, tmp_path, monkeypatch):
    """`-input outprob`: the input file already holds the [T][S] state scores (what -outprobout
    writes); the shim hands them to the device search unscored and Julius' 2nd pass reads them from
    the parameter itself.  Same trellis (strict order) and results as the plain reference."""
    monkeypatch.setenv("JAMD_STRICT_ORDER", "1")
    monkeypatch.setenv("JAMD_STREAM_CHUNK", "0")
    if not pyoracle.REF_AMD_SO.exists():
        pytest.skip("oracle/_ref/libjref_amd.so not built")
    task = synth.make_triphone_task(tmp_path, seed=81, nword=100, nphone=10, S=160)
    args = ["-h", task["hmmdefs"], "-hlist", task["hmmlist"], "-v", task["dict"], "-nlr", task["arpa"],
            "-input", "outprob", "-b", "150", "-b2", "30", "-n", "1", "-s", "500", "-sepnum", "5"]
    am = ref.am_load(task["hmmdefs"], task["hmmlist"]).export()
    plain = pyoracle.RefEngine(ref, args)
    amd = pyoracle.RefEngine(pyoracle.Ref(so=pyoracle.REF_AMD_SO), args)
    for u in range(3):
        fr, _ = synth.make_utterance(task, nwords=3 + u, seed=8100 + u)
        synth.write_htk_param(tmp_path / "u.prob", oracle.gmm_outprob(am, fr), parmkind=synth.PARM_USER)
        tr0, (w0, s0) = plain.recognize(tmp_path / "u.prob")
        st0, f0, fs0 = plain.final_result()
        tr1, (w1, s1) = amd.recognize(tmp_path / "u.prob")
        st1, f1, fs1 = amd.final_result()
        assert st1 == st0 and np.array_equal(w1, w0) and s1 == s0
        assert np.array_equal(f1, f0) and fs1 == fs0
        for k in tr0:
            assert np.array_equal(tr1[k], tr0[k]), k


@pytest.mark.parametrize("kind", ["triphone", "c1"])
def test_official_plugin_slot(tmp_path, kind):
    """Boundary P: the UNMODIFIED reference (libjref.so, nothing relinked) dlopen()s
    oracle/_ref/plugin/jamd_calcmix.jpi through `-plugindir ... -gprune jamd`; its Gaussian
    computation slot is then served from device-computed per-Gaussian scores.  Same word trellis,
    pass-1 and final results as the reference's own gprune_none; the counters prove the slot ran.
    Runs in a subprocess because the reference must be loaded with its symbols visible to the plugin."""
    import subprocess, sys, textwrap
    if not (pyoracle.PLUGIN_DIR / "jamd_calcmix.jpi").exists():
        pytest.skip("oracle/_ref/plugin/jamd_calcmix.jpi not built")
    code = textwrap.dedent(f"""
        import sys, ctypes, numpy as np
        sys.path.insert(0, {str(pyoracle.HERE.parent)!r}); sys.path.insert(0, {str(pyoracle.HERE.parent / 'tests')!r})
        from pathlib import Path
        from julius_amd import synth
        from oracle import pyoracle
        tmp = Path({str(tmp_path)!r})
        ref = pyoracle.Ref(global_symbols=True)
        if {kind!r} == "triphone":
            task = synth.make_triphone_task(tmp, seed=95, nword=100, nphone=10, S=160)
            base = ["-h", task["hmmdefs"], "-hlist", task["hmmlist"], "-v", task["dict"], "-nlr", task["arpa"],
                    "-input", "htkparam", "-b", "150", "-b2", "30", "-n", "1", "-s", "500", "-sepnum", "5"]
            mk = lambda u: synth.make_utterance(task, nwords=3 + u, seed=9500 + u)[0]
        else:       # BASELINE configs[0] shape: tied-mixture monophones + 100-word grammar (the slot is entered per codebook)
            task = synth.make_grammar_task(tmp, seed=96)
            base = ["-h", task["hmmdefs"], "-dfa", task["dfa"], "-v", task["dict"], "-input", "htkparam", "-b", "200"]
            mk = lambda u: synth.make_grammar_utterance(task, nwords=3 + u, seed=9600 + u)[0]
        plain = pyoracle.RefEngine(ref, base + ["-gprune", "none"])
        plug = pyoracle.RefEngine(ref, ["-plugindir", {str(pyoracle.PLUGIN_DIR)!r}] + base + ["-gprune", "jamd"])
        jpi = ctypes.CDLL({str(pyoracle.PLUGIN_DIR / 'jamd_calcmix.jpi')!r})
        jpi.jamd_calcmix_calls.restype = ctypes.c_long; jpi.jamd_calcmix_fills.restype = ctypes.c_long
        for u in range(3):
            synth.write_htk_param(tmp / "u.mfc", mk(u))
            tr0, (w0, s0) = plain.recognize(tmp / "u.mfc"); st0, f0, fs0 = plain.final_result()
            tr1, (w1, s1) = plug.recognize(tmp / "u.mfc"); st1, f1, fs1 = plug.final_result()
            assert st1 == st0 and np.array_equal(w1, w0) and s1 == s0 and np.array_equal(f1, f0) and fs1 == fs0
            for k in tr0:
                assert np.array_equal(tr1[k], tr0[k]), k
        assert jpi.jamd_calcmix_fills() == 3 and jpi.jamd_calcmix_calls() > 500, (jpi.jamd_calcmix_fills(), jpi.jamd_calcmix_calls())
        print("PLUGIN_OK", jpi.jamd_calcmix_calls(), jpi.jamd_calcmix_fills())
    """)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert "PLUGIN_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


def test_per_gaussian_scores(engine):
    """jamd_gmm_dens_*: the values the plugin slot hands to calc_mix(), against an fp32 restatement
    of compute_g_base() (gprune_none.c:59-82) in numpy, bit for bit."""
    from julius_amd import lib
    model = synth.make_gmm(S=40, M=5, D=39, seed=7, ragged=True, null_frac=0.1)
    fr = synth.make_frames(model, T=70, seed=2)
    got = lib.Gmm(engine, model).dens_host(fr)
    dens = model["ent_dens"]
    acc = np.where(dens >= 0, model["gconst"][np.maximum(dens, 0)], np.float32(0))[None, :].repeat(len(fr), 0).astype(np.float32)
    for d in range(fr.shape[1]):
        x = (fr[:, None, d] - model["mean"][np.maximum(dens, 0), d][None, :]).astype(np.float32)
        x = (x * x).astype(np.float32)
        x = (x * model["ivar"][np.maximum(dens, 0), d][None, :]).astype(np.float32)
        acc = (acc + x).astype(np.float32)
    want = (acc * np.float32(-0.5)).astype(np.float32)
    want[:, dens < 0] = np.float32(-1000000.0)
    assert np.array_equal(got, want)
