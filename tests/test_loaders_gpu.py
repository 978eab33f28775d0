"""GPU: the library's own readers of on-disk models (jamd_gmm_load / jamd_lexicon_load /
jamd_dnn_load) give the same device models as the descriptor path: the golden
outputs of the compiled reference are reproduced bit for bit from files alone."""
import numpy as np
import pytest

from beamutil import assert_trellis_equal, load_beam_golden
from conftest import GOLDEN
from julius_amd import lexblob, lib, synth

pytestmark = pytest.mark.gpu


def _model(z):
    return dict(mean=z["mean"], ivar=z["ivar"], gconst=z["gconst"], st_off=z["st_off"], ent_dens=z["ent_dens"],
                ent_logw=z["ent_logw"], st_book=z["st_book"] if "st_book" in z.files else None,
                nbook=int(z["nbook"]) if "nbook" in z.files else 0, nstream=1)


def test_gmm_blob(engine, tmp_path):
    z = np.load(GOLDEN / "gmm_plain_none.npz")
    lexblob.save_gmm(_model(z), tmp_path / "am.blob")
    gm = lib.Gmm.from_file(engine, tmp_path / "am.blob")
    assert (gm.S, gm.D) == (len(z["st_off"]) - 1, z["mean"].shape[1])
    assert np.array_equal(gm.outprob_host(z["frames"]), z["out"])


def test_tied_mixture_blob(engine, tmp_path):
    z = np.load(GOLDEN / "gmm_tied.npz")
    lexblob.save_gmm(_model(z), tmp_path / "am.blob")
    assert lexblob.load_gmm(tmp_path / "am.blob")["nbook"] == int(z["nbook"])      # python reader agrees on the format
    gm = lib.Gmm.from_file(engine, tmp_path / "am.blob", lib.GPRUNE_SAFE, 2)
    assert np.array_equal(gm.outprob_host(z["frames"]), z["out_safe2"])


@pytest.mark.parametrize("name", ["beam_rank.npz", "beam_grammar.npz"])
def test_lexicon_blob(engine, oracle, tmp_path, name):
    g = load_beam_golden(name)
    lexblob.save(g["lex"], tmp_path / "lex.blob")
    lx = lib.Lexicon.from_file(engine, tmp_path / "lex.blob")
    bm = lib.Beam(engine, lx, g["beam_width"], g["score_pruning_width"], max_utts=len(g["utts"]))
    bm.set_strict_order(True)                                  # exact, no tie caveat
    res, tre = bm.pass1_host([oracle.gmm_outprob(g["am"], u["frames"]) for u in g["utts"]])
    for r, atoms, u in zip(res, tre, g["utts"]):
        assert r.status == 0 and r.score == u["score"] and np.array_equal(np.array(r.wseq[:r.wnum]), u["wseq"])
        assert_trellis_equal(atoms, u["trellis"])


def test_dnn_from_dnnconf(engine, tmp_path):
    """The reference's own on-disk DNN format: dnnconf + .npy + prior list."""
    z = np.load(GOLDEN / "dnn_small.npz")
    dnn = synth.make_dnn(dims=(48, 64, 64, 64, 40), seed=51)   # the network tests/make_golden.py wrote for the reference
    assert all(np.array_equal(dnn["w"][l], z[f"w{l}"]) for l in range(4))
    for l in range(4):
        synth.write_npy(tmp_path / f"W{l}.npy", dnn["w"][l])
        synth.write_npy(tmp_path / f"b{l}.npy", np.asarray(dnn["b"][l]).reshape(-1, 1))
    with open(tmp_path / "prior", "w") as f:
        for i, v in enumerate(dnn["prior_lin"]):
            f.write(f"{i} {float(v):.9e}\n")
    (tmp_path / "dnn.conf").write_text(
        "feature_type USER\nfeature_len 48\ncontext_len 1\ninput_nodes 48\noutput_nodes 40\nhidden_nodes 64\n"
        "hidden_layers 3\nW1 W0.npy\nW2 W1.npy\nW3 W2.npy\nB1 b0.npy\nB2 b1.npy\nB3 b2.npy\noutput_W W3.npy\n"
        "output_B b3.npy\nstate_prior prior   # relative to this file\nstate_prior_factor 1.0\nstate_prior_log10nize yes\nnum_threads 1\n")
    net = lib.Dnn.from_dnnconf(engine, tmp_path / "dnn.conf")
    assert (net.D, net.S) == (48, 40)
    assert np.array_equal(net.outprob_host(z["frames"]), z["out"])


def test_selection_model_blob(engine, tmp_path):
    """jamd_gms_load(): the selection model and its state map from a file; same outputs as the
    committed reference fixture."""
    z = np.load(GOLDEN / "gms.npz")
    sub = lambda p: {k[len(p):]: z[k] for k in z.files if k.startswith(p)}
    lexblob.save_gmm(sub("gs_"), tmp_path / "m.gms", state2gs=z["state2gs"], nbest=4)
    back = lexblob.load_gmm(tmp_path / "m.gms")
    assert back["nbest"] == 4 and np.array_equal(back["state2gs"], z["state2gs"])
    stage = lib.Gms.from_file(engine, tmp_path / "m.gms", veclen=z["frames"].shape[1])
    assert stage.S == len(z["state2gs"])
    real = lib.Gmm(engine, sub("full_")).outprob_host(z["frames"])
    used = z["state2gs"] >= 0
    assert np.array_equal(stage.apply_host(z["frames"], real, z["utt_off"])[:, used], z["out_4"][:, used])
    lexblob.save_gmm(sub("gs_"), tmp_path / "plain.am")
    with pytest.raises(lib.JamdError):                 # an acoustic-model blob is not a selection model
        lib.Gms.from_file(engine, tmp_path / "plain.am", veclen=39)


def test_loaders_report_bad_files(engine, tmp_path):
    (tmp_path / "junk").write_bytes(b"not a blob at all")
    for call in (lambda: lib.Gmm.from_file(engine, tmp_path / "junk"), lambda: lib.Lexicon.from_file(engine, tmp_path / "junk"),
                 lambda: lib.Dnn.from_dnnconf(engine, tmp_path / "missing.conf")):
        with pytest.raises(lib.JamdError):
            call()


def test_standalone_c_driver(oracle, tmp_path):
    """julius_amd/jamd_batch (C, links only libjulius_amd.so): model blob + lexicon blob + a list of
    HTK parameter files in, the reference's pass-1 sentences and scores out."""
    import subprocess
    exe = lib._PKG / "jamd_batch"
    assert exe.exists(), "build it with make -C julius_amd/csrc"
    g = load_beam_golden("beam_rank.npz")
    lexblob.save_gmm(g["am"], tmp_path / "am.blob")
    lexblob.save(g["lex"], tmp_path / "lex.blob")
    names = []
    for u, utt in enumerate(g["utts"]):
        names.append(str(tmp_path / f"u{u}.mfc"))
        synth.write_htk_param(names[-1], utt["frames"])
    (tmp_path / "list").write_text("\n".join(names) + "\n")
    out = subprocess.run([str(exe), "-am", str(tmp_path / "am.blob"), "-lex", str(tmp_path / "lex.blob"), "-filelist",
                          str(tmp_path / "list"), "-b", str(g["beam_width"]), "-strict"],
                         check=True, capture_output=True, text=True).stdout.strip().splitlines()
    assert len(out) == len(g["utts"])
    for line, name, utt in zip(out, names, g["utts"]):
        f = line.split(" ", 3)
        assert f[0] == name and f[1] == "status=0"
        assert np.float32(float(f[2].split("=")[1])) == np.float32(utt["score"])
        assert [int(x) for x in f[3].split("=", 1)[1].split()] == list(utt["wseq"])
    # several launches (reading + upload of launch k+1 under the kernels of launch k, two chunk buffers in turn), default
    # order mode: the same lines
    for per_launch in ("1", "2"):
        again = subprocess.run([str(exe), "-am", str(tmp_path / "am.blob"), "-lex", str(tmp_path / "lex.blob"), "-filelist",
                                str(tmp_path / "list"), "-b", str(g["beam_width"]), "-launch", per_launch],
                               check=True, capture_output=True, text=True).stdout.strip().splitlines()
        assert again == out
    # -time: the same lines on stdout, and one JSON record per device on stderr (the product's own clock: models up ->
    # last result line; what bench.py's `batch` entries report); launches grow the pinned staging buffers (1 then 2 files)
    tm = subprocess.run([str(exe), "-am", str(tmp_path / "am.blob"), "-lex", str(tmp_path / "lex.blob"), "-filelist",
                         str(tmp_path / "list"), "-b", str(g["beam_width"]), "-launch", "2", "-time"], check=True, capture_output=True, text=True)
    assert tm.stdout.strip().splitlines() == out
    import json
    rec = [json.loads(x)["jamd_batch_time"] for x in tm.stderr.splitlines() if x.startswith('{"jamd_batch_time"')]
    assert len(rec) == 1 and rec[0]["utts"] == len(names) and rec[0]["launches"] == (len(names) + 1) // 2
    assert rec[0]["frames"] == sum(len(u["frames"]) for u in g["utts"]) and rec[0]["decode_s"] > 0 and rec[0]["h2d_bytes"] == 4 * 39 * rec[0]["frames"]
    # BASELINE configs[4] inside one process: one host thread + engine + work area per listed device (here the same
    # device twice and three times), utterances dealt round-robin, result lines merged in file-list order
    for devices in ("0,0", "0-0,0,0"):
        merged = subprocess.run([str(exe), "-am", str(tmp_path / "am.blob"), "-lex", str(tmp_path / "lex.blob"), "-filelist",
                                 str(tmp_path / "list"), "-b", str(g["beam_width"]), "-devices", devices],
                                check=True, capture_output=True, text=True).stdout.strip().splitlines()
        assert merged == out
    # and combined with process-level shards: 2 processes x 2 device threads = 4 shards
    lines = {}
    for r in range(2):
        part = subprocess.run([str(exe), "-am", str(tmp_path / "am.blob"), "-lex", str(tmp_path / "lex.blob"), "-filelist",
                               str(tmp_path / "list"), "-b", str(g["beam_width"]), "-devices", "0,0", "-shard", str(r), "2"],
                              check=True, capture_output=True, text=True).stdout.strip().splitlines()
        for ln in part:
            lines[ln.split(" ", 1)[0]] = ln
    assert [lines[n] for n in names] == out


@pytest.mark.parametrize("gms,mp", [(False, False), (True, False), (False, True)])
def test_export_then_standalone_batch(ref, tmp_path, gms, mp):
    """The whole no-Julius-at-run-time flow: jamd_export (Julius' loaders -> files), then jamd_batch
    (C, C ABI only) over a file list; pass-1 sentences and scores equal the plain reference's --
    also with Gaussian mixture selection (-gshmm -> PREFIX.gms -> jamd_batch -gms), and with -multipath (the
    exported lexicon is the reference's multipath lexicon; jamd_batch in its default order = the multipath frame)."""
    import subprocess
    from oracle import pyoracle
    export = pyoracle.HERE.parent / "julius_amd" / "jamd_export"
    exe = lib._PKG / "jamd_batch"
    if not export.exists():
        pytest.skip("julius_amd/jamd_export not built")
    task = synth.make_triphone_task(tmp_path, seed=93, nword=120, nphone=10, S=160)
    args = [str(a) for a in ["-h", task["hmmdefs"], "-hlist", task["hmmlist"], "-v", task["dict"], "-nlr", task["arpa"],
                             "-input", "htkparam", "-gprune", "safe", "-tmix", "3", "-b", "150", "-sepnum", "5", "-1pass"]]
    if gms:
        args += ["-gshmm", str(synth.make_gs_model(task, seed=93)[0]), "-gsnum", "7"]
    if mp:
        args += ["-multipath"]
    subprocess.run([str(export)] + args + ["-jamdout", str(tmp_path / "m")], check=True, capture_output=True)
    assert (tmp_path / "m.gms").exists() == gms
    eng = pyoracle.RefEngine(ref, args)
    names, want = [], []
    for u in range(5):
        fr, _ = synth.make_utterance(task, nwords=2 + u, seed=9300 + u)
        names.append(str(tmp_path / f"u{u}.mfc"))
        synth.write_htk_param(names[-1], fr)
        _, p1 = eng.recognize(names[-1])
        want.append(p1)
    (tmp_path / "list").write_text("\n".join(names) + "\n")
    out = subprocess.run([str(exe), "-am", str(tmp_path / "m.am"), "-lex", str(tmp_path / "m.lex"), "-filelist",
                          str(tmp_path / "list"), "-b", "150", "-gprune", "safe", "3"] + ([] if mp else ["-strict"]) +
                         (["-gms", str(tmp_path / "m.gms")] if gms else []),
                         check=True, capture_output=True, text=True).stdout.strip().splitlines()
    assert len(out) == len(names)
    for line, (wseq, score) in zip(out, want):
        f = line.split(" ", 3)
        assert f[1] == "status=0" and np.float32(float(f[2].split("=")[1])) == np.float32(score)
        assert [int(x) for x in f[3].split("=", 1)[1].split()] == list(wseq)


def test_export_then_standalone_batch_with_verification(ref, tmp_path):
    """jamd_export writes PREFIX.rej for a -gmm configuration; jamd_batch -rej scores the verification
    GMMs of every input on the device and prints gmm_end()'s verdict: winner, confidence, accepted --
    as the plain reference decides (gc->gmm_score[], gmm_max_cm, gmm_valid_input())."""
    import subprocess
    from oracle import pyoracle
    export = pyoracle.HERE.parent / "julius_amd" / "jamd_export"
    exe = lib._PKG / "jamd_batch"
    if not export.exists():
        pytest.skip("julius_amd/jamd_export not built")
    task = synth.make_triphone_task(tmp_path, seed=95, nword=120, nphone=10, S=160)
    gpath, _, names = synth.make_rejection_gmm(tmp_path, task["model"]["centre"], seed=95, M=24, null_frac=0.05)
    base = [str(a) for a in ["-h", task["hmmdefs"], "-hlist", task["hmmlist"], "-v", task["dict"], "-nlr", task["arpa"],
                             "-input", "htkparam", "-gprune", "none", "-b", "150", "-sepnum", "5", "-1pass",
                             "-gmm", gpath, "-gmmnum", "6"]]
    files = []
    for u in range(4):
        fr, _ = synth.make_utterance(task, nwords=2 + u, seed=9500 + u)
        files.append(str(tmp_path / f"u{u}.mfc"))
        synth.write_htk_param(files[-1], fr)
    (tmp_path / "list").write_text("\n".join(files) + "\n")
    probe = pyoracle.RefEngine(ref, base)
    probe.recognize(files[0])
    rev = names[::-1]
    winner_name = rev[probe.gmm_result()[1]]
    for reject in ([n for n in names if n != winner_name][0], winner_name):
        args = base + ["-gmmreject", reject]
        subprocess.run([str(export)] + args + ["-jamdout", str(tmp_path / "m")], check=True, capture_output=True)
        eng = pyoracle.RefEngine(ref, args)
        out = subprocess.run([str(exe), "-am", str(tmp_path / "m.am"), "-lex", str(tmp_path / "m.lex"), "-rej", str(tmp_path / "m.rej"),
                              "-filelist", str(tmp_path / "list"), "-b", "150", "-strict"],
                             check=True, capture_output=True, text=True).stdout.strip().splitlines()
        assert len(out) == len(files)
        for line, f in zip(out, files):
            eng.recognize(f)
            sums, win, cm, valid, _ = eng.gmm_result()
            kv = dict(x.split("=", 1) for x in line.split(" ") if "=" in x)
            assert kv["gmm"] == rev[win] and int(kv["accepted"]) == int(valid)
            assert np.float32(float(kv["gmmscore"])) == sums[win] and np.float32(float(kv["cm"])) == np.float32(cm)
