"""Seeded synthetic models and inputs for tests and bench.py.

Everything here is *data preparation*: numpy restatements of what the
reference's model loader leaves in memory (inverse variances, gconst, ln
weights), plus writers for the HTK text/binary formats so that the very same
model can be loaded by the compiled reference (oracle/_ref) for parity pinning.

Reference formulas followed:
  ivar   = 1.0/var in double, stored float   libsent/src/hmminfo/rdhmmdef.c:162-173
  gconst = (float)(D*LOGTPI) then += (float)log(var_d) in float, d ascending
                                              libsent/src/hmminfo/rdhmmdef_dens.c:39-48
  ln w   = (float)log(w)                      libsent/src/hmminfo/rdhmmdef_mpdf.c:189
"""
from __future__ import annotations

import struct
from pathlib import Path

import numpy as np

LOGTPI = 1.83787706640935  # libsent/include/sent/stddefs.h:105
MFCC_E_D_A = 6 | 0x40 | 0x100 | 0x200  # htk_defs: MFCC + _E + _D + _A
PARM_USER = 9


def gconst_of(var: np.ndarray) -> np.ndarray:
    """update_gconst(): sequential float32 accumulation over dimensions."""
    var = np.asarray(var, dtype=np.float32)
    D = var.shape[-1]
    g = np.full(var.shape[:-1], np.float32(D * LOGTPI), dtype=np.float32)
    for d in range(D):
        g = (g + np.log(var[..., d].astype(np.float64)).astype(np.float32)).astype(np.float32)
    return g


def make_gmm(S=3000, M=16, D=39, seed=0, ragged=False, null_frac=0.0):
    """Plain (non tied-mixture) GMM state pool: S states x M diagonal Gaussians.

    ragged=True gives every state its own mixture count in [1, M];
    null_frac>0 replaces that fraction of entries by NULL densities.
    Returns the flat layout of include/julius_amd.h::jamd_gmm_desc plus the raw
    `var` and `weight` arrays used by the HTK writer.
    """
    rng = np.random.default_rng(seed)
    if ragged:
        nmix = rng.integers(1, M + 1, size=S)
    else:
        nmix = np.full(S, M, dtype=np.int64)
    st_off = np.zeros(S + 1, dtype=np.int32)
    st_off[1:] = np.cumsum(nmix)
    E = int(st_off[-1])
    # state centres spread out so different states win on different frames
    centre = rng.normal(0.0, 1.0, size=(S, D)).astype(np.float32)
    owner = np.repeat(np.arange(S), nmix)
    mean = (centre[owner] + rng.normal(0.0, 0.5, size=(E, D))).astype(np.float32)
    var = rng.uniform(0.5, 2.0, size=(E, D)).astype(np.float32)
    weight = np.empty(E, dtype=np.float64)
    for s in range(S):
        n = int(nmix[s])
        w = rng.dirichlet(np.full(n, 2.0)) if n > 1 else np.ones(1)
        # weights as they would be parsed back from a 6-significant-digit text file
        weight[st_off[s]:st_off[s + 1]] = np.array([float(f"{x:.6e}") for x in np.maximum(w, 1e-5)])
    ent_dens = np.arange(E, dtype=np.int32)
    if null_frac > 0:
        kill = rng.random(E) < null_frac
        kill[st_off[:-1]] = False  # keep the first mixture of every state
        ent_dens[kill] = -1
    model = dict(
        mean=mean,
        var=var,
        ivar=(1.0 / var.astype(np.float64)).astype(np.float32),
        gconst=gconst_of(var),
        weight=weight,
        st_off=st_off,
        ent_dens=ent_dens,
        ent_logw=np.log(weight).astype(np.float32),
        st_book=None,
        nbook=0,
        nstream=1,
        centre=centre,
    )
    # a <Mixture> that is absent from the file leaves b[i]=NULL, bweight=LOG_ZERO
    # (libsent/src/hmminfo/rdhmmdef_mpdf.c:170-174)
    model["ent_logw"][ent_dens < 0] = np.float32(-1000000.0)
    return model


def make_tied_gmm(S=120, nbook=3, K=64, D=39, seed=0):
    """Tied-mixture model: nbook codebooks of K Gaussians, every state is a
    weight vector over one codebook (HTK <TMix>)."""
    rng = np.random.default_rng(seed)
    G = nbook * K
    mean = rng.normal(0.0, 1.5, size=(G, D)).astype(np.float32)
    var = rng.uniform(0.5, 2.0, size=(G, D)).astype(np.float32)
    st_book = rng.integers(0, nbook, size=S).astype(np.int32)
    st_book[:nbook] = np.arange(nbook)  # every codebook used
    st_off = (np.arange(S + 1) * K).astype(np.int32)
    ent_dens = np.concatenate([np.arange(b * K, (b + 1) * K) for b in st_book]).astype(np.int32)
    weight = np.empty(S * K, dtype=np.float64)
    for s in range(S):
        w = rng.dirichlet(np.full(K, 0.3))
        weight[s * K:(s + 1) * K] = np.array([float(f"{x:.6e}") for x in np.maximum(w, 1e-7)])
    return dict(
        mean=mean, var=var, ivar=(1.0 / var.astype(np.float64)).astype(np.float32),
        gconst=gconst_of(var), weight=weight, st_off=st_off, ent_dens=ent_dens,
        ent_logw=np.log(weight).astype(np.float32), st_book=st_book, nbook=nbook, nstream=1,
        book_size=K,
    )


def make_frames(model, T=1000, seed=1, noise=1.0):
    """Synthetic 'MFCC' frames: a random walk over state centres plus noise, so
    likelihood rankings vary over time (non-degenerate best paths)."""
    rng = np.random.default_rng(seed)
    D = model["mean"].shape[1]
    if "centre" in model and model["centre"] is not None:
        S = model["centre"].shape[0]
        seg = rng.integers(0, S, size=(T + 9) // 10)
        base = model["centre"][np.repeat(seg, 10)[:T]]
    else:
        base = np.zeros((T, D), dtype=np.float32)
    return (base + rng.normal(0.0, noise, size=(T, D))).astype(np.float32)


# --------------------------------------------------------------------- HTK I/O
def write_htk_param(path, frames: np.ndarray, parmkind=MFCC_E_D_A, samp_period=100000):
    """HTK parameter file: 12-byte big-endian header + big-endian float32 rows
    (reader: libsent/src/anlz/rdparam.c:110-170)."""
    frames = np.ascontiguousarray(frames, dtype=np.float32)
    T, D = frames.shape
    with open(path, "wb") as f:
        f.write(struct.pack(">iihh", T, samp_period, D * 4, parmkind))
        f.write(frames.astype(">f4").tobytes())


def read_htk_param(path):
    raw = Path(path).read_bytes()
    T, period, size, kind = struct.unpack(">iihh", raw[:12])
    D = size // 4
    return np.frombuffer(raw[12:12 + T * size], dtype=">f4").reshape(T, D).astype(np.float32), kind


def _vec(v):
    return " ".join(f"{float(x):.9e}" for x in v)


def write_hmmdefs(path, model, phones=None, trans=None, kind="MFCC_E_D_A", state_names=None, sp_state=None):
    """HTK ascii hmmdefs holding the model's states as ~s macros (state id =
    order of appearance) and 3-state left-to-right ~h models.

    phones: list of (name, (s1, s2, s3)) physical HMMs; default = monophones
    "p0".."pN" covering the state pool in order (S must be a multiple of 3).
    """
    S = len(model["st_off"]) - 1
    D = model["mean"].shape[1]
    tied = model.get("st_book") is not None and model.get("nbook", 0) > 0
    if phones is None:
        assert S % 3 == 0, "default topology needs S % 3 == 0"
        phones = [(f"p{i}", (3 * i, 3 * i + 1, 3 * i + 2)) for i in range(S // 3)]
    if trans is None:
        trans = np.array([[0, 1, 0, 0, 0], [0, .6, .4, 0, 0], [0, 0, .6, .4, 0],
                          [0, 0, 0, .7, .3], [0, 0, 0, 0, 0]], dtype=np.float64)
    L = []
    L.append(f"~o <STREAMINFO> 1 {D} <VECSIZE> {D} <NULLD> <{kind}> <DIAGC>")
    if tied:
        K = model["book_size"]
        for b in range(model["nbook"]):
            for k in range(K):
                g = b * K + k
                L.append(f'~m "book{b}_{k + 1}"')
                L.append(f"<MEAN> {D}\n {_vec(model['mean'][g])}")
                L.append(f"<VARIANCE> {D}\n {_vec(model['var'][g])}")
    for s in range(S):
        e0, e1 = int(model["st_off"][s]), int(model["st_off"][s + 1])
        name = state_names[s] if state_names else f"s{s}"
        L.append(f'~s "{name}"')
        if tied:
            b = int(model["st_book"][s])
            L.append(f"<NUMMIXES> {e1 - e0}")
            L.append(f"<TMIX> book{b}_ " + " ".join(f"{w:.6e}" for w in model["weight"][e0:e1]))
        else:
            L.append(f"<NUMMIXES> {e1 - e0}")
            for m, e in enumerate(range(e0, e1)):
                if model["ent_dens"][e] < 0:
                    continue  # a missing <MIXTURE> leaves a NULL density in the reference
                L.append(f"<MIXTURE> {m + 1} {model['weight'][e]:.6e}")
                L.append(f"<MEAN> {D}\n {_vec(model['mean'][model['ent_dens'][e]])}")
                L.append(f"<VARIANCE> {D}\n {_vec(model['var'][model['ent_dens'][e]])}")
    L.append('~t "t0"\n<TRANSP> 5')
    for row in trans:
        L.append(" " + " ".join(f"{x:.6e}" for x in row))
    for name, (a, b, c) in phones:
        sn = (lambda i: state_names[i]) if state_names else (lambda i: f"s{i}")
        L.append(f'~h "{name}"\n<BEGINHMM>\n<NUMSTATES> 5')
        L.append(f'<STATE> 2\n~s "{sn(a)}"\n<STATE> 3\n~s "{sn(b)}"\n<STATE> 4\n~s "{sn(c)}"')
        L.append('~t "t0"\n<ENDHMM>')
    if sp_state is not None:
        # a short-pause model the reference appends to every word with -iwsp (wchmm.c; multipath only): one emitting
        # state and an entry -> exit skip (a tee model)
        sn = state_names[sp_state] if state_names else f"s{sp_state}"
        L.append(f'~h "sp"\n<BEGINHMM>\n<NUMSTATES> 3\n<STATE> 2\n~s "{sn}"')
        L.append("<TRANSP> 3\n 0.000000e+00 6.000000e-01 4.000000e-01\n 0.000000e+00 7.000000e-01 3.000000e-01\n"
                 " 0.000000e+00 0.000000e+00 0.000000e+00\n<ENDHMM>")
    Path(path).write_text("\n".join(L) + "\n")
    return phones


# ------------------------------------------------------------------------ DNN
def make_dnn(dims=(528, 2048, 2048, 2048, 2048, 2048, 2048, 4000), seed=0):
    """Random-init DNN of the ENVR-v5.4 shape (BASELINE.json configs[3]):
    W[l] ~ N(0, 1/sqrt(in)) stored [out][in], b ~ N(0, 0.1), Dirichlet prior
    stored as log10 (state_prior_log10nize, calc_dnn.c:699-703)."""
    rng = np.random.default_rng(seed)
    w, b = [], []
    for l in range(len(dims) - 1):
        w.append((rng.standard_normal((dims[l + 1], dims[l])) / np.sqrt(dims[l])).astype(np.float32))
        b.append((0.1 * rng.standard_normal(dims[l + 1])).astype(np.float32))
    p = rng.dirichlet(np.full(dims[-1], 5.0)).astype(np.float32)
    prior = np.log10(p.astype(np.float64)).astype(np.float32)
    return dict(dims=np.asarray(dims, dtype=np.int32), w=w, b=b, prior=prior, prior_lin=p)


def _dnn_hidden(dnn, x):
    """The hidden layers of dnn_calc_outprob() (calc_dnn.c:837-845) in numpy: h = logistic(W h + b)."""
    h = np.asarray(x, dtype=np.float32)
    for w, b in zip(dnn["w"][:-1], dnn["b"][:-1]):
        h = h @ w.T + b
        h = (1.0 / (1.0 + np.exp(-np.clip(h, -8.0, 8.0)))).astype(np.float32)
    return h


def make_decodable_dnn(dims=(528, 2048, 2048, 2048, 2048, 2048, 2048, 4000), seed=0, gain=2.5, sharp=20.0):
    """A DNN of the ENVR-v5.4 shape whose posteriors are PEAKED on the state a frame was drawn from, without a
    training loop (there are no trained weights offline): random hidden layers, and an output layer that is the
    nearest-centroid classifier over the last hidden layer's activations of one centre input per state --
    logit_c(h) = a (z_c . (h - mu) - |z_c|^2 / 2) with z_c = h(centre_c) - mu, i.e. -a/2 |h - h(centre_c)|^2 + const(h),
    `a` chosen so that the mean squared distance between two centres is worth `sharp` log10 units.  A frame
    centre[c] + noise then scores state c near 0 and the other states tens of log10 units below, the way a trained
    acoustic model does: the first pass over these scores ends in a sentence (configs[3] with an input that decodes;
    README.md:104-127 of the reference shows the recipe on a real model).  `gain` scales the hidden weights
    (N(0, gain / sqrt(in))): at gain 1 six logistic layers contract every input onto nearly one point.
    Returns make_dnn()'s dict plus `centre[S][dims[0]]`."""
    rng = np.random.default_rng(seed)
    dims = [int(d) for d in dims]
    w, b = [], []
    for l in range(len(dims) - 2):
        w.append((gain * rng.standard_normal((dims[l + 1], dims[l])) / np.sqrt(dims[l])).astype(np.float32))
        b.append((0.1 * rng.standard_normal(dims[l + 1])).astype(np.float32))
    # hidden layers l >= 1 see logistic outputs (mean 1/2): centre their pre-activations so they stay in the steep part
    for l in range(1, len(w)):
        b[l] = (b[l] - 0.5 * w[l].sum(axis=1)).astype(np.float32)
    S = dims[-1]
    centre = rng.standard_normal((S, dims[0])).astype(np.float32)
    dnn = dict(dims=np.asarray(dims, dtype=np.int32), w=w + [None], b=b + [None])
    H = _dnn_hidden(dnn, centre).astype(np.float64)
    mu = H.mean(axis=0)
    Z = H - mu
    # mean squared distance between two centres in the last hidden layer = 2 * mean |z|^2 (the z are centred)
    msd = 2.0 * float((Z * Z).sum(axis=1).mean())
    a = sharp * np.log(10.0) * 2.0 / msd
    dnn["w"][-1] = (a * Z).astype(np.float32)
    dnn["b"][-1] = (-a * ((Z * Z).sum(axis=1) * 0.5 + Z @ mu)).astype(np.float32)
    p = rng.dirichlet(np.full(S, 5.0)).astype(np.float32)
    dnn.update(prior=np.log10(p.astype(np.float64)).astype(np.float32), prior_lin=p, centre=centre)
    return dnn


def word_state_path(task, ws, rng):
    """The emitting states of silB + words `ws` + silE through the task's own models: every phone becomes the logical
    triphone l-c+r of its neighbours (across word boundaries, silB / silE at the ends), mapped by the HMMList to its
    physical model's three states; a triphone the HMMList leaves out (make_triphone_task's defined_frac: the
    pseudo-phone sets of libsent/src/hmminfo/cdset.c serve it) takes some variant of its centre phone."""
    phys, logical = task["phys"], task["logical"]
    ph = ["silB"] + [p for _, pp in ws for p in pp] + ["silE"]
    seq = list(phys["silB"])
    for i in range(1, len(ph) - 1):
        name = logical.get(f"{ph[i - 1]}-{ph[i]}+{ph[i + 1]}")
        if name is None:
            var = sorted(k for k in phys if k.startswith(ph[i] + "_v"))
            name = var[int(rng.integers(0, len(var)))]
        seq += list(phys[name])
    return seq + list(phys["silE"])


def make_path_utterance(task, nwords=8, seed=0, frames_per_state=3, noise=0.6):
    """make_utterance() through the task's own triphone models (word_state_path): frames drawn around the centres of
    the states the random word sequence really passes, so that both passes of the reference recognise it."""
    rng = np.random.default_rng(seed)
    model = task["model"]
    ws = [task["words"][int(i)] for i in rng.integers(0, len(task["words"]), size=nwords)]
    st = np.repeat(np.array(word_state_path(task, ws, rng)), frames_per_state)
    fr = model["centre"][st] + rng.normal(0, noise, size=(len(st), model["mean"].shape[1]))
    return fr.astype(np.float32), [w for w, _ in ws]


def make_dnn_utterance(task, dnn, nwords=30, seed=0, frames_per_state=3, noise=0.5):
    """An utterance for the DNN-HMM task: frames in the network's INPUT space (one dims[0]-vector per frame, the
    form `-input htkparam` hands to dnn_calc_outprob(): parvec[t] is the spliced vector already, calc_dnn.c:800-803)
    drawn around the centres of the states of a random word sequence taken through the task's own triphone models
    (word_state_path), `frames_per_state` frames each.  Returns (frames, words)."""
    rng = np.random.default_rng(seed)
    ws = [task["words"][int(i)] for i in rng.integers(0, len(task["words"]), size=nwords)]
    st = np.repeat(np.array(word_state_path(task, ws, rng)), frames_per_state)
    fr = dnn["centre"][st] + rng.normal(0, noise, size=(len(st), dnn["centre"].shape[1]))
    return fr.astype(np.float32), [w for w, _ in ws]


def write_npy(path, a):
    """NPY v1 '<f4' C-order, the only form load_npy() accepts (calc_dnn.c:225-335)."""
    np.save(path, np.ascontiguousarray(a, dtype="<f4"))


def write_dnnconf(workdir, dnn, feature_len=None, context_len=1, num_threads=1):
    """The network in the reference's own files: W<l>.npy / b<l>.npy, the state prior list and the dnnconf that
    `-dnnconf` reads (libjulius/src/m_jconf.c:600-660, libsent/src/phmm/calc_dnn.c:528 dnn_setup).  All hidden
    layers share one width (reference limitation).  Returns the dnnconf path."""
    workdir = Path(workdir)
    dims = [int(x) for x in dnn["dims"]]
    nl = len(dims) - 1
    assert len(set(dims[1:-1])) == 1, "reference needs equal hidden widths"
    feature_len = feature_len or dims[0] // context_len
    assert feature_len * context_len == dims[0]
    for l in range(nl):
        write_npy(workdir / f"W{l}.npy", dnn["w"][l])
        write_npy(workdir / f"b{l}.npy", np.asarray(dnn["b"][l]).reshape(-1, 1))
    with open(workdir / "prior", "w") as f:
        for i, v in enumerate(dnn["prior_lin"]):
            f.write(f"{i} {float(v):.9e}\n")
    nh = nl - 1
    lines = [f"feature_type USER", f"feature_len {feature_len}", f"context_len {context_len}", f"input_nodes {dims[0]}",
             f"output_nodes {dims[-1]}", f"hidden_nodes {dims[1]}", f"hidden_layers {nh}"]
    lines += [f"W{l + 1} W{l}.npy" for l in range(nh)] + [f"B{l + 1} b{l}.npy" for l in range(nh)]
    lines += [f"output_W W{nh}.npy", f"output_B b{nh}.npy", "state_prior prior", "state_prior_factor 1.0",
              "state_prior_log10nize yes", f"num_threads {num_threads}"]
    (workdir / "dnn.conf").write_text("\n".join(lines) + "\n")
    return workdir / "dnn.conf"


# ---------------------------------------------------- triphone task (beam tests)
def make_triphone_task(workdir, nphone=8, S=120, M=4, D=39, nword=60, nvar=3, seed=0,
                       maxlen=5, nbigram_per_word=6, defined_frac=0.8, with_rl3=False, ntransparent=0, nunk=0, trans=None, sp=False,
                       ntee=0):
    """Write a complete synthetic recognition task the reference can load:
    tied-state triphone hmmdefs + HMMList, HTK dictionary with <s>/</s>, ARPA
    forward 2-gram (optionally a backward 3-gram).  Returns a dict of paths plus
    the GMM model.  Some logical triphones are deliberately left out of the
    HMMList so word-boundary nodes fall back to pseudo-phone state sets
    (libsent/src/hmminfo/cdset.c) and exercise outprob_cd()."""
    workdir = Path(workdir)
    workdir.mkdir(parents=True, exist_ok=True)
    rng = np.random.default_rng(seed)
    phones = [f"p{i}" for i in range(nphone)]
    sil = ["silB", "silE"]
    model = make_gmm(S=S, M=M, D=D, seed=seed + 1)
    # state pool partition: each centre phone owns a slice of the pool
    per = (S - 6) // nphone
    assert per >= 3
    phys = []          # (name, (s1,s2,s3))
    for ci, c in enumerate(phones):
        base = ci * per
        for v in range(nvar):
            st = tuple(int(base + rng.integers(0, per)) for _ in range(3))
            phys.append((f"{c}_v{v}", st))
    phys.append(("silB", (S - 6, S - 5, S - 4)))
    phys.append(("silE", (S - 3, S - 2, S - 1)))
    write_hmmdefs(workdir / "hmmdefs", model, phones=phys, trans=trans, sp_state=(S - 5) if sp else None)
    # logical triphones -> physical variants; contexts include silB/silE
    ctx = phones + sil
    lines = ["silB silB", "silE silE"] + (["sp sp"] if sp else [])      # -iwsp -spmodel sp
    for c in phones:
        for l in ctx:
            for r in ctx:
                if rng.random() > defined_frac:
                    continue          # undefined -> pseudo phone fallback at word edges
                v = int(rng.integers(0, nvar))
                lines.append(f"{l}-{c}+{r} {c}_v{v}")
        # word-internal triphones must exist for every (l, r) in phones: guarantee them
    have = {ln.split()[0] for ln in lines}
    for c in phones:
        for l in phones:
            for r in phones:
                name = f"{l}-{c}+{r}"
                if name not in have:
                    lines.append(f"{name} {c}_v{int(rng.integers(0, nvar))}")
    (workdir / "hmmlist").write_text("\n".join(lines) + "\n")
    # dictionary
    words = []
    seen = set()
    while len(words) < nword:
        n = int(rng.integers(1, maxlen + 1))
        ph = tuple(rng.choice(phones, size=n))
        if ph in seen:
            continue
        seen.add(ph)
        words.append((f"W{len(words):04d}", ph))
    # the first `ntransparent` words are written {transparent} (libsent/src/voca/voca_load_htkdict.c:463-468):
    # the LM context skips them
    dl = ["<s> [] silB", "</s> [] silE"] + [
        (f"{w} {{{w}}} " if i < ntransparent else f"{w} [{w}] ") + " ".join(ph) for i, (w, ph) in enumerate(words)]
    # `ntee` words made of the short-pause model only (the dictation kits' punctuation words, "sp"): with a tee `sp` the
    # word's head node reaches its tail along its own arcs (multipath lexicons; beam.c:2467-2510)
    tee_words = [f"T{i:02d}" for i in range(ntee)]
    assert not ntee or sp, "tee-only words need the sp model"
    dl += [f"{w} [{w}] sp" for w in tee_words]
    (workdir / "dict").write_text("\n".join(dl) + "\n")
    # forward 2-gram ARPA (log10), entries in 1-gram order.  With nunk > 0 the last `nunk`
    # dictionary words are left out of the LM and an <unk> entry is added: the reference maps
    # them to it and divides its probability among them (unk_num_log,
    # libsent/src/ngram/init_ngram.c; used by libsent/src/ngram/ngram_access.c:296-306)
    lm_words = [w for w, _ in words][:len(words) - nunk] if nunk > 0 else [w for w, _ in words]
    lm_words = lm_words + tee_words
    vocab = ["<s>", "</s>"] + (["<unk>"] if nunk > 0 else []) + lm_words
    V = len(vocab)
    uni = rng.dirichlet(np.full(V, 1.0))
    uni = np.log10(uni)
    bo = -rng.uniform(0.1, 1.0, size=V)
    big = []
    for i in range(V):
        if vocab[i] == "</s>":
            continue
        js = np.sort(rng.choice(np.arange(1, V), size=min(nbigram_per_word, V - 1), replace=False))
        ps = np.log10(rng.dirichlet(np.full(len(js), 1.0)) * 0.8)
        for j, p in zip(js, ps):
            big.append((i, int(j), float(p)))
    with open(workdir / "lm.arpa", "w") as f:
        f.write(f"\\data\\\nngram 1={V}\nngram 2={len(big)}\n\n\\1-grams:\n")
        for i in range(V):
            f.write(f"{uni[i]:.6f}\t{vocab[i]}\t{bo[i]:.6f}\n")
        f.write("\n\\2-grams:\n")
        for i, j, p in big:
            f.write(f"{p:.6f}\t{vocab[i]} {vocab[j]}\n")
        f.write("\n\\end\\\n")
    rl = None
    if with_rl3:
        # backward (RL) 3-gram for the reference's standard "-nlr 2-gram -nrl 3-gram" set-up: the
        # first pass then reads the forward 2-gram through bi_prob_additional() (RL index, LR
        # probabilities in the additional area; libsent/src/ngram/ngram_access.c:351).  Every
        # forward 2-gram (a b) needs its reversed tuple (b a) among the RL 2-grams
        # (ngram_read_arpa.c:305-318); tuples are sorted in 1-gram order.
        rl2 = sorted({(j, i) for i, j, _ in big})
        first2 = {}                          # i -> the first two k with (i, k) among the RL 2-grams, in their order
        for (i2, k) in rl2:
            lst = first2.setdefault(i2, [])
            if len(lst) < 2:
                lst.append(k)
        rl3 = []
        for (j, i) in rl2[::3]:
            rl3 += [(j, i, k) for k in first2.get(i, [])]
        rl3 = sorted(set(rl3))
        rl = workdir / "lm_rl.arpa"
        with open(rl, "w") as f:
            f.write(f"\\data\\\nngram 1={V}\nngram 2={len(rl2)}\nngram 3={len(rl3)}\n\n\\1-grams:\n")
            for i in range(V):
                f.write(f"{uni[i]:.6f}\t{vocab[i]}\t{-rng.uniform(0.1, 1.0):.6f}\n")
            f.write("\n\\2-grams:\n")
            for j, i in rl2:
                f.write(f"{-rng.uniform(0.2, 2.0):.6f}\t{vocab[j]} {vocab[i]}\t{-rng.uniform(0.1, 0.8):.6f}\n")
            f.write("\n\\3-grams:\n")
            for j, i, k in rl3:
                f.write(f"{-rng.uniform(0.2, 2.0):.6f}\t{vocab[j]} {vocab[i]} {vocab[k]}\n")
            f.write("\n\\end\\\n")
    return dict(dir=workdir, hmmdefs=workdir / "hmmdefs", hmmlist=workdir / "hmmlist", dict=workdir / "dict",
                arpa=workdir / "lm.arpa", arpa_rl=rl, model=model, words=words, vocab=vocab, phones=phones,
                phys=dict(phys), logical=dict(ln.split() for ln in lines), tee_words=tee_words)


def make_utterance(task, nwords=6, seed=0, frames_per_state=3, noise=0.7):
    """Frames that roughly follow a random word sequence through the task's
    models (so the best path is a real sentence, not noise)."""
    rng = np.random.default_rng(seed)
    model = task["model"]
    ws = [task["words"][int(i)] for i in rng.integers(0, len(task["words"]), size=nwords)]
    # walk: silB, words' phones (ignoring exact triphone identity: centre-phone states), silE
    S = len(model["st_off"]) - 1
    nphone = len(task["phones"])
    per = (S - 6) // nphone
    seq = [S - 6, S - 5, S - 4]
    for _, ph in ws:
        for p in ph:
            ci = task["phones"].index(p)
            base = ci * per
            seq += [int(base + rng.integers(0, per)) for _ in range(3)]
    seq += [S - 3, S - 2, S - 1]
    st = np.repeat(np.array(seq), frames_per_state)
    fr = model["centre"][st] + rng.normal(0, noise, size=(len(st), model["mean"].shape[1]))
    return fr.astype(np.float32), [w for w, _ in ws]


# ------------------------------------------------- synthetic tree lexicon (bench / scale tests)
def make_lexicon(nword=20000, nphone=40, S=3000, seed=0, minlen=2, maxlen=8, sepnum=150, nshort=20,
                 nbigram_per_word=20, defined_frac=0.9, lm_weight=8.0, lm_penalty=-2.0):
    """A seeded tree lexicon + 2-gram in the flat form of jamd_lexicon_desc, built
    directly (no reference code involved) with the same structural rules the
    reference's builder applies, so that a 20k-word task of BASELINE.json's shape
    exists on machines without the reference:

      * every phone is three left-to-right nodes (self 0.6 / next 0.4, exit 0.3);
      * words are merged into a prefix tree over their LOGICAL triphone names
        (head "a+b", inner "a-b+c", tail "b-c"), so word ends are always leaves
        (build_wchmm2(), libjulius/src/wchmm.c:1749; sharing test :417-430);
      * the `sepnum` most frequent words and the one-phone words stay outside the
        tree, as do <s> and </s>; <s> is not a cross-word target (wchmm.c:1585-1640);
      * 1-gram factoring ids: a node whose successor-word set differs from its
        predecessor's carries scid > 0 (one word: index into scword) or scid < 0
        (several: index into fscore = max 1-gram of the set)
        (libjulius/src/factoring_sub.c:345-468); roots with scid > 0 are the
        "isolated" ones (make_iwcache_index(), factoring_sub.c:719-735);
      * head phones are AS_RSET (row of the left-context table), tail phones
        AS_LSET (state set over the right contexts), one-phone words AS_LRSET
        (wchmm_add_word(), wchmm.c:1093-1128).

    State ids index a pool of S tied states (per centre phone and state position).
    """
    rng = np.random.default_rng(seed)
    per = S // nphone
    ploc = max(per // 3, 1)
    assert per >= 3
    nsil = 2
    W = nword + nsil                     # word 0 = <s>, word 1 = </s>
    SILB, SILE = nphone, nphone + 1      # context classes of the two silence words
    nlc = nphone + 2

    def tri(l, c, r, loc):
        return int(c * per + loc * ploc + ((l * 131 + r * 31 + 7) % ploc))

    # ---- vocabulary -------------------------------------------------------------
    seqs, seen = [], set()
    while len(seqs) < nword:
        n = 1 if len(seqs) < min(nshort, nphone // 2) else int(rng.integers(minlen, maxlen + 1))
        ph = tuple(int(x) for x in rng.integers(0, nphone, size=n))
        if ph in seen:
            continue
        seen.add(ph)
        seqs.append(ph)
    order = rng.permutation(nword)
    seqs = [seqs[i] for i in order]
    uni = np.log10(rng.dirichlet(np.full(W, 0.3)) + 1e-9).astype(np.float32)
    uni[0] = uni[1] = np.float32(np.log10(0.05))          # sentence delimiters are frequent
    bo = (-rng.uniform(0.1, 1.0, size=W)).astype(np.float32)
    rank = np.argsort(-uni[nsil:])
    separated = np.zeros(nword, bool)
    separated[rank[:sepnum]] = True
    for i, ph in enumerate(seqs):
        if len(ph) == 1:
            separated[i] = True

    # ---- nodes --------------------------------------------------------------------
    self_a, next_a, stend, scid, out_kind, out_id = [], [], [], [], [], []
    ac = {}                                   # node -> list of (to, a)
    LZ = np.float32(-1000000.0)
    A_SELF, A_NEXT = np.float32(np.log10(0.6)), np.float32(np.log10(0.4))
    A_SELF3, A_EXIT = np.float32(np.log10(0.7)), np.float32(np.log10(0.3))
    rows, row_list, sets, set_list = {}, [], {}, []

    def row_id(kind, key):
        k = (kind, key)
        if k not in rows:
            rows[k] = len(row_list)
            row_list.append(k)
        return rows[k]

    def set_id(key):
        if key not in sets:
            sets[key] = len(set_list)
            set_list.append(key)
        return sets[key]

    def add_phone(spec):
        """three nodes; returns first node id.  spec(loc) -> (out_kind, out_id)"""
        first = len(self_a)
        for loc in range(3):
            self_a.append(A_SELF if loc < 2 else A_SELF3)
            next_a.append(A_NEXT if loc < 2 else LZ)      # last state: linked by the caller
            stend.append(-1)
            scid.append(0)
            k, i = spec(loc)
            out_kind.append(k)
            out_id.append(i)
        return first

    def phone_spec(ph, i):
        L = len(ph)
        c = ph[i]
        if L == 1:
            return lambda loc: (3, row_id("lr", (c, loc)))
        if i == 0:
            return lambda loc: (2, row_id("r", (c, ph[1], loc)))
        if i == L - 1:
            return lambda loc: (1, set_id((ph[i - 1], c, loc)))
        return lambda loc: (0, tri(ph[i - 1], c, ph[i + 1], loc))

    def link(frm_last, to_first):
        if to_first == frm_last + 1:
            next_a[frm_last] = A_EXIT
        else:
            ac.setdefault(frm_last, []).append((to_first, A_EXIT))

    word_head = np.zeros(W, np.int32)
    word_end = np.zeros(W, np.int32)
    words_below = {}                           # first node of a tree edge -> list of words
    trie = {}                                  # (parent_first_node or -1, logical name) -> first node
    children = {}                              # parent edge first node (-1 root) -> count
    roots = []                                 # first nodes of root edges, in creation order

    def logical(ph, i):
        L = len(ph)
        if L == 1:
            return ("m", ph[0])
        if i == 0:
            return ("h", ph[0], ph[1])
        if i == L - 1:
            return ("t", ph[i - 1], ph[i])
        return ("i", ph[i - 1], ph[i], ph[i + 1])

    def add_chain(w, ph, start_i, prev_last):
        """append phones ph[start_i:] as new nodes after node prev_last (or as a root)"""
        firsts = []
        for i in range(start_i, len(ph)):
            f = add_phone(phone_spec(ph, i))
            if prev_last is not None:
                link(prev_last, f)
            prev_last = f + 2
            firsts.append(f)
        stend[prev_last] = w
        word_end[w] = prev_last
        return firsts

    # silence words: single "phone" chains with plain states from the top of the pool
    for w, base in ((0, S - 6), (1, S - 3)):
        f = len(self_a)
        for loc in range(3):
            self_a.append(A_SELF if loc < 2 else A_SELF3)
            next_a.append(A_NEXT if loc < 2 else LZ)
            stend.append(-1); scid.append(0); out_kind.append(0); out_id.append(base + loc)
        stend[f + 2] = w
        word_head[w], word_end[w] = f, f + 2
    single_roots = [(word_head[1], 1)]         # </s> is a cross-word target, <s> is not
    edge_parent = {}
    for wi, ph in enumerate(seqs):
        w = wi + nsil
        if separated[wi]:
            firsts = add_chain(w, ph, 0, None)
            word_head[w] = firsts[0]
            single_roots.append((firsts[0], w))
            continue
        parent, depth, prev_last = -1, 0, None
        while depth < len(ph) and (parent, logical(ph, depth)) in trie:
            parent = trie[(parent, logical(ph, depth))]
            words_below[parent].append(w)
            prev_last = parent + 2
            depth += 1
        firsts = add_chain(w, ph, depth, prev_last)
        p = parent
        for i, f in zip(range(depth, len(ph)), firsts):
            trie[(p, logical(ph, i))] = f
            words_below[f] = [w]
            edge_parent[f] = p
            children[p] = children.get(p, 0) + 1
            if p == -1:
                roots.append(f)
            p = f
    # word_head for tree words = root edge containing the word
    for f in roots:
        for w in words_below[f]:
            word_head[w] = f

    # ---- factoring ids ----------------------------------------------------------------
    fscore, scword = [LZ], [0]
    def assign(f, ws):
        if len(ws) == 1:
            scid[f] = len(scword); scword.append(ws[0])
        else:
            scid[f] = -len(fscore); fscore.append(np.float32(max(uni[x] for x in ws)))
    for f, ws in words_below.items():
        p = edge_parent[f]
        if p == -1 or len(words_below[p]) != len(ws):
            assign(f, ws)
    for f, w in single_roots:
        assign(f, [w])
    # <s>: its head is entered only at t = 0; the reference gives it a successor id too
    assign(int(word_head[0]), [0])

    startnode = np.array(roots + [f for f, _ in single_roots], np.int32)
    s2i, niso = [], 0
    for f in startnode:
        if scid[f] >= 0:
            s2i.append(niso); niso += 1
        else:
            s2i.append(-1)

    # ---- context tables ---------------------------------------------------------------------
    nset0 = len(set_list)
    lc_tab = np.zeros((len(row_list), nlc + 1), np.int32)
    for r, (kind, key) in enumerate(row_list):
        for c in range(nlc + 1):
            lctx = c if c < nphone else nphone - 1 - (c - nphone) % nphone   # silence / none: borrow a phone
            if kind == "r":
                ce, rc, loc = key
                if c < nlc and rng.random() < defined_frac:
                    lc_tab[r, c] = tri(lctx, ce, rc, loc)
                else:
                    lc_tab[r, c] = ~set_id(("rs", ce, rc, loc))
            else:
                ce, loc = key
                lc_tab[r, c] = ~set_id(("lrs", ce, loc, c if c < nlc else -1))
    set_off, set_states = [0], []
    for key in set_list:
        if key[0] == "rs":
            _, ce, rc, loc = key
            st = sorted({tri(l, ce, rc, loc) for l in range(nphone)})
        elif key[0] == "lrs":
            _, ce, loc, c = key
            st = sorted({tri((c * 7 + r) % nphone, ce, r, loc) for r in range(nphone)})
        else:
            l, ce, loc = key
            st = sorted({tri(l, ce, r, loc) for r in range(nphone)})
        set_states += st
        set_off.append(len(set_states))

    # ---- words / LM -----------------------------------------------------------------------------
    word_lc = np.zeros(W, np.int32)
    word_lc[0], word_lc[1] = SILB, SILE
    for wi, ph in enumerate(seqs):
        word_lc[wi + nsil] = ph[-1]
    n = len(self_a)
    ac_off = np.zeros(n + 1, np.int32)
    ac_to, ac_a = [], []
    for i in range(n):
        for to, a in ac.get(i, []):
            ac_to.append(to); ac_a.append(a)
        ac_off[i + 1] = len(ac_to)
    wordend_a = np.full(W, A_EXIT, np.float32)
    bgn = np.full(W, -1, np.int32); num = np.zeros(W, np.int32)
    bw, bp = [], []
    for w in range(W):
        if w == 1:
            continue
        k = min(nbigram_per_word, W - 1)
        js = np.unique(np.concatenate([[1], rng.choice(np.arange(2, W), size=k - 1, replace=False)]))
        k = len(js)
        ps = np.log10(rng.dirichlet(np.full(k, 1.0)) * 0.7 + 1e-9)
        ps[0] = np.log10(0.1)                              # every word can end the sentence
        bgn[w], num[w] = len(bw), k
        bw += [int(j) for j in js]; bp += [float(p) for p in ps]
    return dict(
        nnode=n, nword=W, startnum=len(startnode), isolatenum=niso, nlc=nlc, nlcrow=len(row_list),
        nset=len(set_list), cdset_method=2, cdmax_num=3, head_silwid=0, tail_silwid=1,
        nfscore=len(fscore), nscword=len(scword), ng_mode=0, ng_nword=W, ng_nbigram=len(bw),
        ng_unk_id=2147483647, ng_unk_num_log=0.0, lm_weight=float(lm_weight), lm_penalty=float(lm_penalty),
        lm_penalty_trans=0.0,
        self_a=np.array(self_a, np.float32), next_a=np.array(next_a, np.float32), ac_off=ac_off,
        ac_to=np.array(ac_to, np.int32), ac_a=np.array(ac_a, np.float32), stend=np.array(stend, np.int32),
        scid=np.array(scid, np.int32), out_kind=np.array(out_kind, np.uint8), out_id=np.array(out_id, np.int32),
        lc_tab=lc_tab.reshape(-1), word_lc=word_lc, set_off=np.array(set_off, np.int32),
        set_states=np.array(set_states, np.int32), startnode=startnode, start2isolate=np.array(s2i, np.int32),
        wordend_a=wordend_a, wton=np.arange(W, dtype=np.int32), cprob=np.zeros(W, np.float32),
        is_transparent=np.zeros(W, np.uint8), word_head=word_head, fscore=np.array(fscore, np.float32),
        scword=np.array(scword, np.int32), ng_uni_prob=uni, ng_uni_bo=bo, ng_bi_bgn=bgn, ng_bi_num=num,
        ng_bi_wid=np.array(bw, np.int32), ng_bi_prob=np.array(bp, np.float32))


def make_lexicon_utterance(lex, model, nwords=8, seed=0, frames_per_state=3, noise=0.6):
    """Frames that follow a random word sequence <s> w1 .. wn </s> through the
    lexicon's own node chains (state centres + noise), so the first pass has a
    real sentence to find.  Returns (frames [T][D], word ids)."""
    rng = np.random.default_rng(seed)
    W = lex["nword"]
    ws = [0] + [int(x) for x in rng.integers(2, W, size=nwords)] + [1]
    # predecessor map of the tree: walk back from each word's end node to its head
    stend = lex["stend"]
    end_of = np.full(W, -1, np.int64)
    idx = np.nonzero(stend >= 0)[0]
    end_of[stend[idx]] = idx
    pred = {}
    ac_off, ac_to = lex["ac_off"], lex["ac_to"]
    src = np.repeat(np.arange(lex["nnode"]), np.diff(ac_off))
    for s_, t_ in zip(src, ac_to):
        pred[int(t_)] = int(s_)
    set_off, set_states = lex["set_off"], lex["set_states"]
    nlc = lex["nlc"]

    def state_of(node, prev_word):
        k, i = int(lex["out_kind"][node]), int(lex["out_id"][node])
        if k == 0:
            return i
        col = nlc if prev_word < 0 else int(lex["word_lc"][prev_word])
        ent = ~i if k == 1 else int(lex["lc_tab"][i * (nlc + 1) + col])
        return ent if ent >= 0 else int(set_states[set_off[~ent]])

    seq = []
    for wi, w in enumerate(ws):
        path, node = [], int(end_of[w])
        while True:
            path.append(node)
            if node in pred:
                node = pred[node]
            elif node - 1 >= 0 and lex["next_a"][node - 1] > -1e5 and stend[node - 1] < 0 and node != int(lex["word_head"][w]):
                node -= 1
            else:
                break
        seq += [state_of(nd, ws[wi - 1] if wi > 0 else -1) for nd in reversed(path)]
    st = np.repeat(np.array(seq), frames_per_state)
    fr = model["centre"][st] + rng.normal(0, noise, size=(len(st), model["mean"].shape[1]))
    return fr.astype(np.float32), ws


# --------------------------------------------------- grammar task (BASELINE configs[0] shape)
def make_grammar_task(workdir, nphone=12, nword=100, nbook=2, K=64, D=39, seed=0, maxlen=4):
    """Tied-mixture monophone GMM-HMM (HTK ascii, <TMix> codebooks) + an `nword`-word
    loop grammar in Julius' DFA format: <s> WORD+ </s>.  The automaton is written
    reversed, as mkdfa.pl emits it (libsent/src/dfa/rddfa.c:141-200): state 0 consumes
    the sentence-final category.  Returns paths and the word list."""
    workdir = Path(workdir)
    workdir.mkdir(parents=True, exist_ok=True)
    rng = np.random.default_rng(seed)
    S = 3 * (nphone + 2)
    model = make_tied_gmm(S=S, nbook=nbook, K=K, D=D, seed=seed + 1)
    names = [f"p{i}" for i in range(nphone)] + ["silB", "silE"]
    phones = [(names[i], (3 * i, 3 * i + 1, 3 * i + 2)) for i in range(nphone + 2)]
    write_hmmdefs(workdir / "hmmdefs", model, phones=phones)
    words, seen = [], set()
    while len(words) < nword:
        ph = tuple(int(x) for x in rng.integers(0, nphone, size=int(rng.integers(1, maxlen + 1))))
        if ph in seen:
            continue
        seen.add(ph)
        words.append((f"W{len(words):03d}", ph))
    # categories: 0 = </s>, 1 = words, 2 = <s>
    dl = ["0 [</s>] silE", "2 [<s>] silB"] + [f"1 [{w}] " + " ".join(f"p{p}" for p in ph) for w, ph in words]
    (workdir / "g.dict").write_text("\n".join(dl) + "\n")
    (workdir / "g.dfa").write_text("0 0 1 0 0\n1 1 1 0 0\n1 2 2 0 0\n2 -1 -1 1 0\n")
    # state centres for utterance synthesis: mean of each state's heaviest codebook Gaussian
    off = model["st_off"]
    centre = np.stack([model["mean"][model["ent_dens"][off[s] + int(np.argmax(model["weight"][off[s]:off[s + 1]]))]]
                       for s in range(S)])
    return dict(dir=workdir, hmmdefs=workdir / "hmmdefs", dfa=workdir / "g.dfa", dict=workdir / "g.dict",
                model=model, words=words, nphone=nphone, centre=centre)


def make_grammar_utterance(task, nwords=4, seed=0, frames_per_state=3, noise=0.5):
    rng = np.random.default_rng(seed)
    n = task["nphone"]
    ws = [task["words"][int(i)] for i in rng.integers(0, len(task["words"]), size=nwords)]
    seq = [3 * n, 3 * n + 1, 3 * n + 2]
    for _, ph in ws:
        for p in ph:
            seq += [3 * p, 3 * p + 1, 3 * p + 2]
    seq += [3 * (n + 1), 3 * (n + 1) + 1, 3 * (n + 1) + 2]
    st = np.repeat(np.array(seq), frames_per_state)
    fr = task["centre"][st] + rng.normal(0, noise, size=(len(st), task["centre"].shape[1]))
    return fr.astype(np.float32), [w for w, _ in ws]


def make_triphone_grammar(task, ncat=3, seed=0, wrap=True):
    """A DFA grammar over the words of a triphone task (make_triphone_task): word category
    c may be followed by category c+1 or c+2 (mod ncat); a sentence is <s> WORD+ </s>.  One
    lexicon tree per category with cross-word triphones, so one-phone words get category-aware
    state sets (lcdset_register_with_category(), libjulius/src/wchmm.c:1068-1110).  The automaton
    is written reversed as mkdfa.pl emits it.  Categories: 0 = </s>, 1 = <s>, 2.. = words.
    wrap=False drops <s>/</s> from the sentences (they stay in the dictionary), so that every word
    may start one: many initial tokens (init_nodescore(), libjulius/src/beam.c:1669-1757)."""
    rng = np.random.default_rng(seed + 77)
    workdir = Path(task["dir"])
    words = task["words"]
    cat = [int(rng.integers(0, ncat)) for _ in words]
    dl = ["0 [</s>] silE", "1 [<s>] silB"] + [f"{2 + c} [{w}] " + " ".join(ph) for (w, ph), c in zip(words, cat)]
    # the task's tee-only words (make_triphone_task(ntee=...)): a pause category that may stand between any two words,
    # once (states P_j = F + 1 + j below)
    tee = task.get("tee_words") or []
    dl += [f"{2 + ncat} [{w}] sp" for w in tee]
    (workdir / "g.dict").write_text("\n".join(dl) + "\n")
    # reversed automaton: 0 --</s>--> E(1); E --cat j--> R_j (2+j); R_j --cat i (i may precede j)--> R_i;
    # R_j --<s>--> F; F accepts
    F = 2 + ncat
    lines = ["0 0 1 0 0"] + [f"1 {2 + j} {2 + j} 0 0" for j in range(ncat)]
    if not wrap:         # state 0 = sentence end, any word category may be last; any R_j accepts
        lines = [f"0 {2 + j} {2 + j} 0 0" for j in range(ncat)] + ["1 0 1 0 0", "1 1 1 0 0"]   # state 1: unreachable, keeps <s>,</s> declared
    for j in range(ncat):
        for i in range(ncat):
            if (i + 1) % ncat == j or (i + 2) % ncat == j:
                lines.append(f"{2 + j} {2 + i} {2 + i} 0 0")
                if tee:                                                # ... or with one pause word between the two
                    lines.append(f"{F + 1 + j} {2 + i} {2 + i} 0 0")
        if tee:
            lines.append(f"{2 + j} {2 + ncat} {F + 1 + j} 0 0")          # (mkcpair.c:79-101: not first, not last, not twice)
        lines.append(f"{2 + j} 1 {F} 0 0" if wrap else f"{2 + j} -1 -1 1 0")
    if wrap:
        lines.append(f"{F} -1 -1 1 0")
    (workdir / "g.dfa").write_text("\n".join(lines) + "\n")
    g = dict(task)
    g.update(dfa=workdir / "g.dfa", gdict=workdir / "g.dict", word_cat=cat, ncat=ncat, wrap=wrap)
    return g


def make_triphone_grammar_utterance(g, nwords=4, seed=0, frames_per_state=3, noise=0.7, pause_prob=0.0):
    """Frames following a random sentence the grammar of make_triphone_grammar() accepts; with pause_prob > 0 a short
    pause (the sp model's state; the grammar's tee-only pause word) stands between some of the words."""
    rng = np.random.default_rng(seed)
    ncat = g["ncat"]
    by_cat = [[i for i, c in enumerate(g["word_cat"]) if c == k] for k in range(ncat)]
    c = int(rng.integers(0, ncat))
    ids = []
    for _ in range(nwords):
        if not by_cat[c]:
            c = (c + 1) % ncat
            continue
        ids.append(int(rng.choice(by_cat[c])))
        c = (c + int(rng.integers(1, 3))) % ncat
    model = g["model"]
    S = len(model["st_off"]) - 1
    per = (S - 6) // len(g["phones"])
    seq = [S - 6, S - 5, S - 4] if g.get("wrap", True) else []
    for n, i in enumerate(ids):
        if n > 0 and pause_prob > 0 and rng.random() < pause_prob:
            seq.append(S - 5)
        for p in g["words"][i][1]:
            base = g["phones"].index(p) * per
            seq += [int(base + rng.integers(0, per)) for _ in range(3)]
    if g.get("wrap", True):
        seq += [S - 3, S - 2, S - 1]
    st = np.repeat(np.array(seq), frames_per_state)
    fr = model["centre"][st] + rng.normal(0, noise, size=(len(st), model["mean"].shape[1]))
    return fr.astype(np.float32), [g["words"][i][0] for i in ids]


def make_wordlist_task(workdir, nphone=10, nword=60, seed=0, maxlen=5, triphone=True):
    """Isolated word recognition task (`-w list -wsil silB silE silB`, libsent/src/voca/
    voca_load_wordlist.c): one word per line, "OutputString phone...", the silence models are
    attached to both ends of every word by the loader.  Uses the models of make_triphone_task()
    (word-internal and silence-context triphones) or of make_grammar_task() (monophones)."""
    workdir = Path(workdir)
    if triphone:
        # cross-word context handling is off in this mode, so every triphone a word needs --
        # including the ones next to the attached silences -- must be a defined model
        task = make_triphone_task(workdir, nphone=nphone, seed=seed, nword=nword, maxlen=maxlen, defined_frac=1.0)
        lines = [f"{w} " + " ".join(ph) for w, ph in task["words"]]
    else:
        task = make_grammar_task(workdir, nphone=nphone, seed=seed, nword=nword, maxlen=maxlen)
        lines = [f"{w} " + " ".join(f"p{p}" for p in ph) for w, ph in task["words"]]
    (workdir / "words.list").write_text("\n".join(lines) + "\n")
    task = dict(task)
    task.update(wordlist=workdir / "words.list", triphone=triphone)
    return task


def make_wordlist_utterance(task, seed=0, frames_per_state=3, noise=0.7):
    """silB + one random word + silE."""
    rng = np.random.default_rng(seed)
    i = int(rng.integers(0, len(task["words"])))
    if task["triphone"]:
        model = task["model"]
        S = len(model["st_off"]) - 1
        per = (S - 6) // len(task["phones"])
        seq = [S - 6, S - 5, S - 4]
        for p in task["words"][i][1]:
            base = task["phones"].index(p) * per
            seq += [int(base + rng.integers(0, per)) for _ in range(3)]
        seq += [S - 3, S - 2, S - 1]
        centre = model["centre"]
    else:
        n = task["nphone"]
        seq = [3 * n, 3 * n + 1, 3 * n + 2]
        for p in task["words"][i][1]:
            seq += [3 * p, 3 * p + 1, 3 * p + 2]
        seq += [3 * (n + 1), 3 * (n + 1) + 1, 3 * (n + 1) + 2]
        centre = task["centre"]
    st = np.repeat(np.array(seq), frames_per_state)
    fr = centre[st] + rng.normal(0, noise, size=(len(st), centre.shape[1]))
    return fr.astype(np.float32), i


def make_gs_model(task, M=3, seed=0):
    """A Gaussian-mixture-selection model (-gshmm) for a triphone task: one 3-state GMM-HMM per
    PHYSICAL model of the task, its states named "<model><i+1>m" as build_state2gs() looks them up
    (libsent/src/phmm/gms.c:104-160: center name of the model + state index).  Written next to the
    task's files; returns the path and the flat model."""
    names = [f"{c}_v{v}" for c in task["phones"] for v in range(3)] + ["silB", "silE"]
    S = 3 * len(names)
    model = make_gmm(S=S, M=M, D=task["model"]["mean"].shape[1], seed=seed + 500)
    # centre the selection model on the task's acoustic space so that selections are non-trivial
    rng = np.random.default_rng(seed + 501)
    cen = task["model"]["centre"]
    for s in range(S):
        e0, e1 = int(model["st_off"][s]), int(model["st_off"][s + 1])
        model["mean"][model["ent_dens"][e0:e1]] = (cen[rng.integers(0, len(cen), e1 - e0)] +
                                                   rng.normal(0, 0.5, (e1 - e0, cen.shape[1]))).astype(np.float32)
    state_names = [f"{names[s // 3]}{s % 3 + 2}m" for s in range(S)]
    phones = [(names[h], (3 * h, 3 * h + 1, 3 * h + 2)) for h in range(len(names))]
    path = Path(task["dir"]) / "gshmm"
    write_hmmdefs(path, model, phones=phones, state_names=state_names)
    return path, model


def make_rejection_gmm(workdir, centre, names=("speech", "noise", "music", "cough"), M=14, seed=0,
                       ragged=True, null_frac=0.0, kind="MFCC_E_D_A"):
    """GMM definitions for input verification / rejection (-gmm FILE, libjulius/src/gmm.c): one HMM
    per name with a single output state (three states in HTK's count, gmm_init() gmm.c:436-442).
    The mixtures are spread around `centre` rows (the task's acoustic space) so that the private safe
    pruning has something to prune.  Returns (path, flat model, names)."""
    D = centre.shape[1]
    model = make_gmm(S=len(names), M=M, D=D, seed=seed + 700, ragged=ragged, null_frac=null_frac)
    rng = np.random.default_rng(seed + 701)
    for s in range(len(names)):
        e0, e1 = int(model["st_off"][s]), int(model["st_off"][s + 1])
        ok = model["ent_dens"][e0:e1] >= 0
        model["mean"][model["ent_dens"][e0:e1][ok]] = (centre[rng.integers(0, len(centre), int(ok.sum()))] +
                                                       rng.normal(0, 1.0 + s, (int(ok.sum()), D))).astype(np.float32)
    L = [f"~o <STREAMINFO> 1 {D} <VECSIZE> {D} <NULLD> <{kind}> <DIAGC>"]
    for s, name in enumerate(names):
        e0, e1 = int(model["st_off"][s]), int(model["st_off"][s + 1])
        L.append(f'~h "{name}"\n<BEGINHMM>\n<NUMSTATES> 3\n<STATE> 2\n<NUMMIXES> {e1 - e0}')
        for m, e in enumerate(range(e0, e1)):
            if model["ent_dens"][e] < 0:
                continue
            L.append(f"<MIXTURE> {m + 1} {model['weight'][e]:.6e}")
            L.append(f"<MEAN> {D}\n {_vec(model['mean'][model['ent_dens'][e]])}")
            L.append(f"<VARIANCE> {D}\n {_vec(model['var'][model['ent_dens'][e]])}")
        L.append("<TRANSP> 3\n 0.0 1.0 0.0\n 0.0 0.6 0.4\n 0.0 0.0 0.0\n<ENDHMM>")
    path = Path(workdir) / "rejgmm"
    path.write_text("\n".join(L) + "\n")
    return path, model, list(names)


def make_forward_grammar(task, ncat=3, maxwords=4, seed=0):
    """A DFA grammar over the words of a triphone task WITH the forward automaton recent mkdfa.pl writes next to the
    reversed one (`g.dfa.forward`, read by libjulius/src/multi-gram.c:868-880): sentences <s> W{2..maxwords} </s> where a
    word of category c is followed by one of category c+1 or c+2 (mod ncat).  The length bound is what the category-pair
    constraint of the first pass (dfa_cp(), from the reversed automaton) cannot express, so the forward automaton's
    states really prune cross-word transitions (libjulius/src/beam.c:2412-2422).  The reversed automaton g.dfa is derived
    from the forward one (NFA reversal + subset construction), so both describe the same language.
    Categories: 0 = </s>, 1 = <s>, 2.. = words.  Returns make_triphone_grammar()'s dict + `fwd` = (arcs, accepting)."""
    rng = np.random.default_rng(seed + 177)
    workdir = Path(task["dir"])
    words = task["words"]
    cat = [int(rng.integers(0, ncat)) for _ in words]
    dl = ["0 [</s>] silE", "1 [<s>] silB"] + [f"{2 + c} [{w}] " + " ".join(ph) for (w, ph), c in zip(words, cat)]
    # the task's tee-only words (make_triphone_task(ntee=...)): a pause category that may stand between any two words,
    # once (states P_j = F + 1 + j below)
    tee = task.get("tee_words") or []
    dl += [f"{2 + ncat} [{w}] sp" for w in tee]
    (workdir / "g.dict").write_text("\n".join(dl) + "\n")
    # forward automaton: 0 --<s>--> 1 (no word yet); (p words, last category c) = state 2 + (p - 1) * ncat + c; final F
    arcs = {(0, 1): 1}
    F = 2 + maxwords * ncat
    st = lambda p, c: 2 + (p - 1) * ncat + c
    for c in range(ncat):
        arcs[(1, 2 + c)] = st(1, c)
    for p in range(1, maxwords + 1):
        for c in range(ncat):
            if p < maxwords:
                for c2 in ((c + 1) % ncat, (c + 2) % ncat):
                    arcs[(st(p, c), 2 + c2)] = st(p + 1, c2)
            if p >= 2:
                arcs[(st(p, c), 0)] = F
    accept = {F}
    fl = [f"{s} {l} {n} {1 if s in accept else 0} 0" for (s, l), n in sorted(arcs.items())] + [f"{F} -1 -1 1 0"]
    (workdir / "g.dfa.forward").write_text("\n".join(fl) + "\n")
    # reversed automaton by subset construction: start = accepting states, accept = subsets holding the forward start
    rev = {}
    for (s, l), n in arcs.items():
        rev.setdefault((n, l), set()).add(s)
    start = frozenset(accept)
    ids, order, lines = {start: 0}, [start], []
    i = 0
    while i < len(order):
        cur = order[i]
        labels = sorted({l for (n, l) in rev if n in cur})
        out_any = False
        for l in labels:
            nxt = frozenset(s for n in cur for s in rev.get((n, l), ()))
            if nxt not in ids:
                ids[nxt] = len(order); order.append(nxt)
            lines.append(f"{ids[cur]} {l} {ids[nxt]} {1 if 0 in cur else 0} 0")
            out_any = True
        if not out_any:
            lines.append(f"{ids[cur]} -1 -1 {1 if 0 in cur else 0} 0")
        i += 1
    (workdir / "g.dfa").write_text("\n".join(lines) + "\n")
    g = dict(task)
    g.update(dfa=workdir / "g.dfa", gdict=workdir / "g.dict", word_cat=cat, ncat=ncat, wrap=True, fwd=(arcs, accept), maxwords=maxwords)
    return g


def make_forward_grammar_utterance(g, seed=0, frames_per_state=3, noise=0.7, nwords=None):
    """Frames along a random sentence of make_forward_grammar()'s language -- or, nwords > maxwords, along a category chain
    that is TOO LONG for it (every adjacent pair allowed, the length not): what the forward automaton is there to cut."""
    rng = np.random.default_rng(seed)
    ncat = g["ncat"]
    by_cat = [[i for i, c in enumerate(g["word_cat"]) if c == k] for k in range(ncat)]
    n = int(rng.integers(2, g["maxwords"] + 1)) if nwords is None else nwords
    c = int(rng.integers(0, ncat))
    ids = []
    for _ in range(n):
        while not by_cat[c]:
            c = (c + 1) % ncat
        ids.append(int(rng.choice(by_cat[c])))
        c = (c + int(rng.integers(1, 3))) % ncat
    model = g["model"]
    S = len(model["st_off"]) - 1
    per = (S - 6) // len(g["phones"])
    seq = [S - 6, S - 5, S - 4]
    for i in ids:
        for p in g["words"][i][1]:
            base = g["phones"].index(p) * per
            seq += [int(base + rng.integers(0, per)) for _ in range(3)]
    seq += [S - 3, S - 2, S - 1]
    st = np.repeat(np.array(seq), frames_per_state)
    fr = model["centre"][st] + rng.normal(0, noise, size=(len(st), model["mean"].shape[1]))
    return fr.astype(np.float32), [g["words"][i][0] for i in ids]
