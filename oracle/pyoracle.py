"""ctypes access to the parity oracle.  TEST INFRASTRUCTURE ONLY.

Two libraries live under oracle/:
  liboracle.so        our CPU restatement (jamd_oracle_*.c)           -> class Oracle
  _ref/libjref.so     the compiled, unmodified reference + taps       -> class Ref
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module, and only as the checker / reported baseline.
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
ORACLE_SO = HERE / "liboracle.so"
REF_SO = HERE / "_ref" / "libjref.so"
REF_AMD_SO = HERE / "_ref" / "libjref_amd.so"   # same reference, first pass served by julius_amd/shim
PLUGIN_DIR = HERE / "_ref" / "plugin"
REF_O_SO = HERE / "_ref" / "libjref_o.so"       # same reference (own beam), scoring entry points wrapped (boundary O)

GPRUNE_NONE, GPRUNE_SAFE, GPRUNE_HEU, GPRUNE_BEAM = 0, 1, 2, 3
# reference enum (libsent/include/sent/hmm_calc.h:45)
REF_GPRUNE = {"none": 1, "safe": 2, "heu": 3, "beam": 4}
IWCD_MAX, IWCD_AVG, IWCD_NBEST = 0, 1, 2
# reference enum iwcd_type (libsent/include/sent/htk_hmm.h:85-90)
REF_IWCD = {"max": 1, "avg": 2, "nbest": 3}
DNN_SCALAR, DNN_FMA, DNN_AVX, DNN_SSE = 0, 1, 2, 3


def build(ref: bool = True):
    """(Re)build liboracle.so and, when /root/reference exists, _ref/libjref.so."""
    subprocess.run(["make", "-s", "-C", str(HERE), "oracle"], check=True)
    if ref:
        subprocess.run(["make", "-s", "-j8", "-C", str(HERE), "ref"], check=True)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class Oracle:
    def __init__(self):
        if not ORACLE_SO.exists():
            build(ref=False)
        self.lib = lib = C.CDLL(str(ORACLE_SO))
        vp, ci, cf = C.c_void_p, C.c_int, C.c_float
        lib.jo_log_tbl.restype = vp
        lib.jo_log_tbl_size.restype = ci
        lib.jo_logistic_tbl.restype = vp
        lib.jo_logistic_tbl_size.restype = ci
        lib.jo_addlog_array.restype = cf
        lib.jo_addlog_array.argtypes = [vp, ci]
        lib.jo_addlog.restype = cf
        lib.jo_addlog.argtypes = [cf, cf]
        lib.jo_logistic.restype = cf
        lib.jo_logistic.argtypes = [cf]
        lib.jo_gmm_outprob.restype = ci
        lib.jo_gmm_outprob.argtypes = [ci, ci, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, vp, ci, vp]
        lib.jo_tmix_topn.restype = ci
        lib.jo_tmix_topn.argtypes = [ci, vp, vp, vp, vp, ci, ci, ci, vp, ci, vp, vp, vp]
        lib.jo_outprob_cd.restype = cf
        lib.jo_outprob_cd.argtypes = [vp, vp, ci, ci, ci]
        lib.jo_dnn_outprob.restype = ci
        lib.jo_dnn_outprob.argtypes = [ci, vp, vp, vp, vp, ci, vp, ci, vp]

    def log_tbl(self):
        n = self.lib.jo_log_tbl_size()
        return np.ctypeslib.as_array(C.cast(self.lib.jo_log_tbl(), C.POINTER(C.c_float)), (n,)).copy()

    def logistic_tbl(self):
        n = self.lib.jo_logistic_tbl_size()
        return np.ctypeslib.as_array(C.cast(self.lib.jo_logistic_tbl(), C.POINTER(C.c_float)), (n,)).copy()

    def addlog_array(self, a):
        a = _f32(a)
        return float(self.lib.jo_addlog_array(_p(a), len(a)))

    def gmm_outprob(self, model, frames, gprune=GPRUNE_NONE, gprune_num=0):
        mean, ivar, gconst = _f32(model["mean"]), _f32(model["ivar"]), _f32(model["gconst"])
        st_off, ent_dens, ent_logw = _i32(model["st_off"]), _i32(model["ent_dens"]), _f32(model["ent_logw"])
        nbook = int(model.get("nbook", 0) or 0)
        st_book = _i32(model["st_book"]) if model.get("st_book") is not None else None
        fr = _f32(frames)
        T, D = fr.shape
        S = len(st_off) - 1
        out = np.empty((T, S), dtype=np.float32)
        rc = self.lib.jo_gmm_outprob(S, D, _p(mean), _p(ivar), _p(gconst), _p(st_off), _p(ent_dens),
                                     _p(ent_logw), _p(st_book) if st_book is not None else None,
                                     nbook, gprune, gprune_num, _p(fr), T, _p(out))
        assert rc == 0
        return out

    def gms_apply(self, gs, frames, scores):
        """Gaussian mixture selection: gs = dict(model=flat selection model, state2gs, nbest) as
        RefAM.gms() returns it; scores [T][S] real state scores -> the scores gms_state() returns."""
        m = gs["model"]
        mean, ivar, gconst = _f32(m["mean"]), _f32(m["ivar"]), _f32(m["gconst"])
        st_off, ent_dens, ent_logw = _i32(m["st_off"]), _i32(m["ent_dens"]), _f32(m["ent_logw"])
        s2g = _i32(gs["state2gs"])
        fr = _f32(frames)
        out = np.array(scores, dtype=np.float32, order="C", copy=True)
        rc = self.lib.jo_gms_apply(len(st_off) - 1, fr.shape[1], _p(mean), _p(ivar), _p(gconst), _p(st_off), _p(ent_dens),
                                   _p(ent_logw), _p(s2g), out.shape[1], int(gs["nbest"]), _p(fr), fr.shape[0], _p(out))
        assert rc == 0
        return out

    def rejgmm_frame_scores(self, gm, frames):
        """gmm.c's per-frame model scores; gm = RefEngine.gmm_info()-shaped dict."""
        m = gm["model"]
        mean, ivar, gconst = _f32(m["mean"]), _f32(m["ivar"]), _f32(m["gconst"])
        st_off, ent_dens, ent_logw = _i32(m["st_off"]), _i32(m["ent_dens"]), _f32(m["ent_logw"])
        ms = _i32(gm["model_state"])
        fr = _f32(frames)
        out = np.zeros((fr.shape[0], len(ms)), np.float32)
        rc = self.lib.jo_rejgmm_frame_scores(fr.shape[1], _p(mean), _p(ivar), _p(gconst), _p(st_off), _p(ent_dens),
                                             _p(ent_logw), _p(ms), len(ms), int(gm["gprune_num"]), _p(fr),
                                             fr.shape[0], _p(out))
        assert rc == 0
        return out

    @staticmethod
    def rejgmm_accumulate(frame_scores):
        """gmm_proceed()'s running sums (gmm.c:599): float adds in frame order."""
        acc = np.zeros(frame_scores.shape[1], np.float32)
        for row in np.asarray(frame_scores, np.float32):
            acc = (acc + row).astype(np.float32)
        return acc

    def tmix_topn(self, model, book, frames, gprune, gprune_num):
        mean, ivar, gconst = _f32(model["mean"]), _f32(model["ivar"]), _f32(model["gconst"])
        # a codebook's densities, in codebook order, are the entries of any state tied to it
        s0 = int(np.nonzero(np.asarray(model["st_book"]) == book)[0][0])
        dens = _i32(model["ent_dens"][model["st_off"][s0]:model["st_off"][s0 + 1]])
        K = len(dens)
        fr = _f32(frames)
        T, D = fr.shape
        cap = K if gprune == GPRUNE_NONE else gprune_num
        sc = np.zeros((T, cap), dtype=np.float32)
        ids = np.zeros((T, cap), dtype=np.int32)
        num = np.zeros(T, dtype=np.int32)
        self.lib.jo_tmix_topn(D, _p(mean), _p(ivar), _p(gconst), _p(dens), K, gprune, gprune_num,
                              _p(fr), T, _p(sc), _p(ids), _p(num))
        return sc, ids, num

    def outprob_cd(self, scores, set_off, states, method, nbest):
        scores = _f32(scores)
        set_off, states = _i32(set_off), _i32(states)
        T = scores.shape[0]
        nset = len(set_off) - 1
        out = np.empty((T, nset), dtype=np.float32)
        for t in range(T):
            row = scores[t]
            for i in range(nset):
                sub = states[set_off[i]:set_off[i + 1]]
                out[t, i] = self.lib.jo_outprob_cd(_p(row), _p(sub), len(sub), method, nbest)
        return out

    def beam_pass1(self, lex, scores, beam_width, score_pruning_width=-1.0, atom_cap=None):
        """First pass over a [T][S] state score matrix.  Returns (atoms structured
        array in emission order, pass-1 word sequence, pass-1 score, rc, died_at)."""
        from julius_amd import lexblob
        lib = self.lib
        lib.jo_beam_pass1.restype = C.c_int
        lib.jo_beam_pass1.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float,
                                      C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                      C.c_void_p, C.c_void_p]
        d, keep = lexblob.make_desc(lex)
        sc = _f32(scores)
        T, S = sc.shape
        cap = atom_cap or (T * max(beam_width, 16) + 16)
        atoms = np.zeros(cap, dtype=lexblob.ATOM_DTYPE)
        natom, wnum, died = C.c_int(), C.c_int(), C.c_int()
        score = C.c_float()
        wseq = np.zeros(4096, np.int32)
        rc = lib.jo_beam_pass1(C.byref(d), _p(sc), T, S, beam_width, score_pruning_width, _p(atoms), cap,
                               C.byref(natom), _p(wseq), len(wseq), C.byref(wnum), C.byref(score), C.byref(died))
        return atoms[:natom.value].copy(), wseq[:wnum.value].copy(), float(score.value), rc, died.value

    def sort_token_no_order(self, scores, beam_width):
        """beam.c:1492: visiting order (token ids) of the next frame for tokens with these scores in creation order."""
        sc = _f32(scores)
        out = np.zeros(max(len(sc), 1), np.int32)
        self.lib.jo_sort_token_no_order.restype = C.c_int
        self.lib.jo_sort_token_no_order.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        k = self.lib.jo_sort_token_no_order(_p(sc), len(sc), beam_width, _p(out))
        return out[:k].copy()

    def sort_token_arrange(self, scores, beam_width):
        """The same with the whole array: (visiting order, tindex[0..n) after the sort)."""
        sc = _f32(scores)
        out = np.zeros(max(len(sc), 1), np.int32)
        arr = np.zeros(max(len(sc), 1), np.int32)
        self.lib.jo_sort_token_arrange.restype = C.c_int
        self.lib.jo_sort_token_arrange.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        k = self.lib.jo_sort_token_arrange(_p(sc), len(sc), beam_width, _p(out), _p(arr))
        return out[:k].copy(), arr[:len(sc)].copy()

    def dnn_outprob(self, dnn, frames, simd=DNN_FMA):
        dims = _i32(dnn["dims"])
        nl = len(dims) - 1
        ws = [_f32(w) for w in dnn["w"]]
        bs = [_f32(b) for b in dnn["b"]]
        wp = (C.c_void_p * nl)(*[w.ctypes.data for w in ws])
        bp = (C.c_void_p * nl)(*[b.ctypes.data for b in bs])
        prior = _f32(dnn["prior"])
        fr = _f32(frames)
        T = fr.shape[0]
        out = np.empty((T, int(dims[-1])), dtype=np.float32)
        rc = self.lib.jo_dnn_outprob(nl, _p(dims), wp, bp, _p(prior), simd, _p(fr), T, _p(out))
        assert rc == 0
        return out


class Ref:
    """The compiled reference (oracle/_ref/libjref.so)."""

    def __init__(self, quiet=True, so=None, global_symbols=False):
        """global_symbols: load with RTLD_GLOBAL, as an executable's symbols are visible -- needed
        when the reference is to dlopen() a plugin that calls back into libsent (jlog, mymalloc).
        Use it in a process that loads no other variant of the reference."""
        so = so or REF_SO
        if not so.exists():
            raise FileNotFoundError(
                f"{so} missing: run `make -C oracle ref` where /root/reference is available")
        self.lib = lib = C.CDLL(str(so), mode=C.RTLD_GLOBAL) if global_symbols else C.CDLL(str(so))
        vp, ci, cd = C.c_void_p, C.c_int, C.c_double
        lib.jref_quiet.argtypes = [ci]
        lib.jref_am_load.restype = vp
        lib.jref_am_load.argtypes = [C.c_char_p, C.c_char_p, ci, ci, ci, ci]
        lib.jref_am_free.argtypes = [vp]
        lib.jref_am_dims.argtypes = [vp, vp]
        lib.jref_am_export.argtypes = [vp] * 8
        lib.jref_am_outprob.restype = cd
        lib.jref_am_outprob.argtypes = [vp, vp, ci, vp]
        lib.jref_am_outprob_list.restype = cd
        lib.jref_am_outprob_list.argtypes = [vp, vp, ci, vp, vp, ci, vp]
        lib.jref_am_tmix_cache.restype = ci
        lib.jref_am_tmix_cache.argtypes = [vp, vp, ci, ci, vp, vp, vp]
        lib.jref_am_outprob_cd.argtypes = [vp, vp, ci, ci, vp, vp, vp]
        lib.jref_am_from_flat.restype = vp
        lib.jref_am_from_flat.argtypes = [ci, ci, ci, vp, vp, vp, vp, vp, vp, ci, ci]
        lib.jref_dnn_load.restype = vp
        lib.jref_dnn_load.argtypes = [ci, ci, ci, ci, ci, ci, vp, vp, C.c_char_p, C.c_char_p, C.c_char_p,
                                      C.c_float, ci, ci]
        lib.jref_dnn_outprob.restype = cd
        lib.jref_dnn_outprob.argtypes = [vp, vp, ci, vp]
        lib.jref_simd_string.restype = C.c_char_p
        lib.jref_simd_avail.restype = ci
        lib.jref_quiet(1 if quiet else 0)

    def am_load(self, hmmdefs, hmmlist=None, gprune="none", gprune_num=2, cdset="max", cdmax=3, gshmm=None, gms_num=24):
        """gshmm: Gaussian mixture selection model (-gshmm), gms_num = -gsnum."""
        if gshmm is not None:
            self.lib.jref_am_load_gms.restype = C.c_void_p
            self.lib.jref_am_load_gms.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int]
            h = self.lib.jref_am_load_gms(str(hmmdefs).encode(), str(hmmlist).encode() if hmmlist else None,
                                          REF_GPRUNE[gprune], gprune_num, REF_IWCD[cdset], cdmax, str(gshmm).encode(), gms_num)
        else:
            h = self.lib.jref_am_load(str(hmmdefs).encode(), str(hmmlist).encode() if hmmlist else None,
                                      REF_GPRUNE[gprune], gprune_num, REF_IWCD[cdset], cdmax)
        if not h:
            raise RuntimeError(f"reference failed to load {hmmdefs}")
        return RefAM(self, h)

    def dnn_load(self, dnn, workdir, num_threads=1):
        """Write the network as .npy / prior files and load it with the reference's
        dnn_setup().  All hidden layers must share one width (reference limitation)."""
        from julius_amd import synth
        from pathlib import Path
        workdir = Path(workdir)
        dims = [int(x) for x in dnn["dims"]]
        nl = len(dims) - 1
        assert len(set(dims[1:-1])) == 1, "reference needs equal hidden widths"
        wf, bf = [], []
        for l in range(nl):
            synth.write_npy(workdir / f"W{l}.npy", dnn["w"][l])
            synth.write_npy(workdir / f"b{l}.npy", np.asarray(dnn["b"][l]).reshape(-1, 1))
            wf.append(str(workdir / f"W{l}.npy").encode()); bf.append(str(workdir / f"b{l}.npy").encode())
        with open(workdir / "prior", "w") as f:
            for i, v in enumerate(dnn["prior_lin"]):
                f.write(f"{i} {float(v):.9e}\n")
        nh = nl - 1
        wa = (C.c_char_p * nh)(*wf[:nh]); ba = (C.c_char_p * nh)(*bf[:nh])
        h = self.lib.jref_dnn_load(dims[0], 1, dims[0], dims[-1], dims[1], nh, wa, ba, wf[-1], bf[-1],
                                   str(workdir / "prior").encode(), 1.0, 1, num_threads)
        if not h:
            raise RuntimeError("reference dnn_setup failed")
        return RefDNN(self, h, dims)

    def am_from_flat(self, model, gprune="none", gprune_num=0):
        """Reference scoring code over in-memory structures built from flat arrays
        (plain states only); the arrays are kept alive by the returned object."""
        keep = dict(mean=_f32(model["mean"]), ivar=_f32(model["ivar"]), gconst=_f32(model["gconst"]),
                    st_off=_i32(model["st_off"]), ent_dens=_i32(model["ent_dens"]),
                    ent_logw=_f32(model["ent_logw"]))
        S = len(keep["st_off"]) - 1
        G, D = keep["mean"].shape
        h = self.lib.jref_am_from_flat(S, D, G, *[_p(keep[k]) for k in
                                                  ("mean", "ivar", "gconst", "st_off", "ent_dens", "ent_logw")],
                                       REF_GPRUNE[gprune], gprune_num)
        if not h:
            raise RuntimeError("jref_am_from_flat failed")
        return RefAM(self, h, flat=keep)


class RefAM:
    def __init__(self, ref: Ref, h, flat=None):
        self.ref, self.h = ref, h
        if flat is not None:  # built by jref_am_from_flat: dims come from the arrays
            self._keep = flat
            self.S = len(flat["st_off"]) - 1
            self.G, self.D = flat["mean"].shape
            self.E, self.nbook, self.is_tied = len(flat["ent_dens"]), 0, 0
            self.from_flat = True
            return
        self.from_flat = False
        dims = np.zeros(8, dtype=np.int32)
        assert ref.lib.jref_am_dims(h, _p(dims)) == 0, "flatten failed"
        (self.S, self.D, self.G, self.E, self.nbook, self.book_size_max, self.is_tied,
         self.maxmix) = [int(x) for x in dims]

    def export(self):
        m = dict(
            mean=np.empty((self.G, self.D), np.float32), ivar=np.empty((self.G, self.D), np.float32),
            gconst=np.empty(self.G, np.float32), st_off=np.empty(self.S + 1, np.int32),
            ent_dens=np.empty(self.E, np.int32), ent_logw=np.empty(self.E, np.float32),
            st_book=np.empty(self.S, np.int32))
        rc = self.ref.lib.jref_am_export(self.h, *[_p(m[k]) for k in
                                                   ("mean", "ivar", "gconst", "st_off", "ent_dens",
                                                    "ent_logw", "st_book")])
        assert rc == 0
        m["nbook"] = self.nbook
        m["nstream"] = 1
        if self.nbook:
            m["book_size"] = self.book_size_max
        else:
            m["st_book"] = None
        return m

    def gms(self):
        """Gaussian mixture selection data of a model loaded with gshmm=...: the flattened selection
        model, state2gs[S] and the number of selected states (gms.c)."""
        lib = self.ref.lib
        lib.jref_am_gms_model.restype = C.c_void_p
        lib.jref_am_gms_model.argtypes = [C.c_void_p]
        lib.jref_am_gms_map.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        v = lib.jref_am_gms_model(self.h)
        if not v:
            raise RuntimeError("model was loaded without a selection model")
        gs = RefAM(self.ref, v).export()
        s2g = np.zeros(self.S, np.int32)
        nb = C.c_int()
        n = lib.jref_am_gms_map(self.h, _p(s2g), C.byref(nb))
        assert n == len(gs["st_off"]) - 1
        return dict(model=gs, state2gs=s2g, nbest=int(nb.value))

    def save_blob(self, path):
        self.ref.lib.jref_am_save.argtypes = [C.c_void_p, C.c_char_p]
        assert self.ref.lib.jref_am_save(self.h, str(path).encode()) == 0

    def outprob(self, frames, want_out=True):
        fr = _f32(frames)
        T = fr.shape[0]
        out = np.empty((T, self.S), np.float32) if want_out else None
        sec = self.ref.lib.jref_am_outprob(self.h, _p(fr), T, _p(out) if want_out else None)
        self.last_seconds = float(sec)
        return out

    def outprob_list(self, frames, tt, ss):
        fr, tt, ss = _f32(frames), _i32(tt), _i32(ss)
        out = np.empty(len(tt), np.float32)
        self.last_seconds = float(
            self.ref.lib.jref_am_outprob_list(self.h, _p(fr), fr.shape[0], _p(tt), _p(ss), len(tt), _p(out)))
        return out

    def tmix_cache(self, frames, book, cap):
        fr = _f32(frames)
        T = fr.shape[0]
        sc = np.zeros((T, cap), np.float32)
        ids = np.zeros((T, cap), np.int32)
        num = np.zeros(T, np.int32)
        got = self.ref.lib.jref_am_tmix_cache(self.h, _p(fr), T, book, _p(sc), _p(ids), _p(num))
        assert got == cap, (got, cap)
        return sc, ids, num

    def outprob_cd(self, frames, set_off, states):
        fr, set_off, states = _f32(frames), _i32(set_off), _i32(states)
        nset = len(set_off) - 1
        out = np.empty((fr.shape[0], nset), np.float32)
        self.ref.lib.jref_am_outprob_cd(self.h, _p(fr), fr.shape[0], nset, _p(set_off), _p(states), _p(out))
        return out

    def close(self):
        if self.h and not self.from_flat:
            self.ref.lib.jref_am_free(self.h)
        self.h = None


class RefDNN:
    def __init__(self, ref: Ref, h, dims):
        self.ref, self.h, self.dims = ref, h, dims

    def outprob(self, frames, want_out=True):
        fr = _f32(frames)
        T = fr.shape[0]
        out = np.empty((T, self.dims[-1]), np.float32) if want_out else None
        self.last_seconds = float(self.ref.lib.jref_dnn_outprob(self.h, _p(fr), T, _p(out) if want_out else None))
        return out


class RefEngine:
    """A complete reference recogniser (j_create_instance_from_jconf) driven
    through the jref_engine_* taps of ref_driver.c: first-pass word trellis,
    pass-1 best sequence, and the flattened lexicon blob."""

    def __init__(self, ref: Ref, args):
        self.ref = ref
        lib = ref.lib
        vp, ci = C.c_void_p, C.c_int
        lib.jref_engine_create.restype = vp
        lib.jref_engine_create.argtypes = [ci, vp]
        lib.jref_engine_recognize.argtypes = [vp, C.c_char_p]
        lib.jref_engine_trellis.argtypes = [vp] * 8
        lib.jref_engine_pass1.argtypes = [vp, vp, vp]
        lib.jref_engine_info.argtypes = [vp, vp]
        lib.jref_engine_save_lexicon.argtypes = [vp, C.c_char_p]
        args = ["julius"] + [str(a) for a in args]
        self._argv = (C.c_char_p * len(args))(*[a.encode() for a in args])
        self.h = lib.jref_engine_create(len(args), self._argv)
        if not self.h:
            raise RuntimeError("reference engine failed to start: " + " ".join(args))
        info = np.zeros(16, np.int32)
        if lib.jref_engine_info(self.h, _p(info)) != 0:
            raise RuntimeError("the reference built no lexicon tree for: " + " ".join(args))
        (self.nnode, self.nword, self.startnum, self.isolatenum, self.beam_width, self.nstate,
         self.lmtype, self.multipath, self.ccd) = [int(x) for x in info[:9]]

    def set_eager(self, on=True):
        """Every state of a frame is scored at the first request (outprob.c:230-242) instead of lazily."""
        self.ref.lib.jref_engine_set_eager.argtypes = [C.c_void_p, C.c_int]
        self.ref.lib.jref_engine_set_eager(self.h, 1 if on else 0)
        return self

    def save_lexicon(self, path):
        rc = self.ref.lib.jref_engine_save_lexicon(self.h, str(path).encode())
        if rc != 0:
            raise RuntimeError(f"jamd_flatten_lexicon/jamd_lexicon_save failed ({rc})")

    def recognize(self, mfcfile):
        """Run the recogniser on one HTK parameter file.  Returns the word trellis
        in bt->rw[t][i] order as a dict of arrays, and (pass1 words, score)."""
        lib = self.ref.lib
        n = lib.jref_engine_recognize(self.h, str(mfcfile).encode())
        if n < 0:
            raise RuntimeError("reference recognition failed")
        a = {k: np.zeros(n, np.int32) for k in ("wid", "begintime", "endtime", "pwid", "pendtime")}
        a["backscore"] = np.zeros(n, np.float32)
        a["lscore"] = np.zeros(n, np.float32)
        got = lib.jref_engine_trellis(self.h, _p(a["wid"]), _p(a["begintime"]), _p(a["endtime"]),
                                      _p(a["backscore"]), _p(a["lscore"]), _p(a["pwid"]), _p(a["pendtime"]))
        assert got == n
        wseq = np.zeros(256, np.int32)
        sc = C.c_float()
        k = lib.jref_engine_pass1(self.h, _p(wseq), C.byref(sc))
        return a, (wseq[:k].copy(), float(sc.value))

    def gmm_info(self):
        """-gmm: dict(nmodel, gprune_num, veclen, model=flat state pool, model_state=[nmodel])."""
        lib = self.ref.lib
        info = np.zeros(3, np.int32)
        if lib.jref_engine_gmm_info(C.c_void_p(self.h), _p(info)) != 0:
            raise RuntimeError("no -gmm in this configuration")
        lib.jref_engine_gmm_model.restype = C.c_void_p
        view = RefAM(self.ref, lib.jref_engine_gmm_model(C.c_void_p(self.h)))
        states = np.zeros(int(info[0]), np.int32)
        assert lib.jref_engine_gmm_states(C.c_void_p(self.h), _p(states)) == info[0]
        return dict(nmodel=int(info[0]), gprune_num=int(info[1]), veclen=int(info[2]), model=view.export(),
                    model_state=states)

    def gmm_frame_scores(self, frames):
        """gmm_proceed()'s per-frame scores [T][nmodel] through the reference's own entry points."""
        fr = _f32(frames)
        info = np.zeros(3, np.int32)
        self.ref.lib.jref_engine_gmm_info(C.c_void_p(self.h), _p(info))
        out = np.zeros((fr.shape[0], int(info[0])), np.float32)
        rc = self.ref.lib.jref_engine_gmm_frame_scores(C.c_void_p(self.h), _p(fr), fr.shape[0], fr.shape[1], _p(out))
        if rc != 0:
            raise RuntimeError("jref_engine_gmm_frame_scores failed")
        return out

    def gmm_device_frames(self):
        """Frames the shim's gmm.c wrapper scored on the device (-1: plain reference / not supported)."""
        self.ref.lib.jref_engine_gmm_device_frames.restype = C.c_long
        return int(self.ref.lib.jref_engine_gmm_device_frames(C.c_void_p(self.h)))

    def gmm_result(self):
        """After recognize(): (accumulated scores, winner index, confidence, accepted?, frames)."""
        info = np.zeros(3, np.int32)
        self.ref.lib.jref_engine_gmm_info(C.c_void_p(self.h), _p(info))
        sc = np.zeros(int(info[0]), np.float32)
        mi, val, fc, cm = C.c_int(), C.c_int(), C.c_int(), C.c_float()
        rc = self.ref.lib.jref_engine_gmm_result(C.c_void_p(self.h), _p(sc), C.byref(mi), C.byref(cm), C.byref(val), C.byref(fc))
        if rc != 0:
            raise RuntimeError("no GMM result")
        return sc, int(mi.value), float(cm.value), bool(val.value), int(fc.value)

    def prefetch(self, mfcfiles):
        """Batch driver of the first-pass shim build (libjref_amd.so): decode all inputs in one
        device launch; the recognize() calls that follow find their first pass done."""
        lib = self.ref.lib
        arr = (C.c_char_p * len(mfcfiles))(*[str(f).encode() for f in mfcfiles])
        lib.jref_engine_prefetch.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.c_int]
        n = lib.jref_engine_prefetch(self.h, arr, len(mfcfiles))
        if n != len(mfcfiles):
            raise RuntimeError(f"batch first pass failed ({n})")

    def prefetch_served(self):
        lib = self.ref.lib
        lib.jref_engine_prefetch_served.argtypes = [C.c_void_p]
        return int(lib.jref_engine_prefetch_served(self.h))

    def cache_fill(self):
        """(defined, total) entries of the outprob cache after the last recognize()."""
        lib = self.ref.lib
        lib.jref_engine_cache_fill.argtypes = [C.c_void_p] * 3
        a, b = C.c_int(), C.c_int()
        lib.jref_engine_cache_fill(self.h, C.byref(a), C.byref(b))
        return int(a.value), int(b.value)

    def final_result(self):
        """(status, sentence-1 word ids, score) after the 2nd pass of the last recognize()."""
        lib = self.ref.lib
        lib.jref_engine_result.argtypes = [C.c_void_p] * 4
        wseq = np.zeros(256, np.int32)
        sc, st = C.c_float(), C.c_int()
        k = lib.jref_engine_result(self.h, _p(wseq), C.byref(sc), C.byref(st))
        return int(st.value), wseq[:k].copy(), float(sc.value)
