"""ctypes binding of the C ABI in include/julius_amd.h (libjulius_amd.so).

This module is plumbing for tests and bench.py; the product is the shared
library.  There is NO fallback: if the HIP library is missing, or no gfx950
device is present when an engine is created, it raises.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

_PKG = Path(__file__).resolve().parent
LIB_PATH = _PKG / "libjulius_amd.so"
if os.environ.get("JAMD_LIB"):          # development: a library variant built by tools/build_variant.sh
    LIB_PATH = Path(os.environ["JAMD_LIB"]).resolve()

LOG_ZERO = -1000000.0
GPRUNE_NONE, GPRUNE_SAFE, GPRUNE_HEU, GPRUNE_BEAM = 0, 1, 2, 3
IWCD_MAX, IWCD_AVG, IWCD_NBEST = 0, 1, 2


class JamdError(RuntimeError):
    pass


class GmmDesc(C.Structure):
    _fields_ = [
        ("nstate", C.c_int), ("veclen", C.c_int), ("ndens", C.c_int),
        ("nentry", C.c_int), ("nbook", C.c_int), ("nstream", C.c_int),
        ("mean", C.c_void_p), ("ivar", C.c_void_p), ("gconst", C.c_void_p),
        ("st_off", C.c_void_p), ("ent_dens", C.c_void_p), ("ent_logw", C.c_void_p),
        ("st_book", C.c_void_p),
    ]



class DnnDesc(C.Structure):
    _fields_ = [
        ("nlayer", C.c_int), ("dims", C.c_void_p), ("w", C.c_void_p), ("b", C.c_void_p),
        ("state_prior", C.c_void_p),
    ]


_lib = None


def load():
    """Load libjulius_amd.so (built by __graft_entry__.build() / csrc/Makefile)."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise JamdError(
            f"{LIB_PATH} not found: build the HIP engine first "
            "(python -c 'import __graft_entry__ as g; g.build()' or make -C julius_amd/csrc)")
    lib = C.CDLL(str(LIB_PATH), mode=C.RTLD_GLOBAL)
    vp, ci, cf = C.c_void_p, C.c_int, C.c_float
    P = C.POINTER
    sig = {
        "jamd_abi_version": (ci, []),
        "jamd_last_error": (C.c_char_p, []),
        "jamd_device_count": (ci, []),
        "jamd_engine_create": (ci, [ci, P(vp)]),
        "jamd_engine_destroy": (None, [vp]),
        "jamd_engine_device": (ci, [vp]),
        "jamd_engine_sync": (ci, [vp]),
        "jamd_malloc": (ci, [vp, C.c_size_t, P(vp)]),
        "jamd_free": (ci, [vp, vp]),
        "jamd_host_alloc": (ci, [vp, C.c_size_t, P(vp)]),
        "jamd_host_free": (ci, [vp, vp]),
        "jamd_memcpy_h2d": (ci, [vp, vp, vp, C.c_size_t]),
        "jamd_memcpy_d2h": (ci, [vp, vp, vp, C.c_size_t]),
        "jamd_gmm_create": (ci, [vp, P(GmmDesc), ci, ci, P(vp)]),
        "jamd_gmm_load": (ci, [vp, C.c_char_p, ci, ci, P(vp)]),
        "jamd_gmm_nentry": (ci, [vp]),
        "jamd_gmm_book_offsets": (ci, [vp, vp, ci]),
        "jamd_gmm_dens_dev": (ci, [vp, vp, ci, vp, vp]),
        "jamd_gmm_dens_host": (ci, [vp, vp, ci, vp]),
        "jamd_gms_create": (ci, [vp, P(GmmDesc), vp, ci, ci, P(vp)]),
        "jamd_gms_destroy": (None, [vp]),
        "jamd_gms_load": (ci, [vp, C.c_char_p, P(vp)]),
        "jamd_gms_nstate": (ci, [vp]),
        "jamd_gms_set_strict_order": (ci, [vp, ci]),
        "jamd_gms_apply_dev": (ci, [vp, vp, ci, vp, ci, vp, vp]),
        "jamd_gms_apply_host": (ci, [vp, vp, ci, vp, ci, vp]),
        "jamd_rejgmm_create": (ci, [vp, P(GmmDesc), vp, ci, ci, P(vp)]),
        "jamd_rejgmm_destroy": (None, [vp]),
        "jamd_rejgmm_nmodel": (ci, [vp]),
        "jamd_rejgmm_veclen": (ci, [vp]),
        "jamd_rejgmm_frame_scores_dev": (ci, [vp, vp, ci, vp, vp]),
        "jamd_rejgmm_utt_scores_dev": (ci, [vp, vp, ci, vp, ci, vp, vp]),
        "jamd_rejgmm_scores_host": (ci, [vp, vp, ci, vp, ci, vp, vp]),
        "jamd_dnn_load": (ci, [vp, C.c_char_p, P(vp)]),
        "jamd_gmm_load_binhmm": (ci, [vp, C.c_char_p, ci, ci, P(vp)]),
        "jamd_binhmm_to_blob": (ci, [C.c_char_p, C.c_char_p]),
        "jamd_lexicon_load": (ci, [vp, C.c_char_p, P(vp)]),
        "jamd_lexicon_load_ngram": (ci, [vp, C.c_char_p, C.c_char_p, P(vp)]),
        "jamd_bingram_check": (ci, [C.c_char_p, C.c_char_p, P(ci)]),
        "jamd_bingram_fscore": (ci, [C.c_char_p, C.c_char_p, vp, ci, P(ci)]),
        "jamd_gmm_destroy": (None, [vp]),
        "jamd_gmm_nstate": (ci, [vp]),
        "jamd_gmm_veclen": (ci, [vp]),
        "jamd_stream_create": (ci, [vp, vp]),
        "jamd_stream_destroy": (ci, [vp, vp]),
        "jamd_stream_wait": (ci, [vp, vp, vp]),
        "jamd_stream_sync": (ci, [vp, vp]),
        "jamd_memcpy_h2d_async": (ci, [vp, vp, vp, C.c_size_t, vp]),
        "jamd_gmm_outprob_dev": (ci, [vp, vp, ci, vp, vp]),
        "jamd_gmm_outprob_utts_dev": (ci, [vp, vp, vp, ci, vp, vp]),
        "jamd_gmm_outprob_host": (ci, [vp, vp, ci, vp]),
        "jamd_gmm_tmix_cache_dev": (ci, [vp, vp, ci, vp, vp, vp, vp]),
        "jamd_gmm_last_kernel": (C.c_char_p, [vp]),
        "jamd_gmm_tmix_cap": (ci, [vp]),
        "jamd_gmm_nbook": (ci, [vp]),
        "jamd_cdset_create": (ci, [vp, ci, vp, vp, ci, ci, P(vp)]),
        "jamd_cdset_destroy": (None, [vp]),
        "jamd_cdset_nset": (ci, [vp]),
        "jamd_cdset_outprob_dev": (ci, [vp, vp, ci, ci, vp, vp]),
        "jamd_dnn_create": (ci, [vp, P(DnnDesc), P(vp)]),
        "jamd_dnn_destroy": (None, [vp]),
        "jamd_dnn_nstate": (ci, [vp]),
        "jamd_dnn_veclen": (ci, [vp]),
        "jamd_dnn_outprob_dev": (ci, [vp, vp, ci, vp, vp]),
        "jamd_dnn_outprob_host": (ci, [vp, vp, ci, vp]),
        "jamd_lexicon_create": (ci, [vp, vp, P(vp)]),
        "jamd_lexicon_destroy": (None, [vp]),
        "jamd_beam_create": (ci, [vp, vp, ci, cf, ci, ci, P(vp)]),
        "jamd_beam_destroy": (None, [vp]),
        "jamd_beam_pass1_dev": (ci, [vp, vp, ci, vp, ci, vp]),
        "jamd_beam_results": (ci, [vp, vp, ci]),
        "jamd_beam_set_strict_order": (ci, [vp, ci]),
        "jamd_beam_set_order_mode": (ci, [vp, ci]),
        "jamd_beam_order_mode": (ci, [vp]),
        "jamd_beam_set_workgroup_shape": (ci, [vp, ci]),
        "jamd_beam_workgroup_shape": (ci, [vp, ci]),
        "jamd_beam_exact_layout": (ci, [vp]),
        "jamd_beam_wait_started": (ci, [vp]),
        "jamd_beam_stream_wait_resident": (ci, [vp, vp]),
        "jamd_beam_debug_preset_resident": (ci, [vp, C.c_uint]),
        "jamd_beam_debug_resident": (ci, [vp, P(C.c_uint), P(C.c_uint)]),
        "jamd_beam_prune_order": (ci, [vp, vp, ci, vp, P(ci)]),
        "jamd_beam_prune_arrange": (ci, [vp, vp, ci, vp, P(ci), vp]),
        "jamd_beam_prune_info": (ci, [vp, P(ci), P(ci), P(ci)]),
        "jamd_beam_prune_stats": (ci, [vp, ci, vp, ci]),
        "jamd_beam_stream_begin": (ci, [vp, ci]),
        "jamd_beam_stream_push_dev": (ci, [vp, vp, ci, vp, ci, ci, vp]),
        "jamd_beam_trellis": (ci, [vp, ci, vp, ci, P(ci)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def declared_symbols():
    """Every `jamd_*` function declared in include/julius_amd.h."""
    import re
    hdr = (_PKG.parent / "include" / "julius_amd.h").read_text()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(jamd_[a-z0-9_]+)\s*\(", hdr)))


def _check(rc, what):
    if rc != 0:
        raise JamdError(f"{what} failed ({rc}): {load().jamd_last_error().decode()}")


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


class Engine:
    def __init__(self, device: int = 0):
        lib = load()
        h = C.c_void_p()
        _check(lib.jamd_engine_create(device, C.byref(h)), "jamd_engine_create")
        self.h = h
        self.device = device

    def sync(self):
        _check(load().jamd_engine_sync(self.h), "jamd_engine_sync")

    def close(self):
        if getattr(self, "h", None):
            load().jamd_engine_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DevBuf:
    """A raw device allocation through the C ABI (jamd_malloc / jamd_free)."""

    def __init__(self, eng: "Engine", nbytes: int):
        self.eng, self.nbytes = eng, int(nbytes)
        p = C.c_void_p()
        _check(load().jamd_malloc(eng.h, self.nbytes, C.byref(p)), "jamd_malloc")
        self.ptr = p.value

    def upload(self, a: np.ndarray):
        a = np.ascontiguousarray(a)
        assert a.nbytes <= self.nbytes
        _check(load().jamd_memcpy_h2d(self.eng.h, self.ptr, a.ctypes.data, a.nbytes), "jamd_memcpy_h2d")
        return self

    def download(self, shape, dtype):
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes
        _check(load().jamd_memcpy_d2h(self.eng.h, out.ctypes.data, self.ptr, out.nbytes), "jamd_memcpy_d2h")
        return out

    def free(self):
        if getattr(self, "ptr", None):
            load().jamd_free(self.eng.h, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Gmm:
    """Device-resident flattened GMM acoustic model (jamd_gmm)."""

    def __init__(self, eng: Engine, model: dict, gprune: int = GPRUNE_NONE, gprune_num: int = 0):
        lib = load()
        self.eng = eng
        self._keep = {
            "mean": _f32(model["mean"]), "ivar": _f32(model["ivar"]), "gconst": _f32(model["gconst"]),
            "st_off": _i32(model["st_off"]), "ent_dens": _i32(model["ent_dens"]),
            "ent_logw": _f32(model["ent_logw"]),
        }
        st_book = model.get("st_book")
        nbook = int(model.get("nbook", 0))
        if st_book is not None:
            self._keep["st_book"] = _i32(st_book)
        d = self._desc = GmmDesc()
        d.nstate = len(self._keep["st_off"]) - 1
        d.veclen = self._keep["mean"].shape[1]
        d.ndens = self._keep["mean"].shape[0]
        d.nentry = len(self._keep["ent_dens"])
        d.nbook = nbook
        d.nstream = int(model.get("nstream", 1))
        for k in ("mean", "ivar", "gconst", "st_off", "ent_dens", "ent_logw"):
            setattr(d, k, self._keep[k].ctypes.data)
        d.st_book = self._keep["st_book"].ctypes.data if "st_book" in self._keep else None
        h = C.c_void_p()
        _check(lib.jamd_gmm_create(eng.h, C.byref(d), gprune, gprune_num, C.byref(h)), "jamd_gmm_create")
        self.h = h
        self.S, self.D = d.nstate, d.veclen
        self.nbook, self.gprune_num = nbook, gprune_num

    @classmethod
    def from_file(cls, eng: Engine, path, gprune: int = GPRUNE_NONE, gprune_num: int = 0):
        """jamd_gmm_load(): a JAMDGMM1 blob read by the library itself."""
        lib = load()
        self = cls.__new__(cls)
        self.eng, self._keep = eng, {}
        h = C.c_void_p()
        _check(lib.jamd_gmm_load(eng.h, str(path).encode(), gprune, gprune_num, C.byref(h)), "jamd_gmm_load")
        self.h = h
        self.S, self.D = lib.jamd_gmm_nstate(h), lib.jamd_gmm_veclen(h)
        self.nbook, self.gprune_num = lib.jamd_gmm_nbook(h), gprune_num
        return self

    @classmethod
    def from_binhmm(cls, eng: Engine, path, gprune: int = GPRUNE_NONE, gprune_num: int = 0):
        """jamd_gmm_load_binhmm(): Julius' binary HMM definition read by the library itself."""
        lib = load()
        self = cls.__new__(cls)
        self.eng, self._keep = eng, {}
        h = C.c_void_p()
        _check(lib.jamd_gmm_load_binhmm(eng.h, str(path).encode(), gprune, gprune_num, C.byref(h)), "jamd_gmm_load_binhmm")
        self.h = h
        self.S, self.D = lib.jamd_gmm_nstate(h), lib.jamd_gmm_veclen(h)
        self.nbook, self.gprune_num = lib.jamd_gmm_nbook(h), gprune_num
        return self

    def outprob_host(self, frames: np.ndarray) -> np.ndarray:
        fr = _f32(frames)
        T = fr.shape[0]
        assert fr.ndim == 2 and fr.shape[1] == self.D
        out = np.empty((T, self.S), dtype=np.float32)
        _check(load().jamd_gmm_outprob_host(self.h, fr.ctypes.data, T, out.ctypes.data),
               "jamd_gmm_outprob_host")
        return out

    def outprob_dev(self, dev_frames: int, T: int, dev_out: int, stream: int = 0):
        _check(load().jamd_gmm_outprob_dev(self.h, dev_frames, T, dev_out, stream or None),
               "jamd_gmm_outprob_dev")

    def outprob_utts_dev(self, dev_frames: int, utt_off, dev_out: int, stream: int = 0):
        """A batch of utterances back to back (the boundaries matter to gprune heu / beam over tied-mixture codebooks)."""
        off = _i32(utt_off)
        _check(load().jamd_gmm_outprob_utts_dev(self.h, dev_frames, off.ctypes.data, len(off) - 1, dev_out, stream or None),
               "jamd_gmm_outprob_utts_dev")

    def dens_host(self, frames: np.ndarray) -> np.ndarray:
        """Per-Gaussian scores [T][nentry] (the plugin slot's compute_gaussset values)."""
        fr = _f32(frames)
        E = load().jamd_gmm_nentry(self.h)
        out = np.empty((fr.shape[0], E), dtype=np.float32)
        _check(load().jamd_gmm_dens_host(self.h, fr.ctypes.data, fr.shape[0], out.ctypes.data), "jamd_gmm_dens_host")
        return out

    def last_kernel(self) -> str:
        return load().jamd_gmm_last_kernel(self.h).decode()

    def tmix_cache_host(self, frames: np.ndarray):
        """Codebook top-N cache (MIXCACHE) for every (frame, book): score, id, num."""
        lib = load()
        fr = _f32(frames)
        T = fr.shape[0]
        cap, nbook = lib.jamd_gmm_tmix_cap(self.h), lib.jamd_gmm_nbook(self.h)
        d_fr = DevBuf(self.eng, fr.nbytes).upload(fr)
        d_sc = DevBuf(self.eng, 4 * T * nbook * cap)
        d_id = DevBuf(self.eng, 4 * T * nbook * cap)
        d_n = DevBuf(self.eng, 4 * T * nbook)
        _check(lib.jamd_gmm_tmix_cache_dev(self.h, d_fr.ptr, T, d_sc.ptr, d_id.ptr, d_n.ptr, None),
               "jamd_gmm_tmix_cache_dev")
        self.eng.sync()
        return (d_sc.download((T, nbook, cap), np.float32), d_id.download((T, nbook, cap), np.int32),
                d_n.download((T, nbook), np.int32))

    def close(self):
        if getattr(self, "h", None):
            load().jamd_gmm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

class Gms:
    """Gaussian mixture selection stage (jamd_gms): gms_state() of libsent/src/phmm/gms.c."""

    def __init__(self, eng: Engine, gs_model: dict, state2gs, nbest: int):
        lib = load()
        self.eng = eng
        self._keep = {k: (_i32 if k in ("st_off", "ent_dens") else _f32)(gs_model[k])
                      for k in ("mean", "ivar", "gconst", "st_off", "ent_dens", "ent_logw")}
        d = GmmDesc()
        d.nstate = len(self._keep["st_off"]) - 1
        d.veclen = self._keep["mean"].shape[1]
        d.ndens = self._keep["mean"].shape[0]
        d.nentry = len(self._keep["ent_dens"])
        d.nbook, d.nstream, d.st_book = 0, 1, None
        for k in self._keep:
            setattr(d, k, self._keep[k].ctypes.data)
        self.state2gs = _i32(state2gs)
        self.S, self.D, self.nbest = len(self.state2gs), d.veclen, int(nbest)
        h = C.c_void_p()
        _check(lib.jamd_gms_create(eng.h, C.byref(d), self.state2gs.ctypes.data, self.S, self.nbest, C.byref(h)),
               "jamd_gms_create")
        self.h = h

    @classmethod
    def from_file(cls, eng: Engine, path, veclen: int):
        """jamd_gms_load(): the selection model file jamd_export writes next to PREFIX.am."""
        self = cls.__new__(cls)
        self.eng, self._keep = eng, {}
        h = C.c_void_p()
        _check(load().jamd_gms_load(eng.h, str(path).encode(), C.byref(h)), "jamd_gms_load")
        self.h = h
        self.S, self.D = load().jamd_gms_nstate(h), int(veclen)
        return self

    def set_strict_order(self, on: bool = True):
        _check(load().jamd_gms_set_strict_order(self.h, int(on)), "jamd_gms_set_strict_order")
        return self

    def apply_dev(self, dev_frames: int, T: int, dev_scores: int, utt_off=None, stream: int = 0):
        if utt_off is not None:
            utt_off = _i32(utt_off)
            _check(load().jamd_gms_apply_dev(self.h, dev_frames, T, utt_off.ctypes.data, len(utt_off) - 1,
                                             dev_scores, stream or None), "jamd_gms_apply_dev")
        else:
            _check(load().jamd_gms_apply_dev(self.h, dev_frames, T, None, 0, dev_scores, stream or None),
                   "jamd_gms_apply_dev")

    def apply_host(self, frames: np.ndarray, scores: np.ndarray, utt_off=None) -> np.ndarray:
        fr, sc = _f32(frames), _f32(scores)
        T = fr.shape[0]
        assert sc.shape == (T, self.S) and fr.shape[1] == self.D
        out = sc.copy()
        off = _i32(utt_off) if utt_off is not None else None
        _check(load().jamd_gms_apply_host(self.h, fr.ctypes.data, T, off.ctypes.data if off is not None else None,
                                          len(off) - 1 if off is not None else 0, out.ctypes.data), "jamd_gms_apply_host")
        return out

    def close(self):
        if getattr(self, "h", None):
            load().jamd_gms_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

class RejGmm:
    """GMM-based input verification scores (jamd_rejgmm): gmm_proceed() of libjulius/src/gmm.c."""

    def __init__(self, eng: Engine, model: dict, model_state, gprune_num: int = 10):
        lib = load()
        self.eng = eng
        self._keep = {k: (_i32 if k in ("st_off", "ent_dens") else _f32)(model[k])
                      for k in ("mean", "ivar", "gconst", "st_off", "ent_dens", "ent_logw")}
        d = GmmDesc()
        d.nstate = len(self._keep["st_off"]) - 1
        d.veclen = self._keep["mean"].shape[1]
        d.ndens = self._keep["mean"].shape[0]
        d.nentry = len(self._keep["ent_dens"])
        d.nbook, d.nstream, d.st_book = 0, 1, None
        for k in self._keep:
            setattr(d, k, self._keep[k].ctypes.data)
        ms = _i32(model_state)
        self.nmodel, self.D = len(ms), d.veclen
        h = C.c_void_p()
        _check(lib.jamd_rejgmm_create(eng.h, C.byref(d), ms.ctypes.data, self.nmodel, int(gprune_num), C.byref(h)),
               "jamd_rejgmm_create")
        self.h = h

    def scores_host(self, frames: np.ndarray, utt_off=None):
        """(frame scores [T][nmodel], utterance sums [nutt][nmodel]); one utterance by default."""
        fr = _f32(frames)
        T = fr.shape[0]
        assert fr.ndim == 2 and fr.shape[1] == self.D
        off = _i32(utt_off if utt_off is not None else [0, T])
        fs = np.empty((T, self.nmodel), np.float32)
        us = np.empty((len(off) - 1, self.nmodel), np.float32)
        _check(load().jamd_rejgmm_scores_host(self.h, fr.ctypes.data, T, off.ctypes.data, len(off) - 1,
                                              fs.ctypes.data, us.ctypes.data), "jamd_rejgmm_scores_host")
        return fs, us

    def close(self):
        if getattr(self, "h", None):
            load().jamd_rejgmm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass



class CdSet:
    """Pseudo-phone state sets (jamd_cdset): outprob_cd() over a score matrix."""

    def __init__(self, eng: Engine, set_off, states, method=IWCD_MAX, nbest=3):
        self.eng = eng
        self.set_off, self.states = _i32(set_off), _i32(states)
        self.nset = len(self.set_off) - 1
        h = C.c_void_p()
        _check(load().jamd_cdset_create(eng.h, self.nset, self.set_off.ctypes.data, self.states.ctypes.data,
                                        method, nbest, C.byref(h)), "jamd_cdset_create")
        self.h = h

    def outprob_dev(self, dev_scores: int, T: int, nstate: int, dev_cd: int, stream: int = 0):
        _check(load().jamd_cdset_outprob_dev(self.h, dev_scores, T, nstate, dev_cd, stream or None),
               "jamd_cdset_outprob_dev")

    def outprob_host(self, scores: np.ndarray) -> np.ndarray:
        sc = _f32(scores)
        T, S = sc.shape
        d_sc = DevBuf(self.eng, sc.nbytes).upload(sc)
        d_cd = DevBuf(self.eng, 4 * T * max(self.nset, 1))
        self.outprob_dev(d_sc.ptr, T, S, d_cd.ptr)
        self.eng.sync()
        return d_cd.download((T, self.nset), np.float32)

    def close(self):
        if getattr(self, "h", None):
            load().jamd_cdset_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Dnn:
    """Device-resident DNN acoustic model (jamd_dnn)."""

    def __init__(self, eng: Engine, dnn: dict):
        self.eng = eng
        self.dims = _i32(dnn["dims"])
        nl = len(self.dims) - 1
        self._w = [_f32(w) for w in dnn["w"]]
        self._b = [_f32(b) for b in dnn["b"]]
        for l in range(nl):
            assert self._w[l].shape == (self.dims[l + 1], self.dims[l]), "W[l] must be [out][in]"
        self._wp = (C.c_void_p * nl)(*[w.ctypes.data for w in self._w])
        self._bp = (C.c_void_p * nl)(*[b.ctypes.data for b in self._b])
        self._prior = _f32(dnn["prior"])
        d = DnnDesc()
        d.nlayer = nl
        d.dims = self.dims.ctypes.data
        d.w = C.cast(self._wp, C.c_void_p)
        d.b = C.cast(self._bp, C.c_void_p)
        d.state_prior = self._prior.ctypes.data
        h = C.c_void_p()
        _check(load().jamd_dnn_create(eng.h, C.byref(d), C.byref(h)), "jamd_dnn_create")
        self.h = h
        self.S, self.D = int(self.dims[-1]), int(self.dims[0])

    @classmethod
    def from_dnnconf(cls, eng: Engine, path):
        """jamd_dnn_load(): Julius' dnnconf + .npy + prior files read by the library itself."""
        lib = load()
        self = cls.__new__(cls)
        self.eng = eng
        h = C.c_void_p()
        _check(lib.jamd_dnn_load(eng.h, str(path).encode(), C.byref(h)), "jamd_dnn_load")
        self.h = h
        self.S, self.D = lib.jamd_dnn_nstate(h), lib.jamd_dnn_veclen(h)
        return self

    def outprob_host(self, frames: np.ndarray) -> np.ndarray:
        fr = _f32(frames)
        T = fr.shape[0]
        assert fr.shape[1] == self.D
        out = np.empty((T, self.S), dtype=np.float32)
        _check(load().jamd_dnn_outprob_host(self.h, fr.ctypes.data, T, out.ctypes.data), "jamd_dnn_outprob_host")
        return out

    def outprob_dev(self, dev_frames: int, T: int, dev_out: int, stream: int = 0):
        _check(load().jamd_dnn_outprob_dev(self.h, dev_frames, T, dev_out, stream or None),
               "jamd_dnn_outprob_dev")

    def close(self):
        if getattr(self, "h", None):
            load().jamd_dnn_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Pass1Result(C.Structure):
    """jamd_pass1_result (include/julius_amd.h)."""
    _fields_ = [("status", C.c_int), ("natom", C.c_int), ("wnum", C.c_int), ("score", C.c_float),
                ("died_at", C.c_int), ("ties", C.c_int), ("frames", C.c_int), ("max_tokens", C.c_int),
                ("ties_node", C.c_int), ("ties_wordend", C.c_int), ("ties_cut", C.c_int),
                ("phase_us", C.c_int * 8),
                ("wseq", C.c_int * 150)]


class Lexicon:
    """Device-resident first-pass tables (jamd_lexicon) from a lexicon dict
    (julius_amd.lexblob.load)."""

    def __init__(self, eng: Engine, lex: dict):
        from . import lexblob
        self.eng, self.lex = eng, lex
        d, self._keep = lexblob.make_desc(lex)
        h = C.c_void_p()
        _check(load().jamd_lexicon_create(eng.h, C.byref(d), C.byref(h)), "jamd_lexicon_create")
        self.h = h

    @classmethod
    def from_file(cls, eng: Engine, path, bingram=None):
        """jamd_lexicon_load(): a JAMDLEX1 blob read by the library itself; with `bingram` the N-gram half comes from
        that binary N-gram file (jamd_lexicon_load_ngram())."""
        self = cls.__new__(cls)
        self.eng, self.lex, self._keep = eng, None, None
        h = C.c_void_p()
        if bingram is not None:
            _check(load().jamd_lexicon_load_ngram(eng.h, str(path).encode(), str(bingram).encode(), C.byref(h)), "jamd_lexicon_load_ngram")
        else:
            _check(load().jamd_lexicon_load(eng.h, str(path).encode(), C.byref(h)), "jamd_lexicon_load")
        self.h = h
        return self

    def close(self):
        if getattr(self, "h", None):
            load().jamd_lexicon_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Beam:
    """First-pass work area for a batch of utterances (jamd_beam)."""

    def __init__(self, eng: Engine, lexicon: Lexicon, beam_width: int, score_pruning_width: float = -1.0,
                 max_utts: int = 1, atoms_per_utt: int = 1 << 16):
        self.eng, self.lexicon = eng, lexicon
        self.max_utts, self.atoms_per_utt = max_utts, atoms_per_utt
        h = C.c_void_p()
        _check(load().jamd_beam_create(eng.h, lexicon.h, beam_width, score_pruning_width, max_utts,
                                       atoms_per_utt, C.byref(h)), "jamd_beam_create")
        self.h = h
        # test knob: JAMD_TEST_SHAPE=half runs every work area that can in the half workgroup shape (the results must
        # not depend on it), so the whole GPU suite can be replayed over that shape
        if os.environ.get("JAMD_TEST_SHAPE") == "half":
            load().jamd_beam_set_workgroup_shape(self.h, 2)

    def set_strict_order(self, on: bool = True):
        _check(load().jamd_beam_set_strict_order(self.h, 1 if on else 0), "jamd_beam_set_strict_order")

    ORDER_MODES = {"fast": 0, "strict": 1, "exact": 2, "exact_serial": 3}

    def set_order_mode(self, mode):
        """'exact' (default where available), 'exact_serial', 'fast' or 'strict' -- see julius_amd.h."""
        m = self.ORDER_MODES[mode] if isinstance(mode, str) else int(mode)
        _check(load().jamd_beam_set_order_mode(self.h, m), "jamd_beam_set_order_mode")
        return self

    def order_mode(self) -> str:
        m = load().jamd_beam_order_mode(self.h)
        return {v: k for k, v in self.ORDER_MODES.items()}[m]

    SHAPES = {"auto": 0, "full": 1, "half": 2}

    def set_workgroup_shape(self, shape):
        """'auto' (default), 'full' (one utterance per CU) or 'half' (two per CU) -- see julius_amd.h."""
        m = self.SHAPES[shape] if isinstance(shape, str) else int(shape)
        _check(load().jamd_beam_set_workgroup_shape(self.h, m), "jamd_beam_set_workgroup_shape")
        return self

    def workgroup_shape(self, nutt: int = 1) -> str:
        m = load().jamd_beam_workgroup_shape(self.h, nutt)
        return {v: k for k, v in self.SHAPES.items()}[m]

    def wait_started(self):
        """Host waits until the latest first-pass launch is next to run (jamd_beam_wait_started)."""
        _check(load().jamd_beam_wait_started(self.h), "jamd_beam_wait_started")

    def stream_wait_resident(self, stream: int = 0):
        """Work queued on `stream` behind this call starts once the latest first-pass launch holds its CUs
        (jamd_beam_stream_wait_resident: a wait on device memory, no host involvement)."""
        _check(load().jamd_beam_stream_wait_resident(self.h, stream), "jamd_beam_stream_wait_resident")

    def debug_preset_resident(self, count: int):
        _check(load().jamd_beam_debug_preset_resident(self.h, count), "jamd_beam_debug_preset_resident")

    def debug_resident(self):
        a, b = C.c_uint(), C.c_uint()
        _check(load().jamd_beam_debug_resident(self.h, C.byref(a), C.byref(b)), "jamd_beam_debug_resident")
        return int(a.value), int(b.value)

    def exact_layout(self) -> str:
        """'narrow' / 'wide' LDS image of the exact-order kernel for this work area, or 'none' (jamd_beam_exact_layout)."""
        return {0: "none", 1: "narrow", 2: "wide"}[load().jamd_beam_exact_layout(self.h)]

    def prune_order(self, scores):
        """sort_token_no_order() alone: the visiting order the exact-order kernel derives for tokens with
        these scores (creation order) under this work area's beam width."""
        sc = _f32(scores)
        out = np.zeros(len(sc), np.int32)
        n = C.c_int()
        _check(load().jamd_beam_prune_order(self.h, sc.ctypes.data, len(sc), out.ctypes.data, C.byref(n)),
               "jamd_beam_prune_order")
        return out[:n.value].copy()

    def prune_arrange(self, scores):
        """sort_token_no_order() with the whole array out: (visiting order, tindex[0..n)) -- jamd_beam_prune_arrange()."""
        sc = _f32(scores)
        out = np.zeros(len(sc), np.int32)
        arr = np.zeros(len(sc), np.int32)
        n = C.c_int()
        _check(load().jamd_beam_prune_arrange(self.h, sc.ctypes.data, len(sc), out.ctypes.data, C.byref(n), arr.ctypes.data),
               "jamd_beam_prune_arrange")
        return out[:n.value].copy(), arr

    def prune_info(self):
        """Rounds of the sweep replay in the latest prune_order() call (-1 = it gave the frame up, 0 = not used)."""
        r, us, ne = C.c_int(), C.c_int(), C.c_int()
        _check(load().jamd_beam_prune_info(self.h, C.byref(r), C.byref(us), C.byref(ne)), "jamd_beam_prune_info")
        self.last_sweep_us, self.last_sweep_events = us.value, ne.value
        return r.value

    def prune_stats(self, utt=0, reset=False):
        """Frames of utterance `utt` by the path their rank pruning step took (jamd_beam_prune_stats)."""
        st = np.zeros(16, np.int32)
        _check(load().jamd_beam_prune_stats(self.h, utt, st.ctypes.data, 1 if reset else 0), "jamd_beam_prune_stats")
        return [int(x) for x in st]

    def stream_begin(self, nutt: int):
        self._nutt = nutt
        _check(load().jamd_beam_stream_begin(self.h, nutt), "jamd_beam_stream_begin")

    def stream_push_dev(self, dev_scores: int, nstate: int, chunk_off, final: bool = False, stream: int = 0):
        off = _i32(chunk_off)
        _check(load().jamd_beam_stream_push_dev(self.h, dev_scores or None, nstate, off.ctypes.data, len(off) - 1,
                                                1 if final else 0, stream or None), "jamd_beam_stream_push_dev")

    def pass1_dev(self, dev_scores: int, nstate: int, utt_off, stream: int = 0):
        off = _i32(utt_off)
        self._nutt = len(off) - 1
        _check(load().jamd_beam_pass1_dev(self.h, dev_scores, nstate, off.ctypes.data, self._nutt,
                                          stream or None), "jamd_beam_pass1_dev")

    def results(self, nutt=None):
        n = self._nutt if nutt is None else nutt
        arr = (Pass1Result * n)()
        _check(load().jamd_beam_results(self.h, arr, n), "jamd_beam_results")
        return list(arr)

    def trellis(self, utt: int):
        from . import lexblob
        n = C.c_int()
        _check(load().jamd_beam_trellis(self.h, utt, None, 0, C.byref(n)), "jamd_beam_trellis")
        atoms = np.zeros(max(n.value, 1), dtype=lexblob.ATOM_DTYPE)
        _check(load().jamd_beam_trellis(self.h, utt, atoms.ctypes.data, n.value, C.byref(n)), "jamd_beam_trellis")
        return atoms[:n.value]

    def pass1_host(self, score_list):
        """Convenience for tests: list of [T_u][S] host score matrices -> (results, trellises)."""
        S = score_list[0].shape[1]
        off = np.zeros(len(score_list) + 1, np.int32)
        off[1:] = np.cumsum([len(x) for x in score_list])
        allsc = _f32(np.concatenate(score_list, axis=0))
        d = DevBuf(self.eng, allsc.nbytes).upload(allsc)
        self.pass1_dev(d.ptr, S, off)
        res = self.results()
        tre = [self.trellis(u) for u in range(len(score_list))]
        d.free()
        return res, tre

    def close(self):
        if getattr(self, "h", None):
            load().jamd_beam_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
