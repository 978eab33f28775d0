"""Reader/writer of the first-pass lexicon blob ("JAMDLEX1").

The blob is what julius_amd/shim/jamd_flatten_lex.c::jamd_lexicon_save() writes
from a loaded, unmodified Julius recogniser: every array of
include/julius_amd.h::jamd_lexicon_desc as a named record.  Host plumbing only.
"""
from __future__ import annotations

import struct
from pathlib import Path

import numpy as np

_DT = {0: np.int32, 1: np.float32, 2: np.uint8}
_INTS = ["nnode", "nword", "startnum", "isolatenum", "nlc", "nlcrow", "nset", "cdset_method", "cdmax_num",
         "head_silwid", "tail_silwid", "nfscore", "nscword", "ng_mode", "ng_nword", "ng_nbigram", "ng_unk_id",
         "_reserved", "lm_type", "ncat", "ninit", "nfwd"]
_FLOATS = ["ng_unk_num_log", "lm_weight", "lm_penalty", "lm_penalty_trans", "penalty1"]
DFA_ARRAYS = ["cat_pair", "start2wid", "init_node", "init_lscore"]
FWD_ARRAYS = ["fwd_off", "fwd_label", "fwd_to", "init_to_state"]      # only with a forward DFA (nfwd > 0)
ARRAYS = ["self_a", "next_a", "ac_off", "ac_to", "ac_a", "stend", "scid", "out_kind", "out_id", "lc_tab",
          "word_lc", "set_off", "set_states", "startnode", "start2isolate", "wordend_a", "wton", "cprob",
          "is_transparent", "word_head", "fscore", "scword", "ng_uni_prob", "ng_uni_bo", "ng_bi_bgn",
          "ng_bi_num", "ng_bi_wid", "ng_bi_prob"]


def load(path) -> dict:
    raw = Path(path).read_bytes()
    if raw[:8] != b"JAMDLEX1":
        raise ValueError(f"{path}: not a JAMDLEX1 blob")
    (nrec,) = struct.unpack_from("<i", raw, 8)
    pos, rec = 12, {}
    for _ in range(nrec):
        name = raw[pos:pos + 24].split(b"\0", 1)[0].decode()
        dtype, count = struct.unpack_from("<ii", raw, pos + 24)
        pos += 32
        dt = np.dtype(_DT[dtype])
        nbytes = count * dt.itemsize
        rec[name] = np.frombuffer(raw, dtype=dt, count=count, offset=pos).copy()
        pos += (nbytes + 3) & ~3
    lex = {k: int(v) for k, v in zip(_INTS, rec.pop("ints"))}
    lex.update({k: float(np.float32(v)) for k, v in zip(_FLOATS, rec.pop("floats"))})
    lex.pop("_reserved", None)
    lex.update(rec)
    return lex


def load_gmm(path) -> dict:
    """Acoustic-model blob written by jamd_gmm_save() -> the model dict lib.Gmm takes."""
    raw = Path(path).read_bytes()
    if raw[:8] != b"JAMDGMM1":
        raise ValueError(f"{path}: not a JAMDGMM1 blob")
    (nrec,) = struct.unpack_from("<i", raw, 8)
    pos, rec = 12, {}
    for _ in range(nrec):
        name = raw[pos:pos + 24].split(b"\0", 1)[0].decode()
        dtype, count = struct.unpack_from("<ii", raw, pos + 24)
        pos += 32
        dt = np.dtype(_DT[dtype])
        rec[name] = np.frombuffer(raw, dtype=dt, count=count, offset=pos).copy()
        pos += (count * dt.itemsize + 3) & ~3
    S, D, G, E, nbook, nstream = (int(x) for x in rec.pop("ints"))
    extra = dict(state2gs=rec["state2gs"], nbest=int(rec["gms"][0])) if "gms" in rec else {}
    if "rej" in rec:                       # verification GMMs (jamd_rejgmm_save)
        extra.update(model_state=rec["model_state"], gprune_num=int(rec["rej"][0]), is_voice=rec["is_voice"],
                     model_names=[n.decode() for n in rec["model_names"].tobytes().split(b"\0")[:len(rec["model_state"])]])
    return dict(**extra, mean=rec["mean"].reshape(G, D), ivar=rec["ivar"].reshape(G, D), gconst=rec["gconst"],
                st_off=rec["st_off"], ent_dens=rec["ent_dens"], ent_logw=rec["ent_logw"],
                st_book=rec.get("st_book") if nbook > 0 else None, nbook=nbook, nstream=nstream)


def save_gmm(model: dict, path, state2gs=None, nbest: int = 0) -> None:
    """Same format as jamd_gmm_save() (julius_amd/shim/jamd_flatten.c); with state2gs and nbest the
    selection-model file of jamd_gms_save()."""
    mean = np.ascontiguousarray(model["mean"], np.float32)
    st_book = model.get("st_book")
    recs = [("ints", np.array([len(model["st_off"]) - 1, mean.shape[1], mean.shape[0], len(model["ent_dens"]),
                               int(model.get("nbook", 0)), int(model.get("nstream", 1))], np.int32)),
            ("mean", mean), ("ivar", np.asarray(model["ivar"], np.float32)), ("gconst", np.asarray(model["gconst"], np.float32)),
            ("st_off", np.asarray(model["st_off"], np.int32)), ("ent_dens", np.asarray(model["ent_dens"], np.int32)),
            ("ent_logw", np.asarray(model["ent_logw"], np.float32))]
    if st_book is not None:
        recs.append(("st_book", np.asarray(st_book, np.int32)))
    if state2gs is not None:
        recs += [("state2gs", np.asarray(state2gs, np.int32)), ("gms", np.array([nbest], np.int32))]
    out = [b"JAMDGMM1", struct.pack("<i", len(recs))]
    for name, arr in recs:
        arr = np.ascontiguousarray(arr)
        out.append(name.encode().ljust(24, b"\0") + struct.pack("<ii", 0 if arr.dtype == np.int32 else 1, arr.size) + arr.tobytes())
    Path(path).write_bytes(b"".join(out))


def save(lex: dict, path) -> None:
    """Same format as jamd_lexicon_save() (used to commit small golden fixtures)."""
    isdfa = lex.get("lm_type", 0) & 0xff != 0
    out = [b"JAMDLEX1", struct.pack("<i", 2 + len(ARRAYS) + (len(DFA_ARRAYS) if isdfa else 0) +
                                    (len(FWD_ARRAYS) if isdfa and lex.get("nfwd", 0) > 0 else 0))]

    def put(name, arr):
        arr = np.ascontiguousarray(arr)
        code = {np.dtype(np.int32): 0, np.dtype(np.float32): 1, np.dtype(np.uint8): 2}[arr.dtype]
        b = arr.tobytes()
        out.append(name.encode().ljust(24, b"\0") + struct.pack("<ii", code, arr.size) + b + b"\0" * (-len(b) % 4))

    put("ints", np.array([lex.get(k, 0) for k in _INTS], dtype=np.int32))
    put("floats", np.array([lex.get(k, 0.0) for k in _FLOATS], dtype=np.float32))
    for k in ARRAYS:
        put(k, lex[k])
    if lex.get("lm_type", 0) & 0xff != 0:
        for k in DFA_ARRAYS:
            put(k, lex[k])
        if lex.get("nfwd", 0) > 0:
            for k in FWD_ARRAYS:
                put(k, lex[k])
    Path(path).write_bytes(b"".join(out))


# ---- ctypes mirror of include/julius_amd.h::jamd_lexicon_desc ------------------
import ctypes as _C

_vp, _ci, _cf = _C.c_void_p, _C.c_int, _C.c_float


class LexiconDesc(_C.Structure):
    _fields_ = [
        ("nnode", _ci), ("nword", _ci), ("startnum", _ci), ("isolatenum", _ci),
        ("self_a", _vp), ("next_a", _vp), ("ac_off", _vp), ("ac_to", _vp), ("ac_a", _vp),
        ("stend", _vp), ("scid", _vp), ("out_kind", _vp), ("out_id", _vp),
        ("nlc", _ci), ("nlcrow", _ci), ("lc_tab", _vp), ("word_lc", _vp),
        ("nset", _ci), ("set_off", _vp), ("set_states", _vp), ("cdset_method", _ci), ("cdmax_num", _ci),
        ("startnode", _vp), ("start2isolate", _vp),
        ("wordend_a", _vp), ("wton", _vp), ("cprob", _vp), ("is_transparent", _vp), ("word_head", _vp),
        ("head_silwid", _ci), ("tail_silwid", _ci),
        ("nfscore", _ci), ("nscword", _ci), ("fscore", _vp), ("scword", _vp),
        ("ng_mode", _ci), ("ng_nword", _ci), ("ng_nbigram", _ci), ("ng_unk_id", _ci), ("ng_unk_num_log", _cf),
        ("ng_uni_prob", _vp), ("ng_uni_bo", _vp), ("ng_bi_bgn", _vp), ("ng_bi_num", _vp), ("ng_bi_wid", _vp),
        ("ng_bi_prob", _vp),
        ("lm_weight", _cf), ("lm_penalty", _cf), ("lm_penalty_trans", _cf),
        ("lm_type", _ci), ("ncat", _ci), ("cat_pair", _vp), ("start2wid", _vp),
        ("ninit", _ci), ("init_node", _vp), ("init_lscore", _vp), ("penalty1", _cf),
        ("nfwd", _ci), ("fwd_off", _vp), ("fwd_label", _vp), ("fwd_to", _vp), ("init_to_state", _vp),
    ]


class TrellisAtom(_C.Structure):
    _fields_ = [("wid", _ci), ("last_tre", _ci), ("backscore", _cf), ("lscore", _cf),
                ("begintime", _C.c_short), ("endtime", _C.c_short)]


ATOM_DTYPE = np.dtype([("wid", "<i4"), ("last_tre", "<i4"), ("backscore", "<f4"), ("lscore", "<f4"),
                       ("begintime", "<i2"), ("endtime", "<i2")])


def make_desc(lex: dict):
    """(LexiconDesc, keepalive) for a lexicon dict; arrays are made contiguous and
    of the exact dtype first and must stay referenced while the desc is in use."""
    keep = {}
    d = LexiconDesc()
    scalars = {n for n, t in LexiconDesc._fields_ if t is not _vp}
    dflt = {"lm_type": 0, "ncat": 0, "ninit": 0, "penalty1": 0.0, "cat_pair": np.zeros(1, np.uint8),
            "start2wid": np.zeros(1, np.int32), "init_node": np.zeros(1, np.int32), "init_lscore": np.zeros(1, np.float32),
            "nfwd": 0, "fwd_off": np.zeros(1, np.int32), "fwd_label": np.zeros(1, np.int32), "fwd_to": np.zeros(1, np.int32),
            "init_to_state": np.zeros(1, np.int32)}
    for name, ctype in LexiconDesc._fields_:
        val = lex.get(name, dflt.get(name))           # fixtures written before grammar mode lack the DFA fields
        if name in scalars:
            setattr(d, name, val)
        else:
            want = {"out_kind": np.uint8, "is_transparent": np.uint8, "cat_pair": np.uint8}.get(
                name, np.float32 if np.asarray(val).dtype.kind == "f" else np.int32)
            a = np.ascontiguousarray(val, dtype=want)
            if a.size == 0:
                a = np.zeros(1, dtype=want)
            keep[name] = a
            setattr(d, name, a.ctypes.data)
    return d, keep


def canonical_trellis(atoms: np.ndarray):
    """Sort trellis atoms the way bt_relocate_rw()+bt_sort_rw() index them
    (libjulius/src/backtrellis.c:218-267, 468-477): by end frame, then word id;
    predecessor links become (wid, endtime) pairs so two runs can be compared
    independently of emission order."""
    n = len(atoms)
    lt = atoms["last_tre"]
    pw = np.where(lt >= 0, atoms["wid"][np.clip(lt, 0, max(n - 1, 0))], -1) if n else np.zeros(0, np.int32)
    pe = np.where(lt >= 0, atoms["endtime"][np.clip(lt, 0, max(n - 1, 0))], -1) if n else np.zeros(0, np.int16)
    order = np.lexsort((atoms["wid"], atoms["endtime"]))
    return dict(wid=atoms["wid"][order].astype(np.int32), begintime=atoms["begintime"][order].astype(np.int32),
                endtime=atoms["endtime"][order].astype(np.int32), backscore=atoms["backscore"][order].copy(),
                lscore=atoms["lscore"][order].copy(), pwid=pw[order].astype(np.int32),
                pendtime=pe[order].astype(np.int32))
