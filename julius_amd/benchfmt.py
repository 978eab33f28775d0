"""The bench line the driver parses: bench.py's LAST stdout line, kept small.

Round 4's line nested nine full results (29 KB) and the driver could not parse it.  The full result tree now goes to
earlier stdout lines (one per nested block) and to bench_detail.json; the final line is `compact_line(full)`: the
contract keys of the top-level workload plus, per nested configuration, only the figures a reader compares
(ms_per_step, rtf_inv, steps, kernel times, roofline.frac, the CPU baseline's rtf_inv, the parity counts).
tests/test_benchfmt.py holds the size bound (< 6 KB on a recorded full result) and the round trip."""
from __future__ import annotations

import json

MAX_LINE_BYTES = 6000

_TOP_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
             "vs_baseline", "dtype", "data", "rtf_inv")
_ROOF_KEYS = ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms")
_CPU_KEYS = ("value", "unit", "cores", "kind", "rtf_inv")


def _num(x, sig=6):
    """Floats to `sig` significant digits (the line is for reading and comparing, the detail file keeps every digit)."""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    if isinstance(x, float):
        if x != x or x in (float("inf"), float("-inf")):
            return None
        return float(f"{x:.{sig}g}")
    return x


def _short(s, n):
    s = str(s)
    return s if len(s) <= n else s[:n - 1] + "~"


def _parity(par):
    """{'utts': compared, 'identical': all three of trellis / sentence / score identical, ...} from a parity block."""
    if not isinstance(par, dict):
        return None
    out = {}
    vs = par.get("device_vs_compiled_reference")
    if isinstance(vs, dict):
        out = {"utts": vs.get("utts"), "wanted": vs.get("wanted", vs.get("utts")),
               "identical": min(vs.get("trellis_identical", 0), vs.get("pass1_sentence_identical", 0), vs.get("score_identical", 0)),
               "ref_sentences": vs.get("reference_found_a_sentence")}
        if out["wanted"] != out["utts"]:
            out["incomplete"] = True
    rl = par.get("result_lines_vs_in_process")
    if isinstance(rl, dict):
        out = {"utts": rl.get("utts"), "identical": rl.get("identical"), "against": "in-process"}
    for k, v in par.items():
        if k.startswith("fast_kernel_vs_") and isinstance(v, dict):
            out["fast_kernel_identical"] = v.get("trellis_identical")
    return out or None


def _cpu(c):
    if not isinstance(c, dict):
        return None
    out = {k: _num(c[k]) for k in _CPU_KEYS if k in c}
    m = c.get("multi")
    if isinstance(m, dict) and "value" in m:
        out["multi"] = {"value": _num(m["value"]), "cores": m.get("cores"), "rtf_inv": _num(m.get("rtf_inv"))}
    return out


def _nested(r, lean=False):
    """One nested configuration: what a reader compares, nothing that only explains.  `lean` drops
    everything but the figures named there."""
    out = {k: _num(r[k]) for k in ("ms_per_step", "rtf_inv", "steps") if k in r}
    if r.get("scaling") == "strong":
        out["scaling"] = "strong"
    roof = r.get("roofline") or {}
    bound = str(roof.get("bound", "")).split(" ")[0]
    rr = {"frac": _num(roof.get("frac"), 4)}
    # the first-pass lines have no algorithmic roofline: their modelled GB/s stay in the detail file
    for k in ("kernel_ms", "beam_kernel_ms", "score_kernels_ms", "traffic") + (() if bound == "latency" else ("achieved", "peak")):
        if roof.get(k) is not None and not (lean and k in ("score_kernels_ms", "achieved", "peak")):
            rr[k] = _num(roof[k], 5)
    if not lean and bound:
        rr["bound"] = bound
    if rr.get("traffic") is not None and not lean:
        rr["traffic_source"] = "profiles/" if str(roof.get("traffic_source", "")).startswith("replayed") else "this run"
    if "timing" not in r or rr["frac"] is not None:
        out["roofline"] = rr
    if isinstance(r.get("cpu_baseline"), dict):
        c = r["cpu_baseline"]
        cb = {"rtf_inv": _num(c.get("rtf_inv"), 4), "cores": c.get("cores"), "kind": c.get("kind")}
        if not lean:
            m = c.get("multi")
            if isinstance(m, dict) and "rtf_inv" in m:
                cb["multi"] = {"cores": m.get("cores"), "rtf_inv": _num(m["rtf_inv"], 4)}
            if isinstance(c.get("two_pass"), dict):
                cb["two_pass_rtf_inv"] = _num(c["two_pass"].get("rtf_inv"), 4)
            if isinstance(c.get("threads"), dict):
                cb["threads"] = {k: _num(v, 4) for k, v in c["threads"].items() if isinstance(v, (int, float))}
        out["cpu_baseline"] = cb
    p = _parity(r.get("parity"))
    if p:
        if p.get("wanted") == p.get("utts"):
            p.pop("wanted", None)
        p.pop("fast_kernel_identical", None)
        if lean:
            p = {k: p[k] for k in ("utts", "identical", "incomplete") if k in p}
        out["parity"] = p
    if "parity_spot_check" in r:
        out["parity_spot_check"] = r["parity_spot_check"]
    p1 = r.get("pass1")
    if isinstance(p1, dict) and not lean:
        out["pass1_ok"] = f"{p1.get('ok')}/{p1.get('utts')}"
    tm = r.get("timing")
    if isinstance(tm, dict):         # the product's serving loop (jamd_batch -time)
        out["timing"] = {k: _num(tm[k], 4) for k in (("decode_s",) if lean else ("decode_s", "models_s", "host_read_s")) if k in tm}
        if "vs_e2e_same_task" in r:
            out["vs_e2e"] = _num(r["vs_e2e_same_task"], 3)
    if "strong_scaling_note" in r and not lean:
        out["note"] = "no N>1 run exists; fixed 512-utt batch: floor = longest utterance alone, ~2.2x at 8 GPUs"
    if "error" in r:
        out["error"] = _short(r["error"], 120)
    cfg = r.get("config") or {}
    if not lean:
        c2 = {k: cfg[k] for k in ("beam", "utts_per_gpu") if k in cfg}
        if "workgroup_shape" in cfg:
            c2["shape"] = str(cfg["workgroup_shape"]).split(" ")[0]
        if c2:
            out["config"] = c2
    return out


def compact_line(full: dict, lean: bool = False) -> dict:
    """The driver's line: contract keys of the top-level result + a short record per nested configuration."""
    line = {k: _num(full[k]) for k in _TOP_KEYS if k in full}
    cfg = full.get("config") or {}
    line["config"] = {"workload": _short(cfg.get("workload", ""), 160)}
    for k in ("frames_per_step_per_gpu", "frames_per_launch", "launches_per_step", "parallelism", "kernel", "beam", "utts_per_gpu",
              "utts_total", "order_mode"):
        if k in cfg:
            line["config"][k] = _short(cfg[k], 80) if isinstance(cfg[k], str) else cfg[k]
    roof = full.get("roofline") or {}
    line["roofline"] = {k: (_num(roof.get(k)) if k not in ("bound", "unit") else _short(roof.get(k, ""), 40)) for k in _ROOF_KEYS}
    if roof.get("traffic") is not None:     # where the figure comes from: "measured" in this run, or replayed from profiles/ 
        line["roofline"]["traffic_source"] = _short(roof.get("traffic_source") or "measured in this run", 90)
    for k in ("beam_kernel_ms", "score_kernels_ms"):
        if roof.get(k) is not None:
            line["roofline"][k] = _num(roof[k])
    hbm = roof.get("hbm")
    if isinstance(hbm, dict):
        line["roofline"]["hbm"] = {k: _num(hbm.get(k)) for k in ("algorithmic_GBs", "algorithmic_frac_of_peak", "compulsory_GBs",
                                                                  "measured_bytes_per_launch", "peak_GBs") if k in hbm}
    sm = full.get("k1_small_T")
    if isinstance(sm, dict):      # K1 called with a handful of frames: [frames per call, us per call, fraction of the HBM peak the model streams at]
        line["k1_small_T"] = [[c.get("frames_per_call"), _num(c.get("us_per_call"), 4), _num(c.get("frac_of_hbm_peak"), 3)] for c in sm.get("calls", [])]
    c = _cpu(full.get("cpu_baseline"))
    if c:
        line["cpu_baseline"] = c
    if "parity_spot_check" in full:
        line["parity_spot_check"] = full["parity_spot_check"]
    p = _parity(full.get("parity"))
    if p:
        line["parity"] = p
    if isinstance(full.get("pass1"), dict):
        line["pass1"] = {"ok": full["pass1"].get("ok"), "utts": full["pass1"].get("utts")}
    for k, v in full.items():
        if isinstance(v, dict) and "ms_per_step" in v and k not in line:
            line[k] = _nested(v, lean)
    if "detail_file" in full:
        line["detail_file"] = full["detail_file"]
    return line


def final_line(full: dict) -> str:
    """json of compact_line(full); if that exceeds MAX_LINE_BYTES the nested records go lean (ms_per_step, rtf_inv, steps,
    kernel times, roofline.frac, cpu_baseline.rtf_inv, parity counts) -- never the contract keys of the top level."""
    s = json.dumps(compact_line(full), separators=(",", ":"))
    if len(s) > MAX_LINE_BYTES:
        s = json.dumps(compact_line(full, lean=True), separators=(",", ":"))
    return s
