#!/usr/bin/env python3
"""profiles/traffic_gmm_tile.json and profiles/traffic_first_pass.json from the PMC summaries of tools/r05_profiles.sh
(gpurun_out/<round>/*_traffic_pmc_summary.json: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs).

Units and corrections (MI355X_MICROARCH.md "HBM"): both counters are in KiB-sized units of 1024 B as rocprofv3 reports them
(FETCH_SIZE = TCC_EA0_RDREQ x 64 B / 1024), memory-side of the L2, Infinity-Cache hits included.  On gfx950 FETCH_SIZE
reports half of the bytes of WIDE COALESCED reads (128-B requests tallied at 64 B): K1's reads are of that kind (frames and
scalar-loaded records) and are doubled, as in rounds 2-4.  The first-pass kernels gather 4 - 32 bytes per lane: the guide
calls other access widths uncalibrated, so their FETCH_SIZE is taken AS REPORTED (the figure could be up to 2x higher) and
both forms are recorded; WRITE_SIZE is used as reported everywhere."""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
rnd = sys.argv[1] if len(sys.argv) > 1 else "r05"
src = ROOT / "gpurun_out" / rnd


def counters(name, sub):
    d = json.loads((src / f"{name}_traffic_pmc_summary.json").read_text())
    out = {}
    for run, v in d.items():
        for k, c in v.get("pmc_avg_per_dispatch", {}).items():
            if sub in k:
                out.update(c)
                out["kernel"] = k
        for x in v.get("kernels", []):
            if sub in x["name"]:
                out.setdefault("avg_ms", []).append(round(x["avg_us"] / 1e3, 3))
                out["rocprof_kernel_name"] = x["name"].replace("void (anonymous namespace)::", "").split("(")[0]
    return out


g = counters("gmm", "gmm_tile")
fetch, write = g["FETCH_SIZE"] * 1024.0, g["WRITE_SIZE"] * 1024.0
gm = {"kernel": "gmm_tile<D=39,FPL=2,NS=16> grid=24000 nsb=16", "rocprof_kernel_name": g["rocprof_kernel_name"],
      "frames_per_launch": 64000,
      "source": f"profiles/{rnd}_gmm_traffic_pmc_summary.json (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes, bench.py --workload gmm "
                "--steps 3 --warmup 1 --no-cpu-baseline: 64 utterances x 1000 frames per launch), the kernel of round 5 (log-sum step of 22 "
                "instructions; the tiling, and so the traffic, is round 4's); 'kernel' is the string gmm.last_kernel() reports, which bench.py requires to match",
      "FETCH_SIZE_KB_avg_per_dispatch": g["FETCH_SIZE"], "WRITE_SIZE_KB_avg_per_dispatch": g["WRITE_SIZE"],
      "correction": "MI355X_MICROARCH.md 'HBM': on gfx950 FETCH_SIZE tallies 128-B requests at 64 B -> doubled (wide coalesced reads); WRITE_SIZE as reported (the [T][S] output is 768.0 MB)",
      "hbm_bytes_per_launch": int(2 * fetch + write), "kernel_ms_under_pmc": g.get("avg_ms")}
(ROOT / "profiles" / "traffic_gmm_tile.json").write_text(json.dumps(gm, indent=1))
print("gmm", gm["hbm_bytes_per_launch"])

entries = []
for name, sub, dnn, mp, utts, beam, shape, frames in (
        ("e2e_512", "beam_exact_kernel", False, False, 512, 800, "half", 727200),
        ("e2e_256", "beam_exact_kernel", False, False, 256, 800, "full", 363600),
        ("e2e_dnn_256", "beam_exact_kernel", True, False, 256, 4000, "full", 358848),
        ("e2e_mp_256", "beam_exact_mp_kernel", False, True, 256, 800, "full", 363600),
        ("e2e_mp_512", "beam_exact_mp_kernel", False, True, 512, 800, "half", 727200),
        ("e2e_dnn_mp_256", "beam_exact_mp_kernel", True, True, 256, 4000, "full", 358848)):
    c = counters(name, sub)
    fetch, write = c["FETCH_SIZE"] * 1024.0, c["WRITE_SIZE"] * 1024.0
    ms = sum(c["avg_ms"]) / len(c["avg_ms"])
    entries.append({"dnn": dnn, "multipath": mp, "flat": False, "utts": utts, "beam": beam, "shape": shape,
                    "rocprof_kernel_name": c["rocprof_kernel_name"], "frames_per_launch": frames,
                    "FETCH_SIZE_KB_avg_per_dispatch": c["FETCH_SIZE"], "WRITE_SIZE_KB_avg_per_dispatch": c["WRITE_SIZE"],
                    "hbm_bytes_per_launch": int(fetch + write), "hbm_bytes_per_launch_if_fetch_doubled": int(2 * fetch + write),
                    "kernel_ms_under_pmc": round(ms, 2), "GBps": round((fetch + write) / ms / 1e6, 1),
                    "bytes_per_utterance_frame": round((fetch + write) / frames),
                    "source": f"profiles/{rnd}_{name}_traffic_pmc_summary.json"})
    print(name, entries[-1]["hbm_bytes_per_launch"], entries[-1]["GBps"], entries[-1]["bytes_per_utterance_frame"])
tf = {"what": "memory-side traffic of ONE first-pass launch (L2 <-> fabric: HBM and Infinity Cache together), rocprofv3 --pmc FETCH_SIZE / "
              "--pmc WRITE_SIZE in separate runs of bench.py --workload e2e|e2e-dnn [--multipath] --utts N --steps 1 --warmup 1 --no-pipeline; "
              "FETCH_SIZE as reported (gather access widths are uncalibrated on gfx950: up to 2x higher, see hbm_bytes_per_launch_if_fetch_doubled), "
              "WRITE_SIZE as reported.  bench.py reports roofline.traffic only for a launch that matches an entry (scorer, -multipath, utterances, beam, shape).",
      "entries": entries}
(ROOT / "profiles" / "traffic_first_pass.json").write_text(json.dumps(tf, indent=1))
