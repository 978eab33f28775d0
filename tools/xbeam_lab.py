#!/usr/bin/env python3
"""Development: first-pass kernel timings of library variants on ONE task set-up.

  xbeam_lab.py prepare DIR                     build the C3 task (reference formats -> jamd_export blobs) + utterances
  xbeam_lab.py run DIR [--lib SO] [--tag T] [--what c3,c3b,wide,wideb] [--order exact]

`run` (one process per library variant, JAMD_LIB semantics) prints one JSON line:
  c3    beam 800, 1 utterance, instrumented launch: phase clocks (us per frame)
  c3b   beam 800, 256 utterances: kernel ms, frames/s
  wide  beam 4000, flat random scores (what a random-init DNN emits), 1 utterance, phase clocks
  wideb beam 4000, 64 utterances
bench.py stays the reported number; this only avoids paying the task set-up once per variant."""
import argparse
import json
import os
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def prepare(d: Path):
    from julius_amd import synth
    import bench
    d.mkdir(parents=True, exist_ok=True)
    task, jargs, prefix = bench.build_reference_task(d, 20000, 800)
    assert prefix is not None, "julius_amd/jamd_export missing"
    for u in range(32):
        np.save(d / f"utt{u}.npy", synth.make_utterance(task, nwords=30, seed=u)[0])
    rng = np.random.default_rng(7)
    for u in range(8):
        np.save(d / f"flat{u}.npy", rng.normal(-8.0, 0.33, (300, 3000)).astype(np.float32))
    print("prepared", d)


def run(d: Path, args):
    import torch
    from julius_amd import lib
    if args.lib:
        lib.LIB_PATH = Path(args.lib).resolve()
    eng = lib.Engine(0)
    gm = lib.Gmm.from_file(eng, str(d / "task.am"))
    lx = lib.Lexicon.from_file(eng, str(d / "task.lex"))
    uniq = [np.load(d / f"utt{u}.npy") for u in range(32)]
    flat = [np.load(d / f"flat{u}.npy") for u in range(8)]
    st = torch.cuda.Stream()
    out = {"tag": args.tag or (Path(args.lib).stem if args.lib else "product")}

    def bench_case(beam, nutt, scores_of, timed, reps=3):
        os.environ["JAMD_BEAM_TIMING"] = "1" if timed else "0"
        bm = lib.Beam(eng, lx, beam, -1.0, max_utts=nutt, atoms_per_utt=1 << (18 if beam > 1600 else 17))
        if args.order:
            bm.set_order_mode(args.order)
        if args.shape:
            bm.set_workgroup_shape(args.shape)
        d_sc, off, maxlen = scores_of(nutt)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        best = 1e30
        for it in range(reps + 1):
            ev[0].record(st)
            bm.pass1_dev(d_sc.data_ptr(), 3000, off, st.cuda_stream)
            ev[1].record(st)
            torch.cuda.synchronize()
            if it:
                best = min(best, ev[0].elapsed_time(ev[1]))
        res = bm.results()
        r = {"beam_ms": round(best, 3), "us_per_frame": round(best * 1e3 / maxlen, 2), "frames_per_s": round(int(off[-1]) / (best * 1e-3)),
             "ok": int(sum(x.status == 0 for x in res)), "peak_tokens": int(res[0].max_tokens), "shape": bm.workgroup_shape(nutt)}
        if timed:
            r["phase_us_per_frame"] = [round(x / maxlen, 2) for x in res[0].phase_us]
        bm.close()
        del d_sc
        return r

    def gmm_scores(nutt):
        utts = [uniq[u % 32] for u in range(nutt)]
        off = np.zeros(nutt + 1, np.int32)
        off[1:] = np.cumsum([len(x) for x in utts])
        fr = torch.from_numpy(np.concatenate(utts)).cuda()
        sc = torch.empty((int(off[-1]), 3000), dtype=torch.float32, device="cuda")
        gm.outprob_dev(fr.data_ptr(), int(off[-1]), sc.data_ptr(), st.cuda_stream)
        torch.cuda.synchronize()
        return sc, off, max(len(x) for x in utts)

    def flat_scores(nutt):
        utts = [flat[u % 8] for u in range(nutt)]
        off = np.zeros(nutt + 1, np.int32)
        off[1:] = np.cumsum([len(x) for x in utts])
        return torch.from_numpy(np.concatenate(utts)).cuda(), off, 300

    what = args.what.split(",")
    if "c3" in what:
        out["c3"] = bench_case(800, 1, gmm_scores, True)
    if "c3b" in what:
        out["c3b"] = bench_case(800, 256, gmm_scores, False)
    if "c3c" in what:
        out["c3c"] = bench_case(800, 512, gmm_scores, False, reps=2)
    if "c3d" in what:
        out["c3d"] = bench_case(800, 1024, gmm_scores, False, reps=2)
    if "wide" in what:
        out["wide"] = bench_case(4000, 1, flat_scores, True, reps=2)
    if "wideb" in what:
        out["wideb"] = bench_case(4000, 64, flat_scores, False, reps=2)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("cmd", choices=["prepare", "run"])
    ap.add_argument("dir")
    ap.add_argument("--lib", default=None)
    ap.add_argument("--tag", default=None)
    ap.add_argument("--what", default="c3,c3b,wide,wideb")
    ap.add_argument("--order", default=None)
    ap.add_argument("--shape", default=None, help="auto | full | half (jamd_beam_set_workgroup_shape)")
    a = ap.parse_args()
    if a.cmd == "prepare":
        prepare(Path(a.dir))
    else:
        run(Path(a.dir), a)
