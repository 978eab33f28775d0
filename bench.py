#!/usr/bin/env python3
"""bench.py -- throughput of the hot path on MI355X (driver contract, see README).

Workload (BASELINE.json configs[1], SURVEY.md section 8d "C2"): tied-state
triphone-sized GMM, S=3000 states x M=16 mixtures x D=39, outprob kernel only.
One "step" = one pass of the GMM outprob path over one batch of
`--utts` (default 64) synthetic utterances x 1000 frames (seeded synthetic MFCC) already
resident in HBM: [T][39] floats in, [T][3000] log10 likelihoods out.

  value   = frame*states scored per second, whole job (all ranks)
  rtf_inv = audio-seconds per wall-second (100 frames = 1 s)
  roofline = algorithmic bytes of the frame-synchronous formulation
             (S*M*(2D+2)*4 + D*4 + S*4 per frame) / HIP-event
             kernel time, against the 8 TB/s HBM peak; `valu` is the fp32 VALU
             issue roofline that actually binds the frame-tiled kernel
             (DESIGN.md "K1 roofline").
  cpu_baseline = the compiled reference (oracle/_ref, kind "reference") scoring
             a bounded sample of the same workload on ONE host core, or the
             oracle port when _ref is absent.

Multi-GPU: one process per GPU (torch.distributed / RCCL used only for the
barrier and the max-over-ranks clock); utterances are sharded, no data-path
collective ("weak" scaling: per-GPU batch fixed).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

S, M, D = 3000, 16, 39
FRAMES_PER_UTT = 1000
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
VALU_PEAK_TOPS = 78.6          # 256 CU x 4 SIMD x 32 lanes x 2.4 GHz, one fp32 op/lane/clk (no FMA allowed)


def cpu_baseline(model, frames, seconds_budget=12.0):
    """Reference CPU path on a bounded sample (rank 0, N=1 only)."""
    from oracle import pyoracle
    try:
        ref = pyoracle.Ref()
        am = ref.am_from_flat(model)
        am.outprob(frames[:20], want_out=False)          # warm + calibrate
        per = max(am.last_seconds / 20, 1e-5)
        n = int(min(len(frames), max(50, seconds_budget / per)))
        am.outprob(frames[:n], want_out=False)
        sec = am.last_seconds
        kind = "reference"
        what = "compiled reference libsent (outprob_state batch loop -> calc_mix -> gprune_none -> addlog_array)"
    except (FileNotFoundError, OSError):
        orc = pyoracle.Oracle()
        t = time.perf_counter(); orc.gmm_outprob(model, frames[:10]); per = (time.perf_counter() - t) / 10
        n = int(min(len(frames), max(20, seconds_budget / per)))
        t = time.perf_counter(); orc.gmm_outprob(model, frames[:n]); sec = time.perf_counter() - t
        kind = "port"
        what = "oracle/jamd_oracle_am.c restatement"
    return {
        "value": n * S / sec, "unit": "frame*states/s", "cores": 1, "kind": kind,
        "sample": f"{n} frames x {S} states eager scoring, {what}, {sec:.2f} s on 1 of {os.cpu_count()} host cores",
        "rtf_inv": n / 100.0 / sec,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--utts", type=int, default=64,
                    help="utterances (x1000 frames) per GPU per step (64 = one GPU's share of the 512-utterance batch of configs[4])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", default="gmm", choices=["gmm", "dnn", "e2e", "e2e-dnn"],
                    help="gmm = BASELINE configs[1] (the contract default); dnn = configs[3] scoring half; "
                         "e2e = configs[2]: GMM outprob + HIP first pass on a 20k-word lexicon; "
                         "e2e-dnn = configs[3]: MFMA DNN outprob + HIP first pass")
    ap.add_argument("--beam", type=int, default=800, help="e2e: rank beam (-b; reference default for triphone models)")
    ap.add_argument("--nword", type=int, default=20000, help="e2e: vocabulary size")
    ap.add_argument("--order", default=None, choices=["fast", "strict", "exact", "exact_serial"],
                    help="e2e: first-pass tie order mode (default: the work area's default = exact)")
    args = ap.parse_args()
    if args.workload == "dnn":
        return main_dnn(args)
    if args.workload in ("e2e", "e2e-dnn"):
        return main_e2e(args)

    import torch
    from julius_amd import lib, synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or "TORCHELASTIC_RUN_ID" in os.environ:   # under torch.distributed.run: always rendezvous
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))

    # same model on every rank (replicated); each rank its own utterances
    model = synth.make_gmm(S=S, M=M, D=D, seed=0)
    T = args.utts * FRAMES_PER_UTT
    frames = np.concatenate([synth.make_frames(model, T=FRAMES_PER_UTT, seed=1000 + rank * args.utts + u)
                             for u in range(args.utts)])
    eng = lib.Engine(local_rank)
    gmm = lib.Gmm(eng, model)
    d_frames = torch.from_numpy(frames).cuda()
    d_out = torch.empty((T, S), dtype=torch.float32, device="cuda")
    # a non-default stream: its handle is what the C ABI launches on, and the
    # HIP events below are recorded on that same stream
    stream = torch.cuda.Stream()
    torch.cuda.synchronize()

    def step():
        gmm.outprob_dev(d_frames.data_ptr(), T, d_out.data_ptr(), stream.cuda_stream)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for a, b in ev:
        a.record(stream)
        step()
        b.record(stream)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # part of the cpu_baseline leg (rank 0, N=1): a few (t, s) entries of the last output are
    # also checked against the oracle's values for the same frames
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import pyoracle
        rng = np.random.default_rng(0)
        ss = np.sort(rng.choice(S, 8, replace=False))
        tt = np.sort(rng.choice(T, 8, replace=False))
        sub = dict(model)
        sub["st_off"] = (np.arange(len(ss) + 1) * M).astype(np.int32)
        idx = np.concatenate([np.arange(model["st_off"][s], model["st_off"][s + 1]) for s in ss])
        sub["ent_dens"], sub["ent_logw"] = model["ent_dens"][idx], model["ent_logw"][idx]
        want = pyoracle.Oracle().gmm_outprob(sub, frames[tt])
        got = d_out[torch.from_numpy(tt).cuda()][:, torch.from_numpy(ss).cuda()].cpu().numpy()
        parity = bool(np.array_equal(got, want))
    else:
        parity = None

    if rank == 0:
        total_frames = T * world * args.steps
        value = total_frames * S / elapsed
        E = int(model["st_off"][-1])
        bytes_per_frame = E * (2 * D + 2) * 4 + D * 4 + S * 4
        alg_bytes = bytes_per_frame * T                  # per launch (one launch per step per rank)
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9    # GB/s
        valu_ops = T * E * D * 4.0                       # sub, mul, mul, add per (frame, Gaussian, dim)
        traffic = None
        tfile = ROOT / "profiles" / "traffic_gmm_tile.json"
        if tfile.exists():
            try:
                tj = json.loads(tfile.read_text())
                if tj.get("frames_per_launch") == T:
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "frames_x_states_scored_per_sec", "value": value, "unit": "frame*states/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "rtf_inv": total_frames / 100.0 / elapsed,
            "config": {"workload": "C2 (BASELINE.json configs[1]): tied-state triphone GMM outprob only, "
                                   f"S={S} x M={M} x D={D}, {args.utts} utterances x {FRAMES_PER_UTT} frames per GPU per step, "
                                   "gprune none", "frames_per_step_per_gpu": T, "parallelism": f"utterance-sharded x{world}",
                       "kernel": gmm.last_kernel()},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel_ms": kern_ms, "algorithmic_bytes_per_launch": alg_bytes,
                         "note": "frame-tiled kernel: the model is streamed once per 512-frame block, so algorithmic "
                                 "(per-frame) bytes exceed real HBM traffic and frac may exceed 1; see valu",
                         "valu": {"achieved": valu_ops / (kern_ms * 1e-3) / 1e12, "peak": VALU_PEAK_TOPS,
                                  "unit": "Tops/s (fp32, unfused)", "frac": valu_ops / (kern_ms * 1e-3) / 1e12 / VALU_PEAK_TOPS}},
            "parity_spot_check": parity,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(model, frames)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main_dnn(args):
    """configs[3] scoring half: 528 -> 6 x 2048 table-sigmoid -> 4000 senones, batched frames.
    Reported against the fp32 MFMA roofline (157.3 TFLOP/s, MI355X_MICROARCH.md)."""
    import torch
    from julius_amd import lib, synth
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or "TORCHELASTIC_RUN_ID" in os.environ:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    dnn = synth.make_dnn(seed=0)
    T = args.utts * FRAMES_PER_UTT
    frames = np.random.default_rng(100 + rank).normal(0, 1, (T, 528)).astype(np.float32)
    eng = lib.Engine(local_rank)
    net = lib.Dnn(eng, dnn)
    d_fr = torch.from_numpy(frames).cuda()
    d_out = torch.empty((T, net.S), dtype=torch.float32, device="cuda")
    stream = torch.cuda.Stream()
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        net.outprob_dev(d_fr.data_ptr(), T, d_out.data_ptr(), stream.cuda_stream)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for a, b in ev:
        a.record(stream)
        net.outprob_dev(d_fr.data_ptr(), T, d_out.data_ptr(), stream.cuda_stream)
        b.record(stream)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    if rank != 0:
        dist.barrier()
        dist.destroy_process_group()
        return
    dims = [int(x) for x in dnn["dims"]]
    flops = 2.0 * T * sum(dims[i] * dims[i + 1] for i in range(len(dims) - 1))
    line = {"metric": "frames_x_states_scored_per_sec", "value": T * world * args.steps * net.S / elapsed,
            "unit": "frame*states/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic", "rtf_inv": T * world * args.steps / 100.0 / elapsed,
            "config": {"workload": f"C4 scoring half (BASELINE.json configs[3]): DNN {dims}, {T} frames per GPU per step",
                       "parallelism": f"utterance-sharded x{world}"},
            "roofline": {"bound": "mfma", "achieved": flops / (ms * 1e-3) / 1e12, "peak": 157.3, "unit": "TFLOP/s",
                         "frac": flops / (ms * 1e-3) / 1e12 / 157.3, "traffic": None, "kernel_ms": ms}}
    if world == 1 and not args.no_cpu_baseline:
        # cpu_baseline leg: the oracle's restatement of the reference FMA kernel on a bounded sample
        # (~10 s on one host core); the same rows double as a parity spot check of the device output
        from oracle import pyoracle
        orc = pyoracle.Oracle()
        t0 = time.perf_counter(); orc.dnn_outprob(dnn, frames[:8], pyoracle.DNN_FMA); per = (time.perf_counter() - t0) / 8
        n = int(min(T, max(16, 10.0 / max(per, 1e-4))))
        t0 = time.perf_counter(); want = orc.dnn_outprob(dnn, frames[:n], pyoracle.DNN_FMA); sec = time.perf_counter() - t0
        line["parity_spot_check"] = bool(np.array_equal(d_out[:n].cpu().numpy(), want))
        line["cpu_baseline"] = {"value": n * net.S / sec, "unit": "frame*states/s", "cores": 1, "kind": "port",
                                "rtf_inv": n / 100.0 / sec,
                                "sample": f"{n} frames: oracle restatement of calc_dnn_fma() + table sigmoid/log-softmax, "
                                          f"{sec:.2f} s on 1 of {os.cpu_count()} host cores"}
    print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main_e2e(args):
    """configs[2]: tied-state triphone GMM (S=3000 x M=16 x D=39) + 20k-word tree
    lexicon with 2-gram, outprob kernel + HIP first pass, end to end on the device:
    frames in HBM -> [T][S] scores in HBM -> word trellis + pass-1 sentence.
    One step = `--utts` utterances (one workgroup each) per GPU.  The lexicon is
    synthetic (julius_amd.synth.make_lexicon, same structural rules as the
    reference's builder); utterances follow random word sequences through it."""
    import torch
    from julius_amd import lib, synth
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or "TORCHELASTIC_RUN_ID" in os.environ:   # under torch.distributed.run: always rendezvous
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    use_dnn = args.workload == "e2e-dnn"
    eng = lib.Engine(local_rank)
    if use_dnn:
        # configs[3]: 528 -> 6 x 2048 -> 4000 senones; the lexicon's states index the DNN outputs.
        # Random-init weights: the scores carry no sentence, the search runs with a saturated beam.
        dnn = synth.make_dnn(seed=0)
        NS = int(dnn["dims"][-1])
        lex = synth.make_lexicon(nword=args.nword, nphone=40, S=NS, seed=0)
        scorer = lib.Dnn(eng, dnn)
        rng = np.random.default_rng(1000 + rank)
        uniq = [(rng.normal(0, 1, (FRAMES_PER_UTT, int(dnn["dims"][0]))).astype(np.float32), None)
                for _ in range(min(args.utts, 16))]
        what = f"DNN {[int(x) for x in dnn['dims']]} (MFMA fp32) outprob"
    else:
        NS = S
        lex = synth.make_lexicon(nword=args.nword, nphone=40, S=S, seed=0)
        model = synth.make_gmm(S=S, M=M, D=D, seed=0)
        scorer = lib.Gmm(eng, model)
        uniq = [synth.make_lexicon_utterance(lex, model, nwords=30, seed=1000 * rank + u) for u in range(min(args.utts, 16))]
        what = f"GMM S={S} x M={M} x D={D} outprob"
    nuniq = len(uniq)
    utts = [uniq[u % nuniq][0] for u in range(args.utts)]
    off = np.zeros(args.utts + 1, np.int32)
    off[1:] = np.cumsum([len(x) for x in utts])
    frames = np.concatenate(utts)
    T = len(frames)
    lx = lib.Lexicon(eng, lex)
    bm = lib.Beam(eng, lx, args.beam, -1.0, max_utts=args.utts, atoms_per_utt=1 << 17)
    if args.order:
        bm.set_order_mode(args.order)
    d_fr = torch.from_numpy(frames).cuda()
    d_sc = torch.empty((T, NS), dtype=torch.float32, device="cuda")
    stream = torch.cuda.Stream()

    def step():
        scorer.outprob_dev(d_fr.data_ptr(), T, d_sc.data_ptr(), stream.cuda_stream)
        bm.pass1_dev(d_sc.data_ptr(), NS, off, stream.cuda_stream)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3 * args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        ev[3 * i].record(stream)
        scorer.outprob_dev(d_fr.data_ptr(), T, d_sc.data_ptr(), stream.cuda_stream)
        ev[3 * i + 1].record(stream)
        bm.pass1_dev(d_sc.data_ptr(), NS, off, stream.cuda_stream)
        ev[3 * i + 2].record(stream)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    gmm_ms = float(np.mean([ev[3 * i].elapsed_time(ev[3 * i + 1]) for i in range(args.steps)]))
    beam_ms = float(np.mean([ev[3 * i + 1].elapsed_time(ev[3 * i + 2]) for i in range(args.steps)]))
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    if rank == 0:
        res = bm.results()
        ok = sum(1 for r in res if r.status == 0)
        correct = None if use_dnn else sum(1 for u, r in enumerate(res) if list(r.wseq[:r.wnum]) == uniq[u % nuniq][1])
        total_frames = T * world * args.steps
        tokens = float(np.mean([r.max_tokens for r in res]))
        cfg = "C4 (BASELINE.json configs[3])" if use_dnn else "C3 (BASELINE.json configs[2])"
        line = {"metric": "frames_x_states_scored_per_sec", "value": total_frames * NS / elapsed,
                "unit": "frame*states/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "f32", "data": "synthetic", "rtf_inv": total_frames / 100.0 / elapsed,
                "config": {"workload": f"{cfg}: {what} + HIP first pass, "
                                       f"{args.nword}-word tree lexicon ({lex['nnode']} nodes, {lex['startnum']} roots) + 2-gram, "
                                       f"beam {args.beam}, {args.utts} utterances ({T} frames) per GPU per step",
                           "parallelism": f"utterance-sharded x{world}"},
                "roofline": {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None,
                             "traffic": None, "note": "irregular gather/scatter: no algorithmic-bytes roofline "
                             "(SURVEY.md 8d); figure of merit is end-to-end frames/s",
                             "score_kernels_ms": gmm_ms, "beam_kernel_ms": beam_ms,
                             "beam_frames_per_s": T / (beam_ms * 1e-3),
                             "beam_us_per_frame_per_utt": beam_ms * 1e3 / max(len(x) for x in utts)},
                "pass1": {"ok": ok, "sentence_correct": correct, "utts": len(res), "mean_peak_tokens": tokens,
                          "ties": int(sum(r.ties for r in res)), "phase_us_utt0": list(res[0].phase_us)}}
        if world == 1 and not args.no_cpu_baseline:
            from oracle import pyoracle
            orc = pyoracle.Oracle()
            fr0 = utts[0]
            tt0 = time.perf_counter()
            if use_dnn:
                fr0 = fr0[:100]      # ~6 ms per frame on one core
                sc0 = orc.dnn_outprob(dnn, fr0, pyoracle.DNN_FMA)
            else:
                sc0 = orc.gmm_outprob(model, fr0)
            tt1 = time.perf_counter()
            atoms, wseq, score, rc, died = orc.beam_pass1(lex, sc0, args.beam, -1.0)
            tt2 = time.perf_counter()
            got = d_sc[:len(fr0)].cpu().numpy()
            line["parity_spot_check"] = bool(np.array_equal(got, sc0)) and (use_dnn or (
                list(wseq) == list(res[0].wseq[:res[0].wnum]) and float(score) == float(res[0].score)))
            line["cpu_baseline"] = {"value": len(fr0) * NS / (tt2 - tt0), "unit": "frame*states/s", "cores": 1,
                                    "kind": "port", "rtf_inv": len(fr0) / 100.0 / (tt2 - tt0),
                                    "sample": f"1 utterance of {len(fr0)} frames: oracle eager {'DNN' if use_dnn else 'GMM'} scoring {tt1 - tt0:.2f} s + "
                                              f"oracle first pass {tt2 - tt1:.2f} s on 1 of {os.cpu_count()} host cores"}
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
