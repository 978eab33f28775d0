#!/usr/bin/env python3
"""bench.py -- throughput of the hot path on MI355X (driver contract, see README).

stdout: every nested result in full as its own line ({"bench_detail": key, ...}), then -- LAST -- one compact JSON line
(julius_amd/benchfmt.py: the contract keys of the top-level workload + a short record per nested configuration, < 6 KB;
the whole tree also goes to bench_detail.json).  Top level = BASELINE.json configs[1] ("C2"): tied-state
triphone-sized GMM, S=3000 states x M=16 mixtures x D=39, outprob kernel only.  One "step" = one
pass of the GMM outprob path over `--utts` (default 64) synthetic utterances x 1000 frames already
resident in HBM: [T][39] floats in, [T][3000] log10 likelihoods out.

  value    = frame*states scored per second, whole job (all ranks)
  roofline = the roof that BINDS the frame-tiled kernel: fp32 VALU issue (4 separately rounded
             operations per (frame, Gaussian, dimension); no FMA allowed: the reference object code has
             none).  Sub-object `hbm` carries SURVEY 8d's algorithmic-bytes figure (the model streamed
             once per FRAME, which the kernel does not do), the compulsory bytes of a launch and the
             measured HBM traffic (PMC passes, profiles/traffic_gmm_tile.json).
  e2e      = nested result for configs[2] ("C3"): GMM outprob + first pass on the device over a
             20k-word lexicon BUILT BY THE REFERENCE (dict + ARPA -> Julius' own loaders and wchmm
             builder -> jamd_export blobs -> jamd_gmm_load / jamd_lexicon_load), >= 32 distinct
             utterances, exact-order kernel.  `parity` compares the device result with the compiled
             reference's first pass (julius -1pass) utterance by utterance, and the exact-order kernel
             with the canonical-tie ("fast") kernel.
  e2e_strong = the same task as configs[4] specifies it: the FIXED batch of 512 utterances sharded over
             the GPUs of the run (512 on one GPU at N=1, 64 per GPU at N=8), scaling "strong", with
             whole-job and per-GPU frames/s.
  e2e_dnn  = nested result for configs[3] ("C4") END TO END at the reference recipe's beam (-b 4000):
             hmmdefs (4000 states) + dnnconf/.npy + dict + ARPA loaded by Julius' own readers, lexicon
             built by wchmm.c, MFMA DNN scores -> exact-order first pass; `parity` against the compiled
             reference's julius -1pass (its own dnn_calc_outprob + beam.c) utterance by utterance.
  e2e_dnn_strong / e2e_dnn_flat = the C4 task as configs[4]'s fixed 512-utterance batch / on the flat-score stream
             (random-init weights over noise: the worst case of the rank pruning step).
  e2e_mp, e2e_dnn_mp = the C3 and C4 tasks DECODED WITH -multipath (the form the reference README's DNN recipe runs:
             -b 4000 -multipath): the lexicon is the reference's multipath lexicon, the first pass is the multipath
             frame of the exact-order kernel (csrc/beam_exact_mp.h); `parity` against julius -1pass [-dnnconf]
             -multipath utterance by utterance (`--workload e2e|e2e-dnn --multipath` runs them alone).
  batch, batch_mp, batch_dnn, batch_dnn_mp = the same four tasks through the PRODUCT's serving loop: `jamd_batch -time` (C over
             the C ABI) over HTK parameter files on tmpfs -- file read, pinned staging, H2D, scoring, first pass, D2H of the
             results, result lines -- on the process's own clock (model load excluded); `vs_e2e_same_task` = its RTF^-1 over
             the e2e entry's, `parity` = its result lines against the in-process results.
  dnn      = nested result for the configs[3] scoring half ("C4"): MFMA fp32 DNN.
  cpu_baseline (top level and nested) = the COMPILED REFERENCE (oracle/_ref, kind "reference") on a
             bounded sample of the same workload: one host core, an N-process figure (C2: eager scoring; C3 / C4: 16 julius
             -1pass processes side by side), the full two-pass recogniser on a few of the same files, DNN num_threads 2 / half the cores.

Multi-GPU: one process per GPU; utterances are sharded, no data-path collective ("weak" scaling:
per-GPU batch fixed; `e2e_strong` / `--workload e2e --strong` = the fixed 512-utterance batch).  RCCL is used for the barrier, the max-over-ranks clock and the gather of the
per-utterance result records (julius_amd/shard.py).  `--gpus N` without a torch.distributed.run
environment spawns the N ranks itself.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

# The ROCm runtime multiplexes HIP streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), round robin in creation
# order; two streams that land on the same queue run their kernels one after the other.  The end-to-end steps rely on the
# scoring stream and the first-pass stream being concurrent, and this process creates more than four streams by then
# (measured: nested in the default run the two kernels serialised, 317 ms per step against 304 ms stand-alone).  Read
# when the runtime initialises, so it is set before anything touches HIP.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

S, M, D = 3000, 16, 39
FRAMES_PER_UTT = 1000
LAUNCHES_PER_STEP = 12         # C2: one step = 12 sub-batches of --utts utterances (>= 100 ms of kernel time per step)
C5_TOTAL_UTTS = 512            # BASELINE.json configs[4]: the fixed batch that is sharded over the GPUs
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
VALU_PEAK_TOPS = 78.6          # 256 CU x 4 SIMD x 32 lanes x 2.4 GHz, one fp32 op/lane/clk (no FMA allowed)
MFMA_F32_PEAK = 157.3          # TFLOP/s, dense fp32 MFMA


# ------------------------------------------------------------------------------------------------ dist
class Dist:
    """torch.distributed (backend nccl = RCCL) when there is more than one rank.  `backend` "gloo" + `share_device` (every
    rank on cuda:0) is the form in which the N > 1 orchestration -- rank spawn, strong split, barrier, max-over-ranks clock,
    gather of the result records -- runs on a ONE-GPU box (RCCL refuses two ranks on one device): tests/test_bench_ranks_gpu.py."""

    def __init__(self, gpus, backend="nccl", share_device=False):
        import torch
        self.torch = torch
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = 0 if share_device else int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.backend = backend
        self.coll_device = "cuda" if backend == "nccl" else "cpu"      # where the (tiny) collectives' tensors live
        if gpus > 1 and self.world != gpus:
            raise SystemExit(f"--gpus {gpus}: WORLD_SIZE={self.world}")
        torch.cuda.set_device(self.local_rank)
        self.dist = None
        if self.world > 1 or "TORCHELASTIC_RUN_ID" in os.environ:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            if backend == "nccl":
                dist.init_process_group("nccl", rank=self.rank, world_size=self.world,
                                        device_id=torch.device("cuda", self.local_rank))
            else:
                dist.init_process_group(backend, rank=self.rank, world_size=self.world)
            self.dist = dist

    def fence(self):
        self.torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def max_over_ranks(self, x: float) -> float:
        if self.dist is None:
            return x
        t = self.torch.tensor([x], dtype=self.torch.float64, device=self.coll_device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, x: int) -> int:
        if self.dist is None:
            return int(x)
        t = self.torch.tensor([int(x)], dtype=self.torch.int64, device=self.coll_device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return int(t.item())

    def close(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()


def timed_steps(dd: Dist, stream, step, steps, warmup, nevents=2):
    """W untimed steps, then exactly K steps between barrier+synchronize fences; HIP events on the launch
    stream around every step.  Returns (wall seconds max over ranks, list of per-step event lists)."""
    torch = dd.torch
    for _ in range(warmup):
        step(None)
    dd.fence()
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(nevents)] for _ in range(steps)]
    t0 = time.perf_counter()
    for e in ev:
        step((lambda i, e=e: e[i].record(stream)) if stream is not None else e)   # stream None: the step records e[i] itself
    dd.fence()
    elapsed = dd.max_over_ranks(time.perf_counter() - t0)
    return elapsed, ev


# ------------------------------------------------------------------------------------------------ C2
def ref_gmm_worker(nframes, seed):
    """One host core of the compiled reference: eager scoring of `nframes` frames x 3000 states."""
    from julius_amd import synth
    from oracle import pyoracle
    model = synth.make_gmm(S=S, M=M, D=D, seed=0)
    frames = synth.make_frames(model, T=nframes, seed=seed)
    am = pyoracle.Ref().am_from_flat(model)
    am.outprob(frames[:8], want_out=False)
    am.outprob(frames, want_out=False)
    return {"frames": nframes, "seconds": am.last_seconds}


def ref_e2e_worker(spec_path):
    """One host core of the compiled reference over some utterances of the end-to-end task: julius -1pass (lazy scoring +
    get_back_trellis_proceed) on the files the spec names; the canonical trellises, sentences and times go to an .npz."""
    from oracle import pyoracle
    spec = json.loads(Path(spec_path).read_text())
    t0 = time.perf_counter()
    eng = pyoracle.RefEngine(pyoracle.Ref(), spec["jargs"])
    load_s = time.perf_counter() - t0
    out = {"load_s": np.float64(load_s)}
    for u, mfc in zip(spec["utts"], spec["files"]):
        t0 = time.perf_counter()
        rtr, (rw, rs) = eng.recognize(mfc)
        out[f"sec_{u}"] = np.float64(time.perf_counter() - t0)
        out[f"w_{u}"] = np.asarray(rw, np.int32)
        out[f"s_{u}"] = np.float64(rs)
        for k, v in rtr.items():
            out[f"tr_{u}_{k}"] = v
    np.savez(spec["out"], **out)
    return {"ok": True}


def cpu_baseline_gmm(model, frames, spot, budget=10.0):
    """Compiled reference on a bounded sample: one core, then N processes side by side.  `spot` = (tt, ss,
    device values): 64 full-size rows checked against the reference's own scores for those frames."""
    from oracle import pyoracle
    ref = pyoracle.Ref()
    am = ref.am_from_flat(model)
    am.outprob(frames[:20], want_out=False)
    per = max(am.last_seconds / 20, 1e-5)
    n = int(min(len(frames), max(50, budget / per)))
    am.outprob(frames[:n], want_out=False)
    sec = am.last_seconds
    tt, ss, got = spot
    want = am.outprob(frames[tt], want_out=True)[:, ss]
    parity = bool(np.array_equal(got, want))
    out = {"value": n * S / sec, "unit": "frame*states/s", "cores": 1, "kind": "reference", "rtf_inv": n / 100.0 / sec,
           "sample": f"{n} frames x {S} states eager scoring, compiled reference libsent (outprob_state batch loop -> "
                     f"calc_mix -> gprune_none -> addlog_array), {sec:.2f} s on 1 of {os.cpu_count()} host cores"}
    ncore = max(1, min(32, (os.cpu_count() or 2) // 2))
    procs = [subprocess.Popen([sys.executable, str(ROOT / "bench.py"), "--cpu-worker", str(n), str(100 + i)],
                              stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for i in range(ncore)]
    t0 = time.perf_counter()
    outs = [p.communicate()[0] for p in procs]
    wall = time.perf_counter() - t0
    try:
        recs = [json.loads(o.strip().splitlines()[-1]) for o in outs]
        slow = max(r["seconds"] for r in recs)
        out["multi"] = {"value": sum(r["frames"] for r in recs) * S / slow, "unit": "frame*states/s", "cores": ncore,
                        "rtf_inv": sum(r["frames"] for r in recs) / 100.0 / slow,
                        "sample": f"{ncore} processes x {n} frames each, slowest {slow:.2f} s (wall incl. start-up {wall:.1f} s)"}
    except Exception as e:          # a worker died: report what happened, keep the one-core figure
        out["multi"] = {"error": repr(e)}
    return out, parity


def run_gmm(args, dd: Dist, steps, warmup):
    """One step = LAUNCHES_PER_STEP sub-batches (each --utts utterances x 1000 frames, its own frames and its own
    [T][S] output rows, all resident in HBM) scored back to back: 12 x 64 000 frames = 768 000 frames x 3000 states
    per GPU per step.  HIP events bracket every launch, so kernel_ms is per LAUNCH (what the roofline is quoted on)."""
    import torch
    from julius_amd import lib, synth
    model = synth.make_gmm(S=S, M=M, D=D, seed=0)
    L = LAUNCHES_PER_STEP
    T = args.utts * FRAMES_PER_UTT                       # frames per launch
    nutt = args.utts * L
    rng_seed = 1000 + dd.rank * nutt
    frames = np.concatenate([synth.make_frames(model, T=FRAMES_PER_UTT, seed=rng_seed + u) for u in range(nutt)])
    eng = lib.Engine(dd.local_rank)
    gmm = lib.Gmm(eng, model)
    d_frames = torch.from_numpy(frames).cuda()
    d_out = torch.empty((L * T, S), dtype=torch.float32, device="cuda")       # 9.2 GB at the default size
    stream = torch.cuda.Stream()      # the C ABI launches on this handle; the HIP events are recorded on it too
    torch.cuda.synchronize()
    fr_ptr, out_ptr = d_frames.data_ptr(), d_out.data_ptr()

    def step(mark):
        for l in range(L):
            if mark:
                mark(l)
            gmm.outprob_dev(fr_ptr + l * T * D * 4, T, out_ptr + l * T * S * 4, stream.cuda_stream)
        if mark:
            mark(L)

    elapsed, ev = timed_steps(dd, stream, step, steps, warmup, nevents=L + 1)
    kern_ms = float(np.mean([e[l].elapsed_time(e[l + 1]) for e in ev for l in range(L)]))
    res = None
    if dd.rank == 0:
        total_frames = L * T * dd.world * steps
        E = int(model["st_off"][-1])
        bytes_per_frame = E * (2 * D + 2) * 4 + D * 4 + S * 4
        alg_bytes = bytes_per_frame * T
        compulsory = E * (2 * D + 2) * 4 + T * D * 4 + T * S * 4      # model once + frames in + scores out
        valu_ops = T * E * D * 4.0
        traffic = None
        traffic_source = None
        tfile = ROOT / "profiles" / "traffic_gmm_tile.json"
        if tfile.exists():
            try:
                tj = json.loads(tfile.read_text())
                # only a measurement of THIS kernel instantiation at THIS launch size counts
                if tj.get("frames_per_launch") == T and tj.get("kernel") == gmm.last_kernel():
                    traffic = tj.get("hbm_bytes_per_launch")
                    traffic_source = "replayed from profiles/traffic_gmm_tile.json (a separate rocprofv3 --pmc pass of this kernel instantiation and launch size, not measured in this run)"
            except Exception:
                traffic = None
        tops = valu_ops / (kern_ms * 1e-3) / 1e12
        res = {
            "metric": "frames_x_states_scored_per_sec", "value": total_frames * S / elapsed, "unit": "frame*states/s",
            "n_gpus": dd.world, "steps": steps, "warmup": warmup,
            "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "rtf_inv": total_frames / 100.0 / elapsed,
            "config": {"workload": f"C2 = configs[1]: triphone GMM outprob only, S={S} M={M} D={D}, step = {L} launches x {T} frames",
                       "detail": "BASELINE.json configs[1]: tied-state triphone GMM outprob only, "
                                 f"S={S} x M={M} x D={D}, one step = {L} launches x {args.utts} utterances x {FRAMES_PER_UTT} "
                                 f"frames per GPU ({L * T} frames per GPU per step), gprune none",
                       "frames_per_step_per_gpu": L * T, "frames_per_launch": T, "launches_per_step": L,
                       "parallelism": f"utterance-sharded x{dd.world}", "kernel": gmm.last_kernel()},
            "roofline": {"bound": "valu", "achieved": tops, "peak": VALU_PEAK_TOPS, "unit": "Tops/s (fp32, unfused)",
                         "frac": tops / VALU_PEAK_TOPS, "traffic": traffic, "traffic_source": traffic_source, "kernel_ms": kern_ms,
                         "ops_per_launch": valu_ops,
                         "note": "4 separately rounded fp32 operations per (frame, Gaussian, dim) -- the reference's "
                                 "arithmetic, no FMA -- against 256 CU x 4 SIMD x 32 lanes x 2.4 GHz",
                         "hbm": {"algorithmic_bytes_per_launch": alg_bytes,
                                 "algorithmic_GBs": alg_bytes / (kern_ms * 1e-3) / 1e9,
                                 "algorithmic_frac_of_peak": alg_bytes / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                 "compulsory_bytes_per_launch": compulsory,
                                 "compulsory_GBs": compulsory / (kern_ms * 1e-3) / 1e9,
                                 "measured_bytes_per_launch": traffic, "peak_GBs": HBM_PEAK_GBS,
                                 "note": "SURVEY 8d counts the model once per FRAME (15.37 MB/frame); the kernel tiles "
                                         "512 frames per model sweep, so that figure exceeds the HBM peak by construction "
                                         "and is not a roofline; real traffic is the measured / compulsory bytes"}},
        }
        # The small-T regime: one call scores T frames and the 15.4 MB model is streamed for them alone --
        # the regime (live input, JAMD_STREAM_CHUNK = 25) in which north_star's "fraction of the HBM roofline" is the bound
        # that could bind.  One lane = one frame, a wave = 128 frame slots: a call of T < 128 frames fills T of them.
        small = []
        model_bytes = E * (2 * D + 2) * 4
        for Ts in (() if args.no_small_T else (1, 25, 100, 1000)):     # (--no-small-T: profiling passes that average a kernel's counters over its dispatches)
            ncall = 200
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(10):
                gmm.outprob_dev(fr_ptr, Ts, out_ptr, stream.cuda_stream)
            e0.record(stream)
            for _ in range(ncall):
                gmm.outprob_dev(fr_ptr, Ts, out_ptr, stream.cuda_stream)
            e1.record(stream)
            e1.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / ncall
            small.append({"frames_per_call": Ts, "us_per_call": us, "model_GBs": model_bytes / (us * 1e-6) / 1e9,
                          "frac_of_hbm_peak": model_bytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, "kernel": gmm.last_kernel()})
        res["k1_small_T"] = {"calls": small, "model_bytes": model_bytes,
                             "note": "back-to-back calls on one stream (launch overhead included); model_GBs = the model's bytes per call / "
                                     "time per call: what a frame-synchronous caller streams; the model stays in L2 / Infinity Cache between "
                                     "calls, so this is a rate, not measured HBM traffic"}
        if dd.world == 1 and not args.no_cpu_baseline:
            rng = np.random.default_rng(0)
            ss = np.sort(rng.choice(S, 16, replace=False))
            tt = np.sort(rng.choice(L * T, 64, replace=False))  # 64 full-size rows (any launch of the step) against the compiled reference
            got = d_out[torch.from_numpy(tt).cuda()][:, torch.from_numpy(ss).cuda()].cpu().numpy()
            res["cpu_baseline"], res["parity_spot_check"] = cpu_baseline_gmm(model, frames, (tt, ss, got))
    del d_out, d_frames
    return res


# ------------------------------------------------------------------------------------------------ C4 scoring
def run_dnn(args, dd: Dist, steps, warmup):
    """configs[3] scoring half: 528 -> 6 x 2048 table-sigmoid -> 4000 senones, batched frames, against the fp32
    MFMA roofline."""
    import torch
    from julius_amd import lib, synth
    dnn = synth.make_dnn(seed=0)
    T = args.utts * FRAMES_PER_UTT
    frames = np.random.default_rng(100 + dd.rank).normal(0, 1, (T, 528)).astype(np.float32)
    eng = lib.Engine(dd.local_rank)
    net = lib.Dnn(eng, dnn)
    d_fr = torch.from_numpy(frames).cuda()
    d_out = torch.empty((T, net.S), dtype=torch.float32, device="cuda")
    stream = torch.cuda.Stream()
    torch.cuda.synchronize()

    def step(mark):
        if mark:
            mark(0)
        net.outprob_dev(d_fr.data_ptr(), T, d_out.data_ptr(), stream.cuda_stream)
        if mark:
            mark(1)

    elapsed, ev = timed_steps(dd, stream, step, steps, warmup)
    ms = float(np.mean([e[0].elapsed_time(e[1]) for e in ev]))
    if dd.rank != 0:
        return None
    dims = [int(x) for x in dnn["dims"]]
    flops = 2.0 * T * sum(dims[i] * dims[i + 1] for i in range(len(dims) - 1))
    res = {"metric": "frames_x_states_scored_per_sec", "value": T * dd.world * steps * net.S / elapsed,
           "unit": "frame*states/s", "n_gpus": dd.world, "steps": steps, "warmup": warmup,
           "ms_per_step": elapsed / steps * 1e3, "dtype": "f32", "rtf_inv": T * dd.world * steps / 100.0 / elapsed,
           "config": {"workload": f"C4 scoring half (BASELINE.json configs[3]): DNN {dims}, {T} frames per GPU per step"},
           "roofline": {"bound": "mfma", "achieved": flops / (ms * 1e-3) / 1e12, "peak": MFMA_F32_PEAK, "unit": "TFLOP/s",
                        "frac": flops / (ms * 1e-3) / 1e12 / MFMA_F32_PEAK, "traffic": None, "kernel_ms": ms}}
    if dd.world == 1 and not args.no_cpu_baseline:
        # the compiled reference's dnn_calc_outprob() (FMA path) on a bounded sample; the same rows are the parity check
        from oracle import pyoracle
        with tempfile.TemporaryDirectory(prefix="jamd_dnn_") as td:
            rd = pyoracle.Ref().dnn_load(dnn, td, num_threads=1)
            rd.outprob(frames[:8], want_out=False)
            per = max(rd.last_seconds / 8, 1e-4)
            n = int(min(T, max(16, 10.0 / per)))
            want = rd.outprob(frames[:n], want_out=True)
            sec = rd.last_seconds
            # the reference's OpenMP split of the layer rows (calc_dnn.c:806-833), num_threads = 2 (its
            # default, Sample.dnnconf) and = half the host's logical cores, on ~3 s samples each
            threads = {}
            for nt in sorted({2, max(2, (os.cpu_count() or 2) // 2)}):
                try:
                    with tempfile.TemporaryDirectory(prefix="jamd_dnn_t_") as td2:
                        rt = pyoracle.Ref().dnn_load(dnn, td2, num_threads=nt)
                        rt.outprob(frames[:8], want_out=False)
                        nn = int(min(T, max(16, 3.0 / max(rt.last_seconds / 8, 1e-4))))
                        rt.outprob(frames[:nn], want_out=False)
                        threads[f"t{nt}_rtf_inv"] = nn / 100.0 / rt.last_seconds
                except Exception as e:      # (reported, not fatal: the one-thread figure is the baseline)
                    threads[f"t{nt}_error"] = repr(e)[:80]
        res["parity_spot_check"] = bool(np.array_equal(d_out[:n].cpu().numpy(), want))
        res["cpu_baseline"] = {"threads": threads, "value": n * net.S / sec, "unit": "frame*states/s", "cores": 1, "kind": "reference",
                               "rtf_inv": n / 100.0 / sec,
                               "sample": f"{n} frames: compiled reference dnn_calc_outprob() (calc_dnn_fma, table sigmoid, "
                                         f"log-softmax), {sec:.2f} s on 1 of {os.cpu_count()} host cores"}
    return res


# ------------------------------------------------------------------------------------------------ C3 / C4 end to end
def build_reference_task(workdir: Path, nword: int, beam: int, dnn=None, multipath=False):
    """The C3 / C4 task in the REFERENCE'S OWN FORMATS (HTK hmmdefs + HMMList, HTK dictionary, ARPA 2-gram; for C4
    also the dnnconf with its .npy weight files and the state prior list), then the device blobs through jamd_export
    = Julius' loaders + wchmm builder + our flattening walk (julius_amd/shim/jamd_export.c, built next to the library).
    C4: the hmmdefs carries 4000 one-Gaussian states only to name them -- with -dnnconf the reference scores state i
    with DNN output i (libsent/src/phmm/outprob.c:218-226)."""
    from julius_amd import synth
    if dnn is None:
        task = synth.make_triphone_task(workdir, nphone=40, S=S, M=M, nword=nword, nvar=25, seed=0, maxlen=8,
                                        nbigram_per_word=10)
        am = ["-gprune", "none"]
    else:
        task = synth.make_triphone_task(workdir, nphone=40, S=int(dnn["dims"][-1]), M=1, nword=nword, nvar=25, seed=0,
                                        maxlen=8, nbigram_per_word=10)
        task["dnnconf"] = synth.write_dnnconf(workdir, dnn, context_len=11)
        am = ["-dnnconf", task["dnnconf"], "-notypecheck"]
    jargs = ["-h", task["hmmdefs"], "-hlist", task["hmmlist"], "-v", task["dict"], "-nlr", task["arpa"]] + am + [
        "-input", "htkparam", "-1pass", "-b", str(beam)] + (["-multipath"] if multipath else [])
    export = ROOT / "julius_amd" / "jamd_export"
    if not export.exists():
        return task, jargs, None
    prefix = workdir / "task"
    subprocess.run([str(export)] + [str(a) for a in jargs] + ["-jamdout", str(prefix)], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return task, jargs, prefix


def build_c1_task(workdir: Path, beam: int):
    """BASELINE configs[0] in the reference's own formats: tied-mixture monophone GMM-HMM (HTK ascii, <TMix> codebooks) + a
    100-word loop grammar (.dfa / .dict), exported through Julius' loaders and wchmm.c by jamd_export like C3 / C4."""
    from julius_amd import synth
    task = synth.make_grammar_task(workdir, seed=0)
    jargs = ["-h", task["hmmdefs"], "-dfa", task["dfa"], "-v", task["dict"], "-input", "htkparam", "-gprune", "safe", "-tmix", "2",
             "-1pass", "-b", str(beam), "-penalty1", "-1.0"]
    export = ROOT / "julius_amd" / "jamd_export"
    if not export.exists():
        return task, jargs, None
    prefix = workdir / "task"
    subprocess.run([str(export)] + [str(a) for a in jargs] + ["-jamdout", str(prefix)], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return task, jargs, prefix


def first_pass_traffic(use_dnn, multipath, flat, nutt, beam, shape):
    """HBM bytes of ONE first-pass launch of this configuration as the PMC passes measured them (profiles/traffic_first_pass.json:
    rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate runs, MI355X_MICROARCH.md "HBM" corrections), or None when this
    exact launch (scorer, -multipath, utterances per launch, beam, workgroup shape) was not measured."""
    f = ROOT / "profiles" / "traffic_first_pass.json"
    if not f.exists():
        return None, None
    try:
        for e in json.loads(f.read_text()).get("entries", []):
            if (bool(e.get("dnn")) == bool(use_dnn) and bool(e.get("multipath")) == bool(multipath) and bool(e.get("flat")) == bool(flat)
                    and e.get("utts") == nutt and e.get("beam") == beam and e.get("shape") == shape):
                return e.get("hbm_bytes_per_launch"), ("replayed from profiles/traffic_first_pass.json <- " + str(e.get("source"))
                                                        + " (a separate rocprofv3 --pmc pass of this launch shape, not measured in this run)")
    except Exception:
        return None, None
    return None, None


def first_pass_roofline(work, beam_ms, nutt, shape, sc_ms):
    """SURVEY 8d's substitute figures for the first pass (irregular gather / scatter: no algorithmic-bytes roofline): the
    work of one step counted by the kernel -- tokens created, survivors visited, word ends -- and the bytes that work
    touches under a fixed per-item model, against the LDS bandwidth of the CUs the launch holds
    (128 B/clk/CU for 4-byte operations, MI355X_MICROARCH.md) and against the L2 bandwidth.  The kernel is a dependent-latency
    chain per utterance, so the fraction is small by nature; it is here to be tracked, not to be close to 1."""
    tokens, surv, wends, frames = (float(x) for x in work)
    sec = beam_ms * 1e-3
    # per survivor visited: its record (32 B) + two node records (32 B) + ~2.3 candidates x (16 B cell + 4 B first visit + 8 B arc);
    # per token created: record 32 B + key 4 B + heap entry 8 B + survivor copy 32 B when it is kept
    lds_bytes = surv * (32 + 2.3 * 20) + tokens * (4 + 8 + 8)
    l2_bytes = surv * (32 + 2.3 * 8) + tokens * (32 + 8) + wends * 800
    cus = min(256, nutt if shape != "half" else (nutt + 1) // 2)
    lds_peak = cus * 128 * 2.4e9
    return {"bound": "latency (LDS / L2 gather-scatter)", "achieved": (lds_bytes + l2_bytes) / sec / 1e9 if sec > 0 else None,
            "peak": lds_peak / 1e9, "unit": "GB/s", "frac": (lds_bytes / sec) / lds_peak if sec > 0 else None, "traffic": None,
            "note": "no algorithmic-bytes roofline (SURVEY.md 8d): achieved = modelled LDS + L2 bytes touched per second, peak = LDS "
                    f"bandwidth of the {cus} CUs the launch holds, frac = LDS share of it",
            "tokens_created_per_s": tokens / sec if sec > 0 else None, "survivors_visited_per_s": surv / sec if sec > 0 else None,
            "word_ends_per_s": wends / sec if sec > 0 else None,
            "tokens_created_per_frame": tokens / frames if frames else None, "survivors_per_frame": surv / frames if frames else None,
            "lds_bytes_per_frame": lds_bytes / frames if frames else None, "l2_bytes_per_frame": l2_bytes / frames if frames else None,
            "l2_GBps": l2_bytes / sec / 1e9 if sec > 0 else None,
            "score_kernels_ms": sc_ms, "beam_kernel_ms": beam_ms,
            "timing_note": "HIP events on each stream; under the two-stream pipeline the scoring events include the wait for CUs that the "
                           "first pass of the previous step still holds",
            "beam_frames_per_s": frames / sec if sec > 0 else None, "beam_us_per_frame_per_utt": beam_ms * 1e3 / (frames / max(1, nutt)) if frames else None}


def trellis_diff(a, b):
    """Two canonical trellises (dicts of arrays in (endtime, wid) order): number of entries that are not in both
    or differ in any field."""
    ka = a["endtime"].astype(np.int64) * (1 << 32) + a["wid"]
    kb = b["endtime"].astype(np.int64) * (1 << 32) + b["wid"]
    common, ia, ib = np.intersect1d(ka, kb, return_indices=True)
    diff = (len(ka) - len(common)) + (len(kb) - len(common))
    same = np.ones(len(common), bool)
    for k in a:
        same &= a[k][ia] == b[k][ib]
    return int(diff + (~same).sum())


def run_batch(args, dd: Dist, wd: Path, prefix, task, uniq, use_dnn, beam, launch, nlaunch, NS, multipath=False):
    """The PRODUCT's serving loop timed from outside the kernels: `jamd_batch -d <gpu> -time` (C over the
    C ABI, julius_amd/host/jamd_batch.c) decodes a file list of launch x nlaunch HTK parameter files that sit on tmpfs --
    fread -> byte swap into pinned staging -> asynchronous H2D -> scoring -> first pass -> D2H of the result records ->
    result lines --, double-buffered over two streams.  The clock is the process's own (`decode_s`: from the first file
    read to the last result line, model load excluded and reported beside it).  One "step" = one launch of `launch`
    utterances.  Every rank runs its own process on its own GPU over its own list (weak scaling, like e2e)."""
    exe = ROOT / "julius_amd" / "jamd_batch"
    if prefix is None or not exe.exists():
        return None
    shm = Path("/dev/shm") if os.access("/dev/shm", os.W_OK) else wd
    fd = Path(tempfile.mkdtemp(prefix=f"jamd_batch_r{dd.rank}_", dir=shm))
    from julius_amd import synth
    try:
        paths = []
        for u, fr in enumerate(uniq):
            synth.write_htk_param(fd / f"u{u}.mfc", fr, parmkind=synth.PARM_USER if use_dnn else synth.MFCC_E_D_A)
            paths.append(str(fd / f"u{u}.mfc"))
        nfile = launch * nlaunch
        # the same round-robin deal as run_e2e: this rank's i-th utterance is distinct utterance (rank + i * world) % nuniq
        mine = [paths[(dd.rank + i * dd.world) % len(paths)] for i in range(nfile)]
        (fd / "list.txt").write_text("\n".join(mine) + "\n")
        cmd = [str(exe)] + (["-dnnconf", task["dnnconf"]] if use_dnn else ["-am", str(prefix) + ".am"]) + [
            "-lex", str(prefix) + ".lex", "-b", str(beam), "-filelist", str(fd / "list.txt"), "-d", str(dd.local_rank),
            "-launch", str(launch), "-time"]
        dd.fence()
        t0 = time.perf_counter()
        pr = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, GPU_MAX_HW_QUEUES="16"))
        wall = time.perf_counter() - t0
        tj = None
        for ln in pr.stderr.splitlines():
            if ln.startswith('{"jamd_batch_time"'):
                tj = json.loads(ln)["jamd_batch_time"]
        ok = pr.returncode == 0 and tj is not None
        decode_s = dd.max_over_ranks(tj["decode_s"] if ok else float("inf"))
        if dd.rank != 0:
            return None
        if not ok:
            return {"error": f"jamd_batch rc {pr.returncode}: {pr.stderr[-300:]}"}
        lines = pr.stdout.strip().splitlines()
        frames = tj["frames"] * dd.world
        return {"metric": "frames_x_states_scored_per_sec", "value": frames * NS / decode_s, "unit": "frame*states/s", "n_gpus": dd.world,
                "steps": nlaunch, "warmup": 0, "scaling": "weak", "ms_per_step": decode_s / nlaunch * 1e3, "dtype": "f32",
                "rtf_inv": frames / 100.0 / decode_s,
                "config": {"workload": (f"{'C4' if use_dnn else 'C3'} through the product's serving loop: jamd_batch -time over {nfile} HTK files on "
                                        f"{'tmpfs' if shm != wd else 'the work directory'}, {launch} utterances per launch, beam {beam}"
                                        + (", -multipath" if multipath else "")),
                           "beam": beam, "utts_per_gpu": nfile, "utts_total": nfile * dd.world, "launch": launch},
                "roofline": {"bound": "latency (host loop + LDS / L2 gather-scatter)", "frac": None, "achieved": None, "peak": None,
                             "unit": "GB/s", "traffic": None},
                "timing": {"clock": "jamd_batch's own CLOCK_MONOTONIC: first file read -> last result line (model load excluded)",
                           "decode_s": tj["decode_s"], "models_s": tj["models_s"], "process_wall_s": wall,
                           "host_read_s": tj["host_read_s"], "host_wait_s": tj["host_wait_s"], "h2d_bytes": tj["h2d_bytes"],
                           "h2d_GBps_if_serial": tj["h2d_bytes"] / max(tj["decode_s"], 1e-9) / 1e9, "launches": tj["launches"]},
                "result_lines": len(lines), "result_lines_text": lines[:len(paths)]}
    finally:
        import shutil
        shutil.rmtree(fd, ignore_errors=True)


def run_e2e(args, dd: Dist, runs, use_dnn=False, flat=False, multipath=False, c1=False):
    """configs[2] (GMM) / configs[3] (DNN) end to end on the device: acoustic scores -> exact-order first pass.
    `runs` = list of (key, utterances per GPU, steps, warmup, scaling): every run shares the models, the lexicon and the
    distinct utterances; the FIRST run carries the parity block and the CPU baseline.  Returns {key: result}."""
    import torch
    from julius_amd import lexblob, lib, shard, synth
    eng = lib.Engine(dd.local_rank)
    tmp = tempfile.TemporaryDirectory(prefix="jamd_e2e_")
    wd = Path(tmp.name)
    beam = args.beam if args.beam else (4000 if use_dnn else (200 if c1 else 800))
    # C4: a network whose posteriors are peaked on the state a frame was drawn from (synth.make_decodable_dnn: the first
    # pass ends in a sentence), or -- flat=True, the worst case for the rank pruning step -- random-init weights over noise
    dnn = (synth.make_dnn(seed=0) if flat else synth.make_decodable_dnn(seed=0)) if use_dnn else None
    NS = int(dnn["dims"][-1]) if use_dnn else S
    task, jargs, prefix = build_c1_task(wd, beam) if c1 else build_reference_task(wd, args.nword, beam, dnn, multipath)
    if c1 and prefix is None:
        return {}      # (no Julius tree to build jamd_export from: the C1 line needs the reference's grammar loader)
    if multipath and prefix is None:
        raise SystemExit("bench.py: the multipath workload needs julius_amd/jamd_export (the multipath lexicon is the reference's)")
    ref_built = prefix is not None
    if ref_built:
        lx = lib.Lexicon.from_file(eng, str(prefix) + ".lex")
        info = lexblob.load(str(prefix) + ".lex")
        lexwhat = (f"{len(task['words']) if c1 else args.nword}-word tree lexicon built by the reference (wchmm.c via jamd_export: {info['nnode']} nodes, "
                   f"{info['startnum']} roots, {info['isolatenum']} isolated) + " + ("loop grammar (DFA, per-category trees)" if c1 else "2-gram"))
    else:           # no Julius tree on this box to build jamd_export from: python-made lexicon over the same state inventory
        lex = synth.make_lexicon(nword=args.nword, nphone=40, S=NS, seed=0)
        lx = lib.Lexicon(eng, lex)
        lexwhat = f"{args.nword}-word synthetic tree lexicon ({lex['nnode']} nodes, {lex['startnum']} roots) + 2-gram"
    maxu = max(r[1] for r in runs)
    ndist = max(1, min(maxu * dd.world, args.distinct))      # distinct utterances of the GLOBAL batch: the same list for every N
    if use_dnn:
        scorer = lib.Dnn.from_dnnconf(eng, task["dnnconf"]) if ref_built else lib.Dnn(eng, dnn)
        if flat:    # random-init weights over noise: the scores carry no sentence, every frame saturates the beam
            rng = np.random.default_rng(1000)
            uniq = [rng.normal(0, 1, (FRAMES_PER_UTT, int(dnn["dims"][0]))).astype(np.float32) for _ in range(min(ndist, 16))]
        else:
            uniq = [synth.make_dnn_utterance(task, dnn, nwords=30, seed=u)[0] for u in range(min(ndist, 16))]
        what = (f"DNN {[int(x) for x in dnn['dims']]} (MFMA fp32) outprob, "
                + ("random-init weights over noise frames (flat scores: no sentence, worst case of the rank pruning step)" if flat
                   else "nearest-centroid output layer over random hidden layers (peaked posteriors), frames drawn along word sequences"))
    elif c1:
        # BASELINE configs[0]: tied-mixture monophones (codebook top-N cache + per-state re-weighting, calc_tied_mix.c:162),
        # -gprune safe -tmix 2 as in the C1 tests; 100-word loop grammar
        scorer = lib.Gmm.from_file(eng, str(prefix) + ".am", lib.GPRUNE_SAFE, 2)
        NS = scorer.S
        uniq = [synth.make_grammar_utterance(task, nwords=6, seed=u)[0] for u in range(ndist)]
        what = (f"tied-mixture GMM S={NS} states over {task['model']['nbook']} codebooks x {task['model']['mean'].shape[0] // max(1, task['model']['nbook'])} "
                f"Gaussians x D={D}, -gprune safe -tmix 2 (K2 tmix_book + tmix_state)")
    else:
        scorer = lib.Gmm.from_file(eng, str(prefix) + ".am") if ref_built else lib.Gmm(eng, task["model"])
        uniq = [synth.make_utterance(task, nwords=30, seed=u)[0] for u in range(ndist)]
        what = f"GMM S={S} x M={M} x D={D} outprob"
    nuniq = len(uniq)
    # (C3 at -b 4000 saves 160 000 - 340 000 trellis words per utterance: a work area of 2^18 atoms reports JAMD_PASS1_OVERFLOW
    # for half of them -- loudly, status != 0 -- so that run gets 2^19)
    bm = lib.Beam(eng, lx, beam, -1.0, max_utts=maxu, atoms_per_utt=1 << ((19 if not use_dnn else 18) if beam > 1600 else 17))
    if args.order:
        bm.set_order_mode(args.order)
    mode = bm.order_mode()
    # Two HIP streams, two score buffers: the scoring kernels of step k+1 run on `s_score` while the first pass of step
    # k runs on `s_beam` (it waits for its own step's scores, and a score buffer is rewritten only when the first pass
    # that read it is done).  With fewer utterances than CUs (configs[4] at N = 8: 64 per GPU, one workgroup each) the
    # scoring fills the idle CUs; with a full batch the two compete and the gain is a few percent.
    # The first pass wants whole CUs (full shape) or half CUs (half shape: two workgroups per CU, all of the LDS).  Queued
    # next to it, the scoring workgroups of step k+1 take LDS its workgroups are waiting for (measured: 512 utterances,
    # first pass 200 ms alone, 413 ms; 64 utterances 128 vs 137 ms; stream priorities do not change the picture).  So the
    # host releases the scoring of step k+1 only once the first pass of step k is running (jamd_beam_wait_started() + 1 ms
    # for its workgroups to be placed): the scoring then fills the CUs the first pass does not use or that its shorter
    # utterances leave -- 512 utterances: 304 ms per step against 321 ms with both kernels in one stream.
    s_score, s_beam = torch.cuda.Stream(), torch.cuda.Stream()
    out = {}
    for ri, (key, nutt, steps, warmup, scaling) in enumerate(runs):
        pipelined = not args.no_pipeline
        tailfill = pipelined
        # the batch is a global list (utterance g = distinct utterance g % nuniq) dealt round-robin: rank r holds g = r + u * world
        utts = [uniq[(dd.rank + u * dd.world) % nuniq] for u in range(nutt)]
        off = np.zeros(nutt + 1, np.int32)
        off[1:] = np.cumsum([len(x) for x in utts])
        frames = np.concatenate(utts)
        T = len(frames)
        d_fr = torch.from_numpy(frames).cuda()
        nbuf = 2 if pipelined else 1
        d_scs = [torch.empty((T, NS), dtype=torch.float32, device="cuda") for _ in range(nbuf)]
        d_sc = d_scs[0]
        scored = [torch.cuda.Event() for _ in range(nbuf)]       # buffer b holds this step's scores
        consumed = [torch.cuda.Event() for _ in range(nbuf)]     # the first pass that read buffer b is done
        for e in consumed:
            e.record(s_beam)
        torch.cuda.synchronize()
        count = [0]
        sb = s_beam if pipelined else s_score

        def step(mark):
            b = count[0] % nbuf
            count[0] += 1
            s_score.wait_event(consumed[b])
            if tailfill:
                bm.stream_wait_resident(s_score.cuda_stream)    # the scoring starts once the previous step's first pass holds its CUs
            if mark:
                mark[0].record(s_score)
            scorer.outprob_dev(d_fr.data_ptr(), T, d_scs[b].data_ptr(), s_score.cuda_stream)
            if mark:
                mark[1].record(s_score)
            scored[b].record(s_score)
            sb.wait_event(scored[b])
            if mark:
                mark[2].record(sb)
            bm.pass1_dev(d_scs[b].data_ptr(), NS, off, sb.cuda_stream)
            if mark:
                mark[3].record(sb)
            consumed[b].record(sb)

        elapsed, ev = timed_steps(dd, None, step, steps, warmup, nevents=4)
        d_sc = d_scs[(count[0] - 1) % nbuf]                      # the scores of the last step (parity leg)
        sc_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in ev]))
        beam_ms = float(np.mean([e[2].elapsed_time(e[3]) for e in ev]))
        beam_ms_steps = [round(float(e[2].elapsed_time(e[3])), 3) for e in ev]       # the spread over the timed steps (detail output)
        res_local = bm.results()
        # per-step counters of the exact-order kernel: the paths of utterance 0's pruning steps, the work of all utterances
        nlaunch = max(1, steps + warmup)
        if mode.startswith("exact"):
            allst = np.array([bm.prune_stats(u, reset=True) for u in range(nutt)], dtype=np.int64) // nlaunch
            pstats, work = [int(x) for x in allst[0][:8]], allst[:, 8:12].sum(axis=0)
            mpstat = [int(x) for x in allst[:, 12:14].sum(axis=0)]
        else:
            pstats, work, mpstat = [0] * 8, np.zeros(4, np.int64), [0, 0]
        # the only collective of the job: every rank gets the per-utterance result records of all ranks (RCCL over xGMI)
        nutt_all = nutt * dd.world
        if dd.world > 1:
            # round-robin table: utterance g lives on rank g % world; this rank's u-th utterance is g = rank + u * world
            table = shard.gather_results(shard.pack_results(res_local), nutt_all, dd.rank, dd.world, device=dd.coll_device)
        else:
            table = shard.pack_results(res_local)
        frames_all = dd.sum_over_ranks(T)                       # the ranks' shares differ in length: the job's frames are their sum
        if dd.rank == 0 and ri == 0 and args.dump_results:
            np.savez(args.dump_results, table=np.asarray(table, np.int32), world=dd.world, utts_per_rank=nutt, frames_all=frames_all,
                     rank0_utts=np.arange(dd.rank, nutt_all, dd.world))
        if dd.rank == 0:
            st = np.asarray(table)[:, 0]
            total_frames = frames_all * steps
            cfg = "C4 (BASELINE.json configs[3])" if use_dnn else ("C1 (BASELINE.json configs[0])" if c1 else "C3 (BASELINE.json configs[2])")
            if multipath:
                cfg += " decoded with -multipath (non-emitting word-begin / word-end nodes, the reference's two-half frame: beam.c:2747-2836)"
            if scaling == "strong":
                cfg += f" as configs[4]: the fixed batch of {nutt_all} utterances sharded over {dd.world} GPU(s)"
            r = {"metric": "frames_x_states_scored_per_sec", "value": total_frames * NS / elapsed,
                 "unit": "frame*states/s", "n_gpus": dd.world, "steps": steps, "warmup": warmup, "scaling": scaling,
                 "ms_per_step": elapsed / steps * 1e3, "dtype": "f32", "rtf_inv": total_frames / 100.0 / elapsed,
                 "frames_per_s": total_frames / elapsed, "frames_per_s_per_gpu": total_frames / elapsed / dd.world,
                 "config": {"workload": (f"{'C4' if use_dnn else ('C1' if c1 else 'C3')} = configs[{3 if use_dnn else (0 if c1 else 2)}]"
                                         + (" as configs[4] (fixed batch)" if scaling == "strong" else "")
                                         + f": {'DNN' if use_dnn else ('tied-mixture GMM' if c1 else 'GMM')} scores + HIP first pass, "
                                         + f"{len(task['words']) if c1 else args.nword} words{' (grammar)' if c1 else ''}, beam {beam}"
                                         + (", -multipath" if multipath else "") + (", flat scores" if flat else "")
                                         + f", {nutt} utts/GPU/step"),
                            "detail": f"{cfg}: {what} + HIP first pass ({mode} tie order), {lexwhat}, beam {beam}, "
                                      f"{nutt} utterances ({T} frames, {nuniq} distinct) per GPU per step",
                            "lexicon_built_by_reference": ref_built, "order_mode": mode, "beam": beam,
                            "utts_per_gpu": nutt, "utts_total": nutt_all,
                            "workgroup_shape": bm.workgroup_shape(nutt) + (" (two utterances per CU)" if bm.workgroup_shape(nutt) == "half" else " (one utterance per CU)"),
                            "pipelined": ("scoring of step k+1 on a second stream, released once the first pass of step k is running (it fills the CUs "
                                          "the first pass leaves)" if pipelined else "no")},
                 "roofline": dict(first_pass_roofline(work, beam_ms, nutt, bm.workgroup_shape(nutt), sc_ms),
                                  **dict(zip(("traffic", "traffic_source"), first_pass_traffic(use_dnn, multipath, flat, nutt, beam, bm.workgroup_shape(nutt))))),
                 "pass1": {"ok": int((st == 0).sum()), "no_sentence": int((st == 1).sum()), "utts": nutt_all,
                           "mean_peak_tokens": float(np.mean([x.max_tokens for x in res_local])),
                           "ties_counted": int(sum(x.ties for x in res_local)), "phase_us_utt0": list(res_local[0].phase_us),
                           "prune_paths_utt0": dict(zip(("frames_pruned", "up_closed_form", "up_wave_replay", "up_sweep", "up_sweep_gave_up",
                                                         "down_closed_form", "extraction_loop", "sweep_rounds"), pstats))}}
            r["roofline"]["beam_kernel_ms_steps"] = beam_ms_steps
            if scaling == "strong":   # 
                r["strong_scaling_note"] = ("no N > 1 run exists (one GPU per lease; the N > 1 orchestration is tested on one device with gloo); a fixed "
                                            "batch of 512 utterances cannot beat its longest utterance alone (1 611 frames x ~80 us = ~128 ms against "
                                            f"this N = 1 step): ~2.2x at 8 GPUs, the weak line scales by construction")
            if multipath:      # frames whose new tokens exceeded the beam (the mid-frame sort really sorted)
                r["pass1"]["multipath_frames"] = {"mid_frame_sorted": mpstat[0], "all": int(work[3])}
            if ri == 0 and dd.world == 1 and not args.no_cpu_baseline:
                r["parity"], cpu = e2e_parity(bm, d_sc, NS, off, uniq, nuniq, res_local, jargs, wd, ref_built, use_dnn, multipath)
                if cpu is not None:
                    r["cpu_baseline"] = cpu
            out[key] = r
        if ri == 0:
            first_key, first_nutt, first_res = key, nutt, res_local
        del d_fr, d_sc, d_scs
    if runs and ref_built and not flat and not c1 and not args.no_batch and mode.startswith("exact"):
        # the same task through the product's own host loop (file read + pinned staging + H2D + kernels + D2H), own clock.
        # AFTER the in-process runs: the exit of a process that held 30 GB of device and pinned memory disturbs the next
        # second of this process's kernels (round 5: one 360 - 440 ms step among e2e_strong's 205 ms steps when it ran between them)
        key, nutt, res_local = first_key, first_nutt, first_res
        bl = min(nutt, 512)
        rb = run_batch(args, dd, wd, prefix, task, uniq, use_dnn, beam, bl, args.batch_launches or (3 if (use_dnn and multipath) else 4), NS, multipath)
        if dd.rank == 0 and rb is not None:
            if "result_lines_text" in rb:       # the serving loop's result lines against the in-process results of the same utterances
                same = 0
                txt = rb.pop("result_lines_text")
                for u in range(min(nuniq, len(txt))):
                    rr = res_local[u] if dd.world == 1 else None
                    if rr is None:
                        break
                    words = " ".join(str(int(w)) for w in rr.wseq[:rr.wnum])
                    want_tail = f"status={rr.status} score={float(rr.score):.9g} words={words}"
                    same += int(txt[u].split(" ", 1)[1].strip() == want_tail)
                if dd.world == 1:
                    rb["parity"] = {"result_lines_vs_in_process": {"utts": min(nuniq, len(txt)), "identical": same}}
                rb["vs_e2e_same_task"] = rb["rtf_inv"] / out[key]["rtf_inv"]
            out["batch" + key[3:]] = rb
    bm.close()
    tmp.cleanup()
    return out


def e2e_parity(bm, d_sc, NS, off, uniq, nuniq, res_exact, jargs, wd, ref_built, use_dnn, multipath=False):
    """Checker leg (after the timed region).  (1) The device first pass against the COMPILED REFERENCE's
    (julius -1pass over the same files: for C4 that is the reference's own dnn_calc_outprob() + beam.c): word trellis
    entry by entry, pass-1 sentence, score -- which is also the lazy-scoring CPU baseline of SURVEY 8d.  (2) exact-order
    kernel against the canonical-tie kernel on every distinct utterance."""
    from julius_amd import lexblob, synth
    from oracle import pyoracle
    canon_exact = [lexblob.canonical_trellis(bm.trellis(u)) for u in range(nuniq)]
    par = {"device_mode": bm.order_mode()}
    mode0 = bm.order_mode()

    def sent(r):
        return list(r.wseq[:r.wnum]) if r.status == 0 else None

    if mode0 != "fast" and not multipath:              # (the canonical-tie kernel does not take multipath lexicons)
        bm.set_order_mode("fast")
        bm.pass1_dev(d_sc.data_ptr(), NS, off)
        res_fast = bm.results()
        fast = {"utts": nuniq, "trellis_identical": 0, "pass1_sentence_identical": 0, "score_identical": 0, "atoms_differing": []}
        for u in range(nuniq):
            d = trellis_diff(canon_exact[u], lexblob.canonical_trellis(bm.trellis(u)))
            fast["trellis_identical"] += int(d == 0)
            fast["atoms_differing"].append(d)
            fast["pass1_sentence_identical"] += int(sent(res_fast[u]) == sent(res_exact[u]))
            fast["score_identical"] += int(res_fast[u].status == res_exact[u].status and
                                           (res_fast[u].status != 0 or res_fast[u].score == res_exact[u].score))
        fast["ties_counted_by_fast_kernel"] = int(sum(r.ties for r in res_fast[:nuniq]))
        par["fast_kernel_vs_" + mode0 + "_kernel"] = fast
        bm.set_order_mode(mode0)
    cpu = None
    if ref_built:
        # (1) the compiled reference, lazy scoring, ONE CORE PER PROCESS: the utterances are dealt to worker processes
        # (at most 16, one core each), so that all distinct C3 utterances and 12 of C4's are compared within the run;
        # the one-core figure below is frames / core-seconds summed over the workers
        want = min(nuniq, 16) if use_dnn else nuniq           # every distinct utterance (C4 has 16; one worker process each)
        nproc = max(1, min(16, want, (os.cpu_count() or 2) // 2))
        specs = []
        for u in range(want):
            synth.write_htk_param(wd / f"ref_u{u}.mfc", uniq[u], parmkind=synth.PARM_USER if use_dnn else synth.MFCC_E_D_A)
        for w in range(nproc):
            mine = list(range(w, want, nproc))
            spec = {"jargs": [str(a) for a in jargs], "utts": mine, "files": [str(wd / f"ref_u{u}.mfc") for u in mine],
                    "out": str(wd / f"ref_out{w}.npz")}
            (wd / f"ref_spec{w}.json").write_text(json.dumps(spec))
            specs.append(spec)
        t0 = time.perf_counter()
        # the FULL two-pass recogniser on a few of the same files (the first pass + Julius' stack decoder),
        # timed by two extra one-core workers next to the others; their results are not part of the comparison
        n2 = min(want, 2 if use_dnn else 4)
        specs2 = []
        for w in range(min(2, n2)):
            mine = list(range(w, n2, 2))
            spec = {"jargs": [str(a) for a in jargs if a != "-1pass"], "utts": mine, "files": [str(wd / f"ref_u{u}.mfc") for u in mine],
                    "out": str(wd / f"ref2_out{w}.npz")}
            (wd / f"ref2_spec{w}.json").write_text(json.dumps(spec))
            specs2.append(spec)
        errs = [open(wd / f"ref_worker{w}.err", "wb") for w in range(nproc + len(specs2))]
        procs = [subprocess.Popen([sys.executable, str(ROOT / "bench.py"), "--ref-e2e-worker", str(wd / f"ref_spec{w}.json")],
                                  stdout=subprocess.DEVNULL, stderr=errs[w]) for w in range(nproc)]
        procs += [subprocess.Popen([sys.executable, str(ROOT / "bench.py"), "--ref-e2e-worker", str(wd / f"ref2_spec{w}.json")],
                                   stdout=subprocess.DEVNULL, stderr=errs[nproc + w]) for w in range(len(specs2))]
        for pr in procs:
            pr.wait()
        for f in errs:
            f.close()
        wall = time.perf_counter() - t0
        vs = {"wanted": want, "utts": 0, "trellis_identical": 0, "pass1_sentence_identical": 0, "score_identical": 0, "atoms_differing": [],
              "reference_atoms": [], "reference_found_a_sentence": 0}
        spent, frames_done, load_s, slowest = 0.0, 0, 0.0, 0.0
        for w, spec in enumerate(specs):
            if not Path(spec["out"]).exists():        # a worker died: its utterances are not counted, and the block says so
                tail = (wd / f"ref_worker{w}.err").read_bytes()[-400:].decode("utf-8", "replace")
                vs.setdefault("workers_failed", []).append({"worker": w, "rc": procs[w].returncode, "utts": spec["utts"], "stderr_tail": tail})
                continue
            z = np.load(spec["out"])
            load_s = max(load_s, float(z["load_s"]))
            slowest = max(slowest, sum(float(z[f"sec_{u}"]) for u in spec["utts"]))
            for u in spec["utts"]:
                rtr = {k[len(f"tr_{u}_"):]: z[k] for k in z.files if k.startswith(f"tr_{u}_")}
                rw, rs = z[f"w_{u}"], float(z[f"s_{u}"])
                spent += float(z[f"sec_{u}"])
                frames_done += len(uniq[u])
                rfound = len(rw) > 0                       # (a failed first pass leaves pass1_wnum = 0)
                d = trellis_diff(canon_exact[u], rtr)
                vs["utts"] += 1
                vs["trellis_identical"] += int(d == 0)
                vs["atoms_differing"].append(d)
                vs["reference_atoms"].append(int(len(rtr["wid"])))
                vs["reference_found_a_sentence"] += int(rfound)
                dsent = sent(res_exact[u])
                vs["pass1_sentence_identical"] += int((dsent == list(rw)) if rfound else (dsent is None))
                vs["score_identical"] += int((res_exact[u].status == 0 and float(res_exact[u].score) == float(rs)) if rfound
                                             else res_exact[u].status != 0)
        vs["complete"] = vs["utts"] == vs["wanted"]
        par["device_vs_compiled_reference"] = vs
        what = ("julius -1pass (compiled reference: dnn_calc_outprob FMA path, 1 thread, + get_back_trellis_proceed)" if use_dnn
                else "julius -1pass (compiled reference: lazy outprob cache + get_back_trellis_proceed)")
        cpu = {"value": frames_done * NS / spent, "unit": "frame*states/s" + ("" if use_dnn else " (nominal: the lazy search scores only the states it visits)"),
               "cores": 1, "kind": "reference", "rtf_inv": frames_done / 100.0 / spent,
               "sample": f"{what} on {vs['utts']} utterances = {frames_done} frames, {spent:.1f} core-seconds over {nproc} worker processes of "
                         f"one core each ({wall:.1f} s wall, {os.cpu_count()} host cores; model load {load_s:.1f} s not counted)"}
        if slowest > 0:     # the same workers read as ONE N-process job: all frames / the slowest worker's recognition time
            cpu["multi"] = {"cores": nproc, "rtf_inv": frames_done / 100.0 / slowest, "value": frames_done * NS / slowest,
                            "note": f"{nproc} julius -1pass processes side by side, one core each: frames of all / recognition time of the slowest"}
        sec2, fr2, n2done = 0.0, 0, 0
        for spec in specs2:
            if Path(spec["out"]).exists():
                z = np.load(spec["out"])
                for u in spec["utts"]:
                    sec2 += float(z[f"sec_{u}"]); fr2 += len(uniq[u]); n2done += 1
        if sec2 > 0:
            cpu["two_pass"] = {"cores": 1, "rtf_inv": fr2 / 100.0 / sec2, "utts": n2done,
                               "note": "the full recogniser (first pass + stack decoding, no -1pass) on the same files, one core"}
    return par, cpu


# ------------------------------------------------------------------------------------------------ main
def emit(full):
    """stdout contract: every nested block in full as its own EARLIER line ({"bench_detail": key, ...}), the whole tree in
    bench_detail.json (next to bench.py, and under gpurun_out/ when that exists), and as the LAST line the compact record
    the driver parses (julius_amd/benchfmt.py: contract keys + a short record per nested configuration, < 6 KB)."""
    from julius_amd import benchfmt
    for k, v in full.items():
        if isinstance(v, dict) and "ms_per_step" in v:
            print(json.dumps({"bench_detail": k, **v}), flush=True)
    print(json.dumps({"bench_detail": "top", **{k: v for k, v in full.items() if not (isinstance(v, dict) and "ms_per_step" in v)}}),
          flush=True)
    for d in (ROOT, ROOT / "gpurun_out"):
        if d.is_dir():
            try:
                (d / "bench_detail.json").write_text(json.dumps(full, indent=1))
            except OSError:
                pass
    full = dict(full, detail_file="bench_detail.json")
    print(benchfmt.final_line(full), flush=True)


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (one per GPU) and relay."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(ROOT / "bench.py")] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps of the top-level workload (default: enough for >= 2 s)")
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--utts", type=int, default=None,
                    help="utterances per GPU per step (gmm/dnn: x1000 frames, default 64 = one GPU's share of the "
                         "512-utterance batch of configs[4]; e2e: default 512)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-small-T", dest="no_small_T", action="store_true", help="skip the k1_small_T calls of the gmm workload (PMC passes: one launch size per kernel name)")
    ap.add_argument("--no-batch", action="store_true", help="e2e: skip the jamd_batch (product serving loop) run of the same task")
    ap.add_argument("--batch-launches", type=int, default=None, help="e2e: launches of the jamd_batch run (default 4; DNN -multipath 3)")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="e2e: scoring and first pass of every step on ONE stream (default: two streams, the scoring of step k+1 "
                         "overlaps the first pass of step k)")
    ap.add_argument("--workload", default="all", choices=["all", "gmm", "dnn", "e2e", "e2e-dnn"],
                    help="all (default) = gmm at top level with e2e (C3), e2e_strong (C5 batch), e2e_dnn (C4) and dnn (C4 scoring "
                         "half) nested; gmm = BASELINE configs[1] alone; dnn = configs[3] scoring half; e2e = configs[2] "
                         "(--strong: configs[4]); e2e-dnn = configs[3] end to end")
    ap.add_argument("--beam", type=int, default=None,
                    help="e2e: rank beam (-b; default 800 = the reference's default for triphone models, e2e-dnn: 4000 = the "
                         "reference's DNN recipe)")
    ap.add_argument("--strong", action="store_true",
                    help="e2e: run configs[4] as specified -- the FIXED batch of --batch-total utterances sharded over the GPUs "
                         "(strong scaling) -- instead of --utts utterances per GPU")
    ap.add_argument("--flat", action="store_true",
                    help="e2e-dnn: the flat-score stream (random-init weights over noise frames: nothing decodes, every frame "
                         "saturates the beam) instead of the input that decodes")
    ap.add_argument("--multipath", action="store_true",
                    help="e2e / e2e-dnn: the task decoded with -multipath (reference-built multipath lexicon; the exact-order kernel's "
                         "multipath frame, csrc/beam_exact_mp.h)")
    ap.add_argument("--batch-total", type=int, default=None, help="e2e --strong: utterances in the fixed batch (default 512)")
    ap.add_argument("--nword", type=int, default=20000, help="e2e: vocabulary size")
    ap.add_argument("--distinct", type=int, default=32, help="e2e: distinct utterances in the batch")
    ap.add_argument("--order", default=None, choices=["fast", "strict", "exact", "exact_serial"],
                    help="e2e: first-pass tie order mode (default: the work area's default = exact)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend of an N > 1 run (nccl = RCCL over xGMI; gloo: barrier / clock / gather over the host, "
                         "with --share-device the way the N > 1 orchestration runs on a one-GPU box)")
    ap.add_argument("--share-device", action="store_true", help="every rank uses cuda:0 (one-GPU box; needs --dist-backend gloo)")
    ap.add_argument("--dump-results", default=None,
                    help="e2e: rank 0 writes the gathered per-utterance result table of the first run (int32 [utts][4 + 150]: status, "
                         "words, frames, score bits, word ids) and each rank's utterance ids to this .npz")
    ap.add_argument("--cpu-worker", nargs=2, default=None, help=argparse.SUPPRESS)
    ap.add_argument("--ref-e2e-worker", default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_worker:
        print(json.dumps(ref_gmm_worker(int(args.cpu_worker[0]), int(args.cpu_worker[1]))))
        return 0
    if args.ref_e2e_worker:
        print(json.dumps(ref_e2e_worker(args.ref_e2e_worker)))
        return 0
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args)

    dd = Dist(args.gpus, args.dist_backend, args.share_device)
    wl = args.workload
    nested = wl == "all"

    def pick(v, d):
        return d if (v is None or nested) else v

    def top(r):
        r.update({"higher_is_better": True, "vs_baseline": None, "data": "synthetic"})
        r.setdefault("scaling", "weak")
        return r

    line = None
    if wl in ("all", "gmm"):
        a = argparse.Namespace(**vars(args))
        a.utts = args.utts or 64
        # one step = 12 launches ~ 120 ms: 20 steps time >= 2 s; --steps/--warmup address this (the contract's) workload
        line = run_gmm(a, dd, args.steps if args.steps is not None else 20, args.warmup if args.warmup is not None else 2)
    if wl in ("all", "e2e"):
        # configs[2] weak (fixed utterances per GPU) and configs[4] strong (the fixed 512-utterance batch sharded over the
        # GPUs: 64 per GPU at N = 8, all 512 on one GPU at N = 1), same models, lexicon and utterances
        # weak: 512 utterances per GPU per step -- two per CU, the exact-order kernel's half shape (throughput); the
        # 256-utterance step of rounds 2-3 (one per CU, full shape: the latency-optimal launch) rides along as e2e_256
        per_gpu = pick(args.utts, 512)
        strong_total = args.batch_total or C5_TOTAL_UTTS
        runs = []
        if wl == "all" or not (args.strong or args.multipath):
            runs.append(("e2e", per_gpu, pick(args.steps, 10), pick(args.warmup, 1), "weak"))
        if wl == "all" or (args.strong and not args.multipath):
            runs.append(("e2e_strong", max(1, strong_total // dd.world), pick(args.steps, 8), pick(args.warmup, 1), "strong"))
        if not args.strong and not args.multipath and args.utts is None:
            runs.append(("e2e_256", 256, pick(args.steps, 10), pick(args.warmup, 1), "weak"))
        r = run_e2e(args, dd, runs, use_dnn=False) if runs else {}
        if wl == "all":
            # SURVEY 8d's table lists C3 at -b 800 AND -b 4000: 256 utterances, wide layout, parity vs julius -1pass -b 4000
            a4 = argparse.Namespace(**vars(args)); a4.beam = 4000; a4.no_batch = True
            r.update(run_e2e(a4, dd, [("e2e_b4000", 256, 3, 1, "weak")], use_dnn=False))
            # BASELINE configs[0] (the reference's CPU-runnable plumbing case) on the device: tied-mixture scoring + grammar first pass
            r.update(run_e2e(args, dd, [("c1", 512, 5, 1, "weak")], c1=True))
        if wl == "all" or args.multipath:
            # the same task decoded with -multipath: 512 utterances per step, two per CU (the multipath frame in its half shape, round 5)
            r.update(run_e2e(args, dd, [("e2e_mp", pick(args.utts, 512), pick(args.steps, 5), pick(args.warmup, 1), "weak")],
                             use_dnn=False, multipath=True))
        if dd.rank == 0:
            if nested:
                line.update(r)
            else:
                ks = list(r)
                line = top(r[ks[0]])
                for k in ks[1:]:
                    line[k] = r[k]
    if wl in ("all", "e2e-dnn"):
        # configs[3] end to end at the reference recipe's beam (-b 4000), reference-built lexicon, parity vs julius -1pass:
        # the input that decodes (weak line + the configs[4] strong line), then the flat-score stream of round 3 as the
        # labelled worst case of the rank pruning step
        strong_total = args.batch_total or C5_TOTAL_UTTS
        runs = []
        if wl == "all" or not (args.strong or args.flat or args.multipath):
            runs.append(("e2e_dnn", pick(args.utts, 256), pick(args.steps, 5), pick(args.warmup, 1), "weak"))
        if wl == "all" or (args.strong and not args.flat and not args.multipath):
            runs.append(("e2e_dnn_strong", max(1, strong_total // dd.world), pick(args.steps, 4), pick(args.warmup, 1), "strong"))
        r = run_e2e(args, dd, runs, use_dnn=True) if runs else {}
        if wl == "all" or args.multipath:
            # the reference's DNN recipe as its README gives it: -b 4000 WITH -multipath (the multipath frame, wide layout)
            r.update(run_e2e(args, dd, [("e2e_dnn_mp", pick(args.utts, 256), pick(args.steps, 3), pick(args.warmup, 1), "weak")],
                             use_dnn=True, multipath=True))
        if wl == "all" or (args.flat and not args.multipath):
            r.update(run_e2e(args, dd, [("e2e_dnn_flat", pick(args.utts, 256), pick(args.steps, 4), pick(args.warmup, 1), "weak")],
                             use_dnn=True, flat=True))
        if dd.rank == 0:
            if nested:
                line.update(r)
            else:
                ks = list(r)
                line = top(r[ks[0]])
                for k in ks[1:]:
                    line[k] = r[k]
    if wl in ("all", "dnn"):
        a = argparse.Namespace(**vars(args))
        a.utts = pick(args.utts, 64)
        r = run_dnn(a, dd, pick(args.steps, 60), pick(args.warmup, 3))
        if dd.rank == 0:
            if nested:
                line["dnn"] = r
            else:
                line = top(r)
    if dd.rank == 0:
        emit(line)
    dd.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
