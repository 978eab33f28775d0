"""CPU, world_size 2, gloo: the batch-of-utterances sharding used for N > 1 GPUs
(julius_amd/shard.py).  Each rank decodes its shard -- here with the CPU oracle
standing in for the device engine, which is what a `not gpu` test may use -- and
the gathered table must equal the serial decode of the whole batch."""
import os
import socket
from types import SimpleNamespace

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from beamutil import load_beam_golden
from julius_amd import shard


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _decode(oracle, g, scores):
    atoms, wseq, score, rc, died = oracle.beam_pass1(g["lex"], scores, g["beam_width"], g["score_pruning_width"])
    return SimpleNamespace(status=rc, wnum=len(wseq), frames=len(scores), score=score, wseq=list(wseq))


def _worker(rank, world, port, nutt, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import pyoracle
    orc = pyoracle.Oracle()
    g = load_beam_golden("beam_rank.npz")
    mine = shard.shard_indices(nutt, rank, world)
    res = []
    for u in mine:
        fr = g["utts"][u % len(g["utts"])]["frames"][: 60 + 7 * int(u)]
        res.append(_decode(orc, g, orc.gmm_outprob(g["am"], fr)))
    table = shard.gather_results(shard.pack_results(res), nutt, rank, world)
    frames, _ = shard.reduce_counters(sum(r.frames for r in res), 0)
    if rank == 0:
        q.put((table, frames))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_indices_partition():
    for nutt in (0, 1, 5, 8, 513):
        for world in (1, 2, 4, 8):
            allidx = np.concatenate([shard.shard_indices(nutt, r, world) for r in range(world)])
            assert sorted(allidx.tolist()) == list(range(nutt))
            sizes = [len(shard.shard_indices(nutt, r, world)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1


def test_pack_roundtrip():
    r = SimpleNamespace(status=0, wnum=3, frames=77, score=-1234.5678, wseq=[0, 42, 1])
    back = shard.unpack_results(shard.pack_results([r]))[0]
    assert back["status"] == 0 and back["frames"] == 77 and back["wseq"].tolist() == [0, 42, 1]
    assert back["score"] == float(np.float32(-1234.5678))


@pytest.mark.timeout(300)
def test_two_ranks_equal_serial(oracle):
    nutt, world = 5, 2
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, nutt, q)) for r in range(world)]
    for p in procs:
        p.start()
    table, frames = q.get()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    g = load_beam_golden("beam_rank.npz")
    want, total = [], 0
    for u in range(nutt):
        fr = g["utts"][u % len(g["utts"])]["frames"][: 60 + 7 * u]
        want.append(_decode(oracle, g, oracle.gmm_outprob(g["am"], fr)))
        total += len(fr)
    assert np.array_equal(table, shard.pack_results(want))
    assert frames == total


# ---------------------------------------------------------------------------------- the same with the device engine
def _gpu_worker(rank, world, port, workdir, nutt, q):
    """A rank of the N-GPU job with the PRODUCT doing the work: `jamd_batch -shard rank world` (C, C ABI only) decodes
    this rank's utterances on the device; the ranks exchange nothing but the result records (gloo here, RCCL in bench.py)."""
    import subprocess
    from julius_amd import lib
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = load_beam_golden("beam_rank.npz")
    out = subprocess.run([str(lib._PKG / "jamd_batch"), "-am", f"{workdir}/am.blob", "-lex", f"{workdir}/lex.blob", "-filelist",
                          f"{workdir}/list", "-b", str(g["beam_width"]), "-shard", str(rank), str(world)],
                         check=True, capture_output=True, text=True).stdout.strip().splitlines()
    mine = shard.shard_indices(nutt, rank, world)
    assert len(out) == len(mine)
    res = []
    for ln in out:
        f = ln.split(" ", 3)
        ws = [int(x) for x in f[3].split("=", 1)[1].split()]
        res.append(SimpleNamespace(status=int(f[1].split("=")[1]), wnum=len(ws), frames=0, score=float(f[2].split("=")[1]), wseq=ws))
    table = shard.gather_results(shard.pack_results(res), nutt, rank, world)
    if rank == 0:
        q.put(table)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_two_ranks_drive_the_device_batch_driver(oracle, tmp_path):
    """world 2 on one device: every rank runs jamd_batch over its shard, the gathered table equals the oracle's serial
    decode of the whole batch (the same check as above with the device engine in place of the oracle)."""
    from julius_amd import lexblob, synth
    g = load_beam_golden("beam_rank.npz")
    nutt, world = 5, 2
    lexblob.save_gmm(g["am"], tmp_path / "am.blob")
    lexblob.save(g["lex"], tmp_path / "lex.blob")
    names, frames = [], []
    for u in range(nutt):
        fr = g["utts"][u % len(g["utts"])]["frames"][: 60 + 7 * u]
        frames.append(fr)
        names.append(str(tmp_path / f"u{u}.mfc"))
        synth.write_htk_param(names[-1], fr)
    (tmp_path / "list").write_text("\n".join(names) + "\n")
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_gpu_worker, args=(r, world, port, str(tmp_path), nutt, q)) for r in range(world)]
    for p in procs:
        p.start()
    table = q.get()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    want = [_decode(oracle, g, oracle.gmm_outprob(g["am"], fr)) for fr in frames]
    for w in want:
        w.frames = 0
    assert np.array_equal(table, shard.pack_results(want))
