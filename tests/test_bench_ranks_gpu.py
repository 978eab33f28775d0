"""GPU: the N > 1 orchestration of bench.py on a ONE-GPU box.  RCCL refuses two ranks on one device, so the
two ranks share cuda:0 and use gloo for what the job needs from torch.distributed -- the barrier, the max-over-ranks clock,
the gather of the per-utterance result records: `bench.py --gpus 2 --dist-backend gloo --share-device`.  What runs on hardware
here: spawn_ranks() (torch.distributed.run, 127.0.0.1), Dist, the strong split of configs[4]'s fixed batch (utterance g on
rank g % N), two engines / models / work areas side by side on the device, shard.gather_results() and the whole-job figures.
No N > 1 run on N GPUs exists in this repo (one GPU per box); the nccl branch differs from this one in the backend string
and in where the 600-byte records live during the gather."""
import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _run(extra, tmp_path, name):
    out = tmp_path / f"{name}.npz"
    cmd = [sys.executable, str(ROOT / "bench.py"), "--workload", "e2e", "--strong", "--batch-total", "12", "--nword", "1500",
           "--distinct", "6", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-batch", "--dump-results", str(out)] + extra
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=str(tmp_path))
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    return json.loads(lines[-1]), np.load(out)


def test_two_ranks_on_one_device_equal_the_single_rank_job(tmp_path):
    if not (ROOT / "julius_amd" / "jamd_export").exists():
        pytest.skip("julius_amd/jamd_export not built (the task's lexicon is the reference's)")
    one, z1 = _run([], tmp_path, "n1")
    two, z2 = _run(["--gpus", "2", "--dist-backend", "gloo", "--share-device"], tmp_path, "n2")
    # the fixed batch is dealt round-robin: 6 utterances per rank, rank 0 holds g = 0, 2, 4, ...
    assert int(z1["world"]) == 1 and int(z1["utts_per_rank"]) == 12
    assert int(z2["world"]) == 2 and int(z2["utts_per_rank"]) == 6
    assert z2["rank0_utts"].tolist() == [0, 2, 4, 6, 8, 10]
    # the gathered table (status, words, frames, score bits, word ids per utterance, in utterance order) is the one-rank job's
    assert z1["table"].shape == z2["table"].shape == (12, 4 + 150)
    assert np.array_equal(z1["table"], z2["table"])
    assert (z1["table"][:, 0] == 0).all()                                  # every first pass ended in a sentence
    # whole-job figures: frames summed over the ranks, one line from rank 0
    assert int(z1["frames_all"]) == int(z2["frames_all"]) == int(z1["table"][:, 2].sum())
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["scaling"] == "strong"
    for line, z in ((one, z1), (two, z2)):
        frames = int(z["frames_all"]) * line["steps"]
        assert line["value"] == pytest.approx(frames * 3000 / (line["ms_per_step"] * line["steps"] * 1e-3), rel=1e-3)
