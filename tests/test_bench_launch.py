"""`python bench.py --gpus N` — the driver's command line for the scaling curve — must start N ranks BY ITSELF when no launcher
set WORLD_SIZE (as shipped it measured one GPU whatever N).  Checked here without a GPU through
`--dry-run-dist`: the same self-launch (torch.distributed.run, 127.0.0.1 rendezvous), the same BatchedExchange protocol and
barrier / max-over-ranks timing as the knn and c5 legs, on CPU tensors over gloo."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*argv):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, f"stdout must carry exactly one JSON line, got {len(lines)}: {r.stdout[:500]}"
    return json.loads(lines[0]), r.stderr


def test_gpus_2_self_launches_two_ranks():
    out, err = _run("--gpus", "2", "--dry-run-dist")
    assert out["n_gpus"] == 2 and out["dry_run"] is True and out["value"] is None
    ex = out["config"]["exchange"]
    assert ex["ranks"] == 2 and ex["gathered_slots_verified"] is True and ex["collectives"] >= 1
    assert out["config"]["launched_by"] == "bench.py self_launch"
    assert "launching 2 ranks" in err
    # the default workload at N > 1 also runs BASELINE configs[4]: its partition rides in config.secondary —
    # 255 sequential pairs (sfm.py:347) of 256 images split 128 / 127, every rank holding its images + ONE halo image
    sec = out["config"]["secondary"]
    assert sec["config5_images"] == 256 and sec["config5_pairs"] == 255 and sec["config5_pairs_per_rank"] == [128, 127]
    assert sec["config5_images_per_rank"] == [129, 128] and sec["rccl_ranks"] == 2 and sec["config5_scaling"] == "strong"
    assert out["scaling"] == "weak"                                     # the headline stays config 2, weak-scaled


def test_gpus_3_c5_partition_is_the_halo_split():
    out, _ = _run("--gpus", "3", "--dry-run-dist", "--workload", "c5", "--images", "20")
    assert out["n_gpus"] == 3 and out["scaling"] == "strong"
    part = out["config"]["partition"]
    assert [p["pairs"] for p in part] == [7, 6, 6]                      # 19 sequential pairs (sfm.py:347), blocks differ by <= 1
    assert all(p["images_held"] == p["pairs"] + 1 for p in part)        # own images + ONE halo image


def test_gpus_1_does_not_spawn():
    out, err = _run("--gpus", "1", "--dry-run-dist")
    assert out["n_gpus"] == 1 and "launching" not in err


def test_refuses_more_ranks_than_gpus():
    """Without --dry-run-dist the launcher counts the node's GPUs first: none here, so --gpus 2 must fail loudly, not fall
    back to one rank."""
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("node has >= 2 GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "refusing" in r.stderr and r.stdout.strip() == ""
