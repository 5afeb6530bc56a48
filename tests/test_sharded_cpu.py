"""World-size-2 `gloo` test of the pair-sharded matcher (the N>1 path of bench.py / config 5), on CPU
tensors with the oracle injected as the per-pair matcher."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from datagen import planted_pair
    from oracle import oracle as O
    from sfm_mvs_amd import sharded

    rng = np.random.default_rng(0)          # same data on every rank
    des = [torch.from_numpy(planted_pair(rng, 120 + 10 * k, 10, 0.0)[0]) for k in range(6)]
    for k in range(5):                       # plant matches between consecutive images
        n = min(len(des[k]), len(des[k + 1])) // 2
        des[k + 1][:n] = des[k][torch.randperm(len(des[k]), generator=torch.Generator().manual_seed(k))[:n]]

    def matcher(a, b):
        idx, d = O.knn2(a.numpy(), b.numpy())
        q, t, _ = O.ratio_filter(idx, d, 0.70)
        return torch.from_numpy(q), torch.from_numpy(t), torch.from_numpy(d[q, 0]), torch.from_numpy(d[q, 1])

    pairs = sharded.sequential_pairs(6)
    got = sharded.match_pairs_sharded(des, pairs, matcher=matcher, device=torch.device("cpu"))
    want = [matcher(des[i], des[j]) for i, j in pairs]
    ok = all(torch.equal(g["q"], w[0]) and torch.equal(g["t"], w[1]) and torch.equal(g["d1"], w[2]) and
             torch.equal(g["d2"], w[3]) for g, w in zip(got, want))
    lo, hi = sharded.shard_range(len(pairs), world, rank)
    ret[rank] = (ok, hi - lo, sum(len(g["q"]) for g in got))
    dist.destroy_process_group()


def test_two_rank_pair_sharding_gathers_every_pair_on_every_rank():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, 29517, ret), nprocs=world, join=True)
    assert ret[0][0] and ret[1][0]
    assert ret[0][1] + ret[1][1] == 5 and abs(ret[0][1] - ret[1][1]) <= 1
    assert ret[0][2] == ret[1][2] > 100


def test_shard_range_partitions():
    from sfm_mvs_amd.sharded import all_pairs, shard_range
    for n in (0, 1, 5, 255, 256):
        for w in (1, 2, 4, 8):
            spans = [shard_range(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
    assert len(all_pairs(10)) == 45
