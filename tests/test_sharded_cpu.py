"""World-size-2 `gloo` tests of the pair-sharded path (the N > 1 code of bench.py / BASELINE config 5) on CPU tensors
with the oracle injected as the per-pair engines: halo partition (a rank holds only its block + one image), batched
exchange with a partial last batch and uneven pair counts, the all-gather of triangulated points, and the
BatchedExchange protocol bench.py drives."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleEngine:
    """CPU stand-in for sharded.HipMatchEngine: the oracle's KNN written into the exchange slot."""

    def __init__(self, oracle):
        self.O, self.streams, self.calls = oracle, (), 0

    def match(self, des0, des1, block, after=None):
        idx, d = self.O.knn2(des0.numpy(), des1.numpy())
        nq = len(idx)
        block[0, :nq] = torch.from_numpy(idx)
        block[1, :nq] = torch.from_numpy(d.view(np.int32))
        self.calls += 1


class OracleTriEngine:
    """CPU stand-in for sharded.HipTriangulateEngine: the oracle's Lowe loop + DLT written into the point slots."""

    streams = ()

    def __init__(self, oracle, ratio=0.70):
        self.O, self.ratio = oracle, ratio

    def triangulate_batch(self, items, after=()):
        for block, nq, kp0, kp1, P0, P1, pts, count in items:
            q, t = host_survivors(self.O, block, nq, self.ratio)
            pts.zero_()
            if len(q):
                pts[:, :len(q)] = torch.from_numpy(self.O.triangulate(P0, P1, kp0.numpy()[q].T.copy(), kp1.numpy()[t].T.copy(), normalise_w=True))
            count[0] = len(q)


class OracleVerifyEngine:
    """CPU stand-in for sharded.HipVerifyEngine: isfm.py:73-94 by the oracle."""

    def __init__(self, oracle, K, ratio=0.70):
        self.O, self.K, self.ratio = oracle, K, ratio

    def verify(self, block, nq, kp0, kp1):
        q, t = host_survivors(self.O, block, nq, self.ratio)
        if len(q) < 5:
            return -1
        a, b = kp0.numpy()[q], kp1.numpy()[t]
        E, m = self.O.find_essential_mat(a, b, self.K, 0.999, 0.4)
        if E is None:
            return -1
        keep = m.ravel() == 1
        return int((self.O.recover_pose(E, a[keep], b[keep], self.K)[3].ravel() > 0).sum())


def host_survivors(O, block, nq, ratio=0.70):
    """sfm.py:262-265 on a gathered KNN block, by the oracle."""
    q, t, _ = O.ratio_filter(block[0, :nq].numpy(), block[1, :nq].numpy().view(np.float32), ratio)
    return q, t


def host_merge_top2(gathered):
    """sfm_knn_merge_top2 on host tensors: order by (distance, global trainIdx) with two stable sorts."""
    world, _, nq, _ = gathered.shape
    cand_i = gathered[:, 0].permute(1, 0, 2).reshape(nq, 2 * world)
    cand_d = gathered[:, 1].permute(1, 0, 2).reshape(nq, 2 * world).contiguous().view(torch.float32)
    cand_d = torch.where(cand_i >= 0, cand_d, torch.full_like(cand_d, float("inf")))
    key_i = torch.where(cand_i >= 0, cand_i, torch.full_like(cand_i, 2 ** 31 - 1))
    o1 = torch.sort(key_i, dim=1, stable=True).indices
    o2 = torch.sort(torch.gather(cand_d, 1, o1), dim=1, stable=True).indices
    order = torch.gather(o1, 1, o2)[:, :2]
    out_i = torch.gather(cand_i, 1, order)
    out_d = torch.gather(cand_d, 1, order)
    return out_i.contiguous(), torch.where(out_i >= 0, out_d, torch.zeros_like(out_d)).contiguous()


def _scene(n_images, seed):
    from datagen import planted_pair
    rng = np.random.default_rng(seed)          # same data on every rank
    des = [torch.from_numpy(planted_pair(rng, 90 + 7 * (k % 5), 10, 0.0)[0]) for k in range(n_images)]
    for k in range(n_images - 1):               # plant matches between consecutive images
        n = min(len(des[k]), len(des[k + 1])) // 2
        des[k + 1][:n] = des[k][torch.randperm(len(des[k]), generator=torch.Generator().manual_seed(k))[:n]]
    kps = [torch.from_numpy(rng.uniform(0, 900, (len(d), 2)).astype(np.float32)) for d in des]
    return des, kps


def _worker(rank, world, port, n_images, batch, ret, all_pairs=False, cyclic=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from datagen import load_pose_csv
    from oracle import oracle as O
    from sfm_mvs_amd import sharded

    des, kps = _scene(n_images, 0)
    n_desc = [len(d) for d in des]
    pairs = sharded.all_pairs(n_images) if all_pairs else sharded.sequential_pairs(n_images)    # isfm.py:56-71 / sfm.py:347
    part = sharded.block_cyclic_partition(pairs, n_images, world, block=2) if cyclic else None
    mine = sharded.halo_images(pairs, world, rank, part)
    lo, hi = (0, len(part[rank])) if cyclic else sharded.shard_range(len(pairs), world, rank)
    held = [d if i in mine else None for i, d in enumerate(des)]        # the halo partition: nothing else is resident
    held_kp = [k if i in mine else None for i, k in enumerate(kps)]
    eng = OracleEngine(O)
    store, nq = sharded.match_pairs_sharded(held, pairs, n_desc=n_desc, engine=eng, device=torch.device("cpu"), batch=batch, partition=part)
    ok = eng.calls == hi - lo and (all_pairs or len(mine) == (hi - lo + 1 if hi > lo else 0))
    total = 0
    for p, (i, j) in enumerate(pairs):                                  # every pair's block, on every rank, = the oracle's
        wi, wd = O.knn2(des[i].numpy(), des[j].numpy())
        ok = ok and np.array_equal(store[p, 0, :nq[p]].numpy(), wi) and np.array_equal(store[p, 1, :nq[p]].numpy().view(np.float32), wd)
        q, t = host_survivors(O, store[p], nq[p])
        total += len(q)
    # second exchange: triangulated points of the survivors (cameras replicated)
    K, P = load_pose_csv()

    pts, counts = sharded.triangulate_pairs_sharded(store, nq, pairs, held_kp, list(P[:n_images]), engine=OracleTriEngine(O), batch=batch, partition=part)
    for p, (i, j) in enumerate(pairs):
        q, t = host_survivors(O, store[p], nq[p])
        m = int(counts[p])
        ok = ok and m == len(q)
        if m:
            want = O.triangulate(P[i], P[j], kps[i].numpy()[q].T.copy(), kps[j].numpy()[t].T.copy(), normalise_w=True)
            ok = ok and np.array_equal(pts[p, :, :m].numpy(), want) and float(pts[p, :, m:].abs().sum()) == 0.0
    if all_pairs:
        # isfm.py:80-94 on every pair, sharded: each rank verifies its own pairs, one all-gather of the counts
        eng_v = OracleVerifyEngine(O, K)
        got = sharded.verify_pairs_sharded(store, nq, pairs, held_kp, K, engine=eng_v, partition=part)
        only = {0, len(pairs) - 1}
        part_got = sharded.verify_pairs_sharded(store, nq, pairs, held_kp, K, engine=eng_v, partition=part, only=only)
        for p, (i, j) in enumerate(pairs):
            want = eng_v.verify(store[p], nq[p], kps[i], kps[j])
            ok = ok and int(got[p]) == want and int(part_got[p]) == (want if p in only else -2)
    ret[rank] = (bool(ok), hi - lo, total)
    dist.destroy_process_group()


@pytest.mark.parametrize("n_images,batch,port", [(6, 8, 29517), (12, 2, 29518), (8, 3, 29519), (2, 4, 29520)])
def test_two_rank_pair_sharding(n_images, batch, port):
    """(6, 8): one partial batch; (12, 2): 11 pairs -> 6 + 5 (uneven), several rounds; (8, 3): partial last round;
    (2, 4): one pair — rank 1 owns nothing and still takes part in every collective."""
    world = 2
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(world, port, n_images, batch, ret), nprocs=world, join=True)
    assert ret[0][0] and ret[1][0]
    assert ret[0][1] + ret[1][1] == n_images - 1 and abs(ret[0][1] - ret[1][1]) <= 1
    assert ret[0][2] == ret[1][2] > 20


def test_two_rank_all_pairs():
    """isfm.py's exhaustive pair list (every j < i) through the same sharded matcher and point gather: 5 images -> 10
    pairs, 5 per rank; a rank holds the images its block of pairs names, nothing else."""
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(2, 29523, 5, 4, ret, True), nprocs=2, join=True)
    assert ret[0][0] and ret[1][0] and ret[0][1] + ret[1][1] == 10 and ret[0][2] == ret[1][2]


def test_two_rank_all_pairs_block_cyclic():
    """The exhaustive pair list under SURVEY 8e's 2-D block-cyclic split (2 x 2-image tiles dealt to a 1 x 2 process grid):
    7 images -> 21 pairs; every pair's block and points still arrive on every rank, a rank holds only the images its
    tiles name."""
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(2, 29525, 7, 4, ret, True, True), nprocs=2, join=True)
    assert ret[0][0] and ret[1][0] and ret[0][1] + ret[1][1] == 21 and ret[0][2] == ret[1][2]


def test_three_rank_all_pairs_block_cyclic_uneven_grid():
    """Three ranks (a 1 x 3 process grid: the tiles of the pair triangle do not divide evenly), 7 images -> 21 pairs dealt
    9 / 6 / 6 or similar: blocks, points and the isfm.py:80-94 inlier counts arrive on every rank."""
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(3, 29527, 7, 4, ret, True, True), nprocs=3, join=True)
    assert all(ret[r][0] for r in range(3)) and sum(ret[r][1] for r in range(3)) == 21 and len({ret[r][2] for r in range(3)}) == 1
    assert len({ret[r][1] for r in range(3)}) > 1          # uneven loads


def test_block_cyclic_partition_covers_all_pairs_with_small_halos():
    """isfm.py:56-71 at config-5 scale: 256 images -> 32 640 pairs over 8 ranks (2 x 4 grid, 16-image tiles).  Every pair
    is owned exactly once, the loads are balanced, and a rank touches the image blocks of one grid row and one grid
    column: at most n / pr + n / pc = 192 images — a contiguous split of the pair list needs all 256 on its last rank (the
    resident set is sized by the worst rank)."""
    from sfm_mvs_amd.sharded import all_pairs, block_cyclic_partition, contiguous_partition, halo_images, process_grid
    assert process_grid(8) == (2, 4) and process_grid(4) == (2, 2) and process_grid(2) == (1, 2) and process_grid(1) == (1, 1)
    n, world = 256, 8
    pairs = all_pairs(n)
    part = block_cyclic_partition(pairs, n, world)
    allp = np.sort(np.concatenate(part))
    assert np.array_equal(allp, np.arange(len(pairs)))
    sizes = [len(x) for x in part]
    assert max(sizes) <= 1.15 * min(sizes), sizes
    halos = [len(halo_images(pairs, world, r, part)) for r in range(world)]
    assert max(halos) <= n // 2 + n // 4, halos
    cont = [len(halo_images(pairs, world, r, contiguous_partition(len(pairs), world))) for r in range(world)]
    assert max(cont) == n and max(halos) <= 0.75 * n, (cont, halos)
    for w in (1, 2, 3, 4, 6):                                # any world size: a partition, nothing lost
        pt = block_cyclic_partition(all_pairs(37), 37, w)
        assert np.array_equal(np.sort(np.concatenate(pt)), np.arange(37 * 36 // 2))


def _train_split_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from datagen import sift_like
    from oracle import oracle as O
    from sfm_mvs_amd import sharded
    rng = np.random.default_rng(5)             # same data on every rank
    ok = True
    for nq, nt, cut in ((200, 301, 120), (50, 3, 1), (40, 2, 2), (30, 1, 1)):
        q, t = sift_like(rng, nq), sift_like(rng, nt)
        if nt > 200:
            t[250] = t[7]                      # exact duplicates across the two shards: the lower global index must win
            t[130] = t[7]
            q[0] = t[7]
        wi, wd = O.knn2(q, t)
        lo, hi = (0, cut) if rank == 0 else (cut, nt)
        knn = lambda a, b: tuple(torch.from_numpy(x) for x in O.knn2(a.numpy(), b.numpy()))
        gi, gd = sharded.knn2_train_split(torch.from_numpy(q), torch.from_numpy(t[lo:hi]), lo, knn2=knn, merge=host_merge_top2)
        valid = wi >= 0
        ok = ok and np.array_equal(gi.numpy(), wi) and np.array_equal(gd.numpy()[valid], wd[valid])
    ret[rank] = bool(ok)
    dist.destroy_process_group()


def test_train_dimension_split_merges_partial_top2():
    """the train rows of one pair split over two ranks, partial top-2 all-gathered and merged 4 -> 2 by
    (distance, global index): equal to the single scan, ties and tiny shards (fewer than two rows on a rank) included."""
    ret = mp.Manager().dict()
    mp.spawn(_train_split_worker, args=(2, 29524, ret), nprocs=2, join=True)
    assert ret[0] and ret[1]


def _exchange_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sfm_mvs_amd import sharded
    ex = sharded.BatchedExchange((3, 2), torch.int32, torch.device("cpu"), batch=4)
    seen, steps = [], 10                       # bench.py's protocol: a slot per step, flush when full, final partial flush
    for s in range(steps):
        slot, ev = ex.next_slot()
        slot.fill_(1000 * rank + s)
        if ex.commit():
            g, filled = ex.flush()
            seen.append((g.clone(), filled))
    g, filled = ex.flush()
    seen.append((g.clone(), filled))
    ok = ex.collectives == 3 and [f for _, f in seen] == [4, 4, 2]
    for b, (g, f) in enumerate(seen):
        for r in range(world):
            for k in range(f):
                ok = ok and bool((g[r, k] == 1000 * r + 4 * b + k).all())
    ret[rank] = bool(ok)
    dist.destroy_process_group()


def test_batched_exchange_protocol():
    ret = mp.Manager().dict()
    mp.spawn(_exchange_worker, args=(2, 29521, ret), nprocs=2, join=True)
    assert ret[0] and ret[1]


def test_shard_range_and_halo_partitions():
    from sfm_mvs_amd.sharded import all_pairs, halo_images, sequential_pairs, shard_range
    for n in (0, 1, 5, 255, 256):
        for w in (1, 2, 4, 8):
            spans = [shard_range(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
    assert len(all_pairs(10)) == 45
    pairs = sequential_pairs(256)               # config 5: 255 pairs over 8 ranks -> 31 or 32 pairs, + 1 halo image each
    for r in range(8):
        lo, hi = shard_range(255, 8, r)
        assert halo_images(pairs, 8, r) == list(range(lo, hi + 1))
    assert sum(len(halo_images(pairs, 8, r)) for r in range(8)) == 256 + 7
