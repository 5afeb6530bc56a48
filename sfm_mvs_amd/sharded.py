"""Pair-sharded feature matching across the GPUs of one node (BASELINE config 5).

The hot path shards by image pair: rank r owns a contiguous block of the pair list (sequential pairs
(k, k+1) as in sfm.py:347, or any list such as isfm.py's all-pairs) and matches it with NO data-path
collective.  The path's one exchange step is an all-gather of fixed-stride match records
{queryIdx, trainIdx, dist1, dist2} (16 B) + a per-pair survivor count, after which every rank holds
every pair's matches (camera registration is sequential and replicated).  One process per GPU,
`torch.distributed` with backend "nccl" (= RCCL over xGMI); the same code runs under "gloo" on CPU
tensors in the tests, with an injected matcher.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_range(n_items, world, rank):
    """Contiguous block [lo, hi) of rank `rank`; blocks differ by at most one item."""
    per, extra = divmod(n_items, world)
    lo = rank * per + min(rank, extra)
    return lo, lo + per + (1 if rank < extra else 0)


def hip_matcher(ratio=0.70):
    """Default matcher: KNN + ratio on the local GPU → (q_idx, t_idx, d1, d2) device tensors of the survivors."""
    from . import ops

    def match(des0, des1):
        idx, d = ops.knn2(des0, des1)
        out_q, out_t, count = ops.ratio_compact(idx, d, ratio)
        m = int(count.item())
        q = out_q[:m].long()
        return out_q[:m], out_t[:m], d[q, 0], d[q, 1]

    return match


def match_pairs_sharded(descriptors, pairs, matcher=None, group=None, device=None):
    """descriptors: list of [n_i,128] float32 tensors (all images resident, or None for images this rank
    never touches); pairs: list of (i, j).  Returns, on EVERY rank, a list with one entry per pair:
    dict(q=int32[m], t=int32[m], d1=float32[m], d2=float32[m])."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    matcher = matcher or hip_matcher()
    lo, hi = shard_range(len(pairs), world, rank)
    cap = max((descriptors[i].shape[0] for i, _ in pairs if descriptors[i] is not None), default=0)
    if world > 1:   # the record stride must agree on all ranks
        capt = torch.tensor([cap], dtype=torch.int64, device=device)
        dist.all_reduce(capt, op=dist.ReduceOp.MAX, group=group)
        cap = int(capt.item())
    slots = -(-len(pairs) // world)                       # pairs per rank, padded
    dev = device if device is not None else (descriptors[pairs[lo][0]].device if hi > lo else torch.device("cpu"))
    rec = torch.zeros((slots, cap, 4), dtype=torch.int32, device=dev)
    cnt = torch.zeros(slots, dtype=torch.int32, device=dev)
    for s, p in enumerate(range(lo, hi)):
        i, j = pairs[p]
        q, t, d1, d2 = matcher(descriptors[i], descriptors[j])
        m = q.shape[0]
        rec[s, :m, 0] = q.to(torch.int32)
        rec[s, :m, 1] = t.to(torch.int32)
        rec[s, :m, 2] = d1.to(torch.float32).view(torch.int32)
        rec[s, :m, 3] = d2.to(torch.float32).view(torch.int32)
        cnt[s] = m
    if world > 1:
        all_rec = torch.empty((world * slots, cap, 4), dtype=torch.int32, device=dev)
        all_cnt = torch.empty(world * slots, dtype=torch.int32, device=dev)
        dist.all_gather_into_tensor(all_rec, rec, group=group)
        dist.all_gather_into_tensor(all_cnt, cnt, group=group)
    else:
        all_rec, all_cnt = rec, cnt
    counts = all_cnt.cpu().numpy()
    out = []
    for p in range(len(pairs)):
        r = next(r for r in range(world) if shard_range(len(pairs), world, r)[0] <= p < shard_range(len(pairs), world, r)[1])
        s = r * slots + (p - shard_range(len(pairs), world, r)[0])
        m = int(counts[s])
        block = all_rec[s, :m]
        out.append(dict(q=block[:, 0], t=block[:, 1], d1=block[:, 2].view(torch.float32), d2=block[:, 3].view(torch.float32)))
    return out


def sequential_pairs(n_images):
    return [(k, k + 1) for k in range(n_images - 1)]


def all_pairs(n_images):
    """isfm.py:56-71: every (j, i) with j < i."""
    return [(j, i) for i in range(n_images) for j in range(i)]
