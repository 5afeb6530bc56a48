"""Pair-sharded feature matching and triangulation across the GPUs of one node (BASELINE config 5).

The hot path shards by image pair: rank r owns a contiguous block of the pair list — sequential pairs (k, k+1) as in
sfm.py:347, or any list such as isfm.py's all-pairs — and matches it with NO data-path collective.  For sequential
pairs a rank needs only its block's images plus ONE halo image (the train image of its last pair): `halo_images`.
The path has two exchange steps, both all-gathers of fixed-stride blocks (one process per GPU, `torch.distributed`
backend "nccl" = RCCL over xGMI; the same code runs under "gloo" on CPU tensors in the tests):

  1. match records: per pair the KNN block {trainIdx x2, distance x2} of every query (16 B per query, = the DMatch
     pairs of sfm.py:260; the Lowe survivors of sfm.py:262-265 are a function of it), and
  2. triangulated points: per pair float32 x 4 per surviving match (sfm.py:349,371 run on the owning rank).

Blocks are exchanged `batch` pairs at a time: a 160 KB all-gather per 10k-query pair costs more in launch and ring
latency than the pair itself, and xGMI is point-to-point (a ring is bound by one link), so fewer, larger collectives.
Batch buffers alternate (double buffering); every rank issues the same collectives in the same order whatever its own
pair count (uneven blocks and a partial last batch send unused slots).  A rank then has more streams in flight than the
HIP runtime's default four hardware queues (matching streams + the collective's stream + RCCL's own): export
GPU_MAX_HW_QUEUES=8 before the process touches the GPU, or streams that share a queue serialise (measured: -14 %).  Camera registration (the PnP chain) is
sequential and stays replicated.  `bench.py --gpus N` drives the same `BatchedExchange`.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_range(n_items, world, rank):
    """Contiguous block [lo, hi) of rank `rank`; blocks differ by at most one item."""
    per, extra = divmod(n_items, world)
    lo = rank * per + min(rank, extra)
    return lo, lo + per + (1 if rank < extra else 0)


def sequential_pairs(n_images):
    return [(k, k + 1) for k in range(n_images - 1)]


def all_pairs(n_images):
    """isfm.py:56-71: every (j, i) with j < i."""
    return [(j, i) for i in range(n_images) for j in range(i)]


def contiguous_partition(n_pairs, world):
    """The default split: rank r owns the contiguous block shard_range(n_pairs, world, r) of the pair list."""
    return [np.arange(*shard_range(n_pairs, world, r)) for r in range(world)]


def process_grid(world):
    """pr x pc with pr the largest divisor of `world` not above its square root (8 -> 2 x 4)."""
    pr = max(d for d in range(1, int(world ** 0.5) + 1) if world % d == 0)
    return pr, world // pr


def block_cyclic_partition(pairs, n_images, world, block=None):
    """SURVEY 8e's split for EXHAUSTIVE matching (isfm.py:56-71: every j < i): the (j, i) pair grid is cut into
    block x block tiles of images and tile (bj, bi) goes to rank (bj mod pr) * pc + (bi mod pc) of a pr x pc process grid.
    A rank then touches only the image blocks of its grid row and grid column — about n / pr + n / pc images instead of
    nearly all n with a contiguous split of the pair list (the halo does not depend on the tile size) — and the cyclic deal
    balances the triangular pair region: the heaviest rank carries 1 + O(pc block / n) of the mean (256 images over 2 x 4:
    1.67 x the lightest at block 16, 1.13 x at 4, 1.06 x at 2); block defaults to n / (32 pc), at least 1.  Returns the list over ranks of pair-index
    arrays (indices into `pairs`, ascending), usable as `partition=` below."""
    pr, pc = process_grid(world)
    if block is None:
        block = max(1, n_images // (32 * pc))
    owner = [[] for _ in range(world)]
    for p, (j, i) in enumerate(pairs):
        owner[((j // block) % pr) * pc + ((i // block) % pc)].append(p)
    return [np.asarray(o, dtype=np.int64) for o in owner]


def halo_images(pairs, world, rank, partition=None):
    """Images rank `rank` must hold to match its pairs: for sequential pairs (contiguous split) its own images + one halo
    image; for a block-cyclic split of all pairs the image blocks of its process-grid row and column."""
    mine = (partition or contiguous_partition(len(pairs), world))[rank]
    return sorted({i for p in mine for i in pairs[int(p)]})


def _round_targets(partition, rounds, batch, dump, device):
    """Destination pair index of every (round, rank, slot) (unused slots go to the dump row), as ONE device tensor
    [rounds][world * batch]: a per-round upload would put a host-to-device copy between every two collectives."""
    dst = np.full((max(rounds, 1), len(partition), batch), dump, dtype=np.int64)
    for r, part in enumerate(partition):
        for rd in range(rounds):
            k = np.asarray(part[rd * batch:(rd + 1) * batch], dtype=np.int64)
            dst[rd, r, :len(k)] = k
    return torch.from_numpy(dst.reshape(max(rounds, 1), -1)).to(device)


def _world_rank(group):
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0


class BatchedExchange:
    """All-gather of fixed-stride blocks, `batch` blocks per collective, two alternating batch buffers.

    A producer asks for the next slot (`next_slot` -> tensor view + the event after which the slot may be overwritten),
    enqueues the work that fills it on any stream, then `commit`s; when the batch is full — or on `flush` — the buffer is
    all-gathered from the CURRENT stream after waiting for `producer_streams`.  `flush` returns the gathered tensor
    [world][batch][*block_shape] (a view of an internal buffer, valid until the second next flush) and how many slots
    this rank had filled; ranks may fill different numbers of slots, but must call flush the same number of times."""

    def __init__(self, block_shape, dtype, device, batch=8, group=None, nbuf=2):
        self.world, self.rank = _world_rank(group)
        self.group, self.batch, self.device = group, int(batch), torch.device(device)
        self.block_shape = tuple(int(x) for x in block_shape)
        self.local = [torch.zeros((self.batch,) + self.block_shape, dtype=dtype, device=self.device) for _ in range(nbuf)]
        self.gathered = [torch.empty((self.world, self.batch) + self.block_shape, dtype=dtype, device=self.device) for _ in range(nbuf)]
        self.free = [None] * nbuf                       # per buffer: event "its previous all-gather has read it"
        self.cur, self.fill, self.collectives = 0, 0, 0
        self.cuda = self.device.type == "cuda"
        # a one-rank process group still goes through the collective (the same RCCL / gloo call as at N > 1)
        self.collective = dist.is_available() and dist.is_initialized()
        self._timed = []                                # (start, end) events of the collectives, for exchange_ms()

    def next_slot(self):
        """(slot view, event to wait for before writing it or None).  Only the first `depth` producers of a batch need
        the event (later ones are ordered behind them on their own streams); handing it to all is harmless."""
        if self.fill >= self.batch:
            raise RuntimeError("BatchedExchange: batch is full — flush first")
        return self.local[self.cur][self.fill], self.free[self.cur]

    def commit(self):
        """The slot returned by the last next_slot() has its producer enqueued.  True when the batch is now full."""
        self.fill += 1
        return self.fill == self.batch

    def flush(self, producer_streams=()):
        """Exchange the current batch buffer now (a partial batch is sent whole).  Every rank must call this the same
        number of times.  Returns (gathered [world][batch][...], slots filled by this rank)."""
        cur, filled = self.cur, self.fill
        if self.cuda:
            main = torch.cuda.current_stream(self.device)
            for st in producer_streams:
                main.wait_stream(st)
            ev0 = torch.cuda.Event(enable_timing=True)
            ev0.record(main)
        if self.collective:
            # (concatenated form [world * batch][...]: gloo accepts no other; nccl takes both)
            dist.all_gather_into_tensor(self.gathered[cur].view((self.world * self.batch,) + self.block_shape), self.local[cur], group=self.group)
        else:
            self.gathered[cur][0].copy_(self.local[cur])
        if self.cuda:
            ev1 = torch.cuda.Event(enable_timing=True)
            ev1.record(main)                            # (the collective is complete on `main` here: async_op=False)
            self.free[cur] = ev1
            self._timed.append((ev0, ev1))
        self.collectives += 1
        self.cur, self.fill = (cur + 1) % len(self.local), 0
        return self.gathered[cur], filled

    def exchange_ms(self, reset=True):
        """Device time spent in the collectives since the last reset (synchronises their events)."""
        ms = 0.0
        for a, b in self._timed:
            b.synchronize()
            ms += a.elapsed_time(b)
        if reset:
            self._timed = []
        return ms


class HipMatchEngine:
    """KNN + ratio of image pairs into caller-provided KNN blocks: pairs of one shape are issued up to 8 per launch set
    (ops.BatchPipeline -> sfm_match_batch_l2_f32: one prep / filter / refine / scatter launch for the batch), launch sets
    pipelined over `depth` HIP streams.  `flush()` launches what is queued (the exchange calls it before a collective)."""

    def __init__(self, device, ratio=0.70, depth=3, batch=8, max_shapes=4):
        self.device, self.ratio, self.depth, self.batch = torch.device(device), ratio, depth, batch
        self.pipes = {}                                    # (nq, nt) -> BatchPipeline, most recently used last
        self._streams = None                               # ONE probed stream set for all of them (ops.independent_streams), made at first use
        self._padded = {}                                  # per pipe: blocks of queued pairs that are wider than nq
        # A pipeline owns depth x (workspace + outputs) for `batch` pairs of its shape (2 GB per launch set at 50k x 50k): a
        # sequence of real images has a different descriptor count per image, so only the `max_shapes` most recently used shapes
        # keep theirs (an evicted one is flushed and drained first).
        self.max_shapes = max(1, int(max_shapes))

    @property
    def streams(self):
        return list(self._streams or ())

    def match(self, des0, des1, block, after=None):
        """block: int32 [2][cap][2] view (cap >= nq) receiving (trainIdx x2, distance bits x2) of the nq queries.
        after = None (no free-event yet: the first rounds): the pair waits for everything already enqueued on the caller's
        current stream — descriptors still being produced there (an asynchronous SIFT or upload) are safe to pass in."""
        from . import ops
        nq, nt = des0.shape[0], des1.shape[0]
        # the pairs of one launch set share their row strides (sfm_match_batch_l2_f32 takes one ldq / ldt): a sliced view and a
        # contiguous tensor of the same shape go to different pipelines instead of failing mid-round; rows must be unit-stride
        if des0.dim() == 2 and des0.stride(1) != 1:
            des0 = des0.contiguous()
        if des1.dim() == 2 and des1.stride(1) != 1:
            des1 = des1.contiguous()
        key = (nq, nt, des0.stride(0) if nq else 128, des1.stride(0) if nt else 128)
        pipe = self.pipes.pop(key, None)
        if pipe is None:
            while len(self.pipes) >= self.max_shapes:       # evict the least recently used shape
                old = next(iter(self.pipes))
                self._launch(old, self.pipes[old].flush())
                self.pipes[old].synchronize()
                del self.pipes[old], self._padded[old]
            if self._streams is None:
                self._streams = ops._pipeline_streams(self.depth, self.device)
            pipe = ops.BatchPipeline(nq, nt, self.device, ratio=self.ratio, depth=self.depth, batch=self.batch, streams=self._streams)
            self._padded[key] = []
        self.pipes[key] = pipe                             # (re-inserted: most recently used)
        direct = block.shape[1] == nq and block.is_contiguous()
        if pipe.pending and direct != (not self._padded[key]):     # a launch set writes either into blocks or into its own result
            self._launch(key, pipe.flush())
        rec = pipe.submit(des0, des1, after=after, result=block if direct else None)     # (may raise: nothing is queued then)
        if not direct:
            self._padded[key].append(block)                # only once the pair IS queued: the list stays in step with the pipeline
        self._launch(key, rec)

    def _launch(self, key, rec):
        """rec: BatchPipeline's launch record (or None: the batch is still filling).  Padded blocks get their rows copied
        behind the launch set, on its stream."""
        if rec is None or not self._padded[key]:
            return
        _, st, bm, n = rec
        nq = key[0]
        with torch.cuda.stream(st):
            for b, block in enumerate(self._padded[key][:n]):
                block[:, :nq].copy_(bm.result[b], non_blocking=True)
        self._padded[key] = self._padded[key][n:]

    def flush(self):
        for key, pipe in self.pipes.items():
            self._launch(key, pipe.flush())


def ratio_survivors(block, nq, ratio=0.70):
    """The Lowe loop of sfm.py:262-265 on a gathered KNN block (int32 [2][cap][2] device tensor) -> (queryIdx, trainIdx)
    int32 device tensors in ascending queryIdx order: `sfm_ratio_compact` (float32 distances promoted to double, strict <, as
    Python compares them); one host read of the survivor count."""
    from . import ops
    out_q, out_t, count = ops.ratio_compact(block[0, :nq], block[1, :nq].view(torch.float32), ratio)
    m = int(count.item())
    return out_q[:m], out_t[:m]


class HipTriangulateEngine:
    """Lowe survivors -> keypoint gather -> guarded DLT of up to 8 pairs per call, straight from the gathered KNN blocks into
    the slots of the point exchange (ops.triangulate_matches_batch): nothing goes through the host."""

    streams = ()

    def __init__(self, device, ratio=0.70):
        self.device, self.ratio = torch.device(device), ratio

    def triangulate_batch(self, items, after=()):
        """items: list of (knn_block int32 [2][cap][2], n_query, kp0 [n,2], kp1 [n,2], P0, P1, points float32 [4][cap] view,
        count int32 [1] view); after: events the slots wait for (their previous all-gather has read them)."""
        from . import ops
        cur = torch.cuda.current_stream(self.device)
        for ev in {id(e): e for e in after if e is not None}.values():
            cur.wait_event(ev)
        for lo in range(0, len(items), 8):
            part = items[lo:lo + 8]
            ops.triangulate_matches_batch([it[0] for it in part], [it[1] for it in part], [it[2] for it in part], [it[3] for it in part],
                                          [it[4] for it in part], [it[5] for it in part], [it[6] for it in part], [it[7] for it in part], self.ratio)


def match_pairs_sharded(descriptors, pairs, n_desc=None, engine=None, group=None, device=None, batch=8, ratio=0.70, partition=None, stats=None):
    """descriptors: list over images of [n_i,128] float32 tensors — None for images this rank never touches
    (`halo_images`); pairs: list of (i, j); n_desc: descriptor count of EVERY image (needed for images held elsewhere;
    taken from `descriptors` when all are present).  Every rank matches its block of pairs and the KNN blocks are
    all-gathered `batch` pairs at a time.  partition: list over ranks of pair-index arrays (default: contiguous blocks;
    `block_cyclic_partition` for exhaustive pair lists).  Returns, on every rank, (store, n_query): store int32 [n_pairs][2][cap][2]
    with store[p][0] = trainIdx x2 and store[p][1] = float32 distance bits x2 of pair p's n_query[p] queries."""
    world, rank = _world_rank(group)
    if n_desc is None:
        n_desc = [d.shape[0] for d in descriptors]
    n_pairs = len(pairs)
    dev = torch.device(device) if device is not None else next(d.device for d in descriptors if d is not None)
    cap = max((n_desc[i] for i, _ in pairs), default=0)
    engine = engine or HipMatchEngine(dev, ratio)
    ex = BatchedExchange((2, cap, 2), torch.int32, dev, batch, group)
    partition = partition or contiguous_partition(n_pairs, world)
    mine = partition[rank]
    per = max((len(part) for part in partition), default=0)
    store = torch.zeros((n_pairs + 1, 2, cap, 2), dtype=torch.int32, device=dev)      # [+1]: dump row for unused slots
    rounds = -(-per // batch) if per else 0
    dst = _round_targets(partition, rounds, batch, n_pairs, dev)
    for rd in range(rounds):
        for p in mine[rd * batch:(rd + 1) * batch]:
            i, j = pairs[int(p)]
            slot, ev = ex.next_slot()
            engine.match(descriptors[i], descriptors[j], slot, after=ev)
            ex.commit()
        getattr(engine, "flush", lambda: None)()           # (engines that batch their launches: issue what is queued)
        gathered, _ = ex.flush(getattr(engine, "streams", ()))
        # scatter the round's blocks to their pairs in ONE indexed copy (unused slots go to the dump row)
        store.index_copy_(0, dst[rd], gathered.reshape((world * batch,) + gathered.shape[2:]))
    if stats is not None:                                  # (bench.py: device time inside the collectives, their number and size)
        stats.update(exchange_ms=ex.exchange_ms() if ex.cuda else 0.0, collectives=ex.collectives,
                     bytes_per_rank_per_collective=batch * 2 * cap * 2 * 4)
    return store[:n_pairs], [n_desc[i] for i, _ in pairs]


def triangulate_pairs_sharded(store, n_query, pairs, keypoints, proj, engine=None, group=None, batch=8, ratio=0.70, partition=None, stats=None):
    """The path's second exchange (north_star: "all-gather of 3D points"): every rank triangulates the Lowe survivors of
    ITS pairs (sfm.py:262-268 then :349,371: ratio loop, keypoint gather, cv2.triangulatePoints + division by w) and the
    float32 x 4 points are all-gathered together with each pair's survivor count.
    keypoints: list over images of contiguous [n_i,2] float32 device tensors (None where not held: a rank needs its block +
    halo, as for the descriptors); proj: list over images of 3x4 float64 projection matrices (replicated: the PnP chain is
    sequential).  engine: `triangulate_batch(items, after)` as HipTriangulateEngine (the default: one fused set of launches
    per round, no host synchronisation; the gloo tests inject a host engine).
    Returns (points float32 [n_pairs][4][cap], counts int64 [n_pairs] on the host) on every rank."""
    world, rank = _world_rank(group)
    dev = store.device
    n_pairs, cap = len(pairs), store.shape[2]
    engine = engine or HipTriangulateEngine(dev, ratio)
    width = 4 * cap + 4                                      # [4][cap] points, then the survivor count (int32 bits) + padding
    ex = BatchedExchange((width,), torch.float32, dev, batch, group)
    partition = partition or contiguous_partition(n_pairs, world)
    mine = partition[rank]
    per = max((len(part) for part in partition), default=0)
    points = torch.zeros((n_pairs + 1, width), dtype=torch.float32, device=dev)
    rounds = -(-per // batch) if per else 0
    dst = _round_targets(partition, rounds, batch, n_pairs, dev)
    for rd in range(rounds):
        items, after = [], []
        for p in mine[rd * batch:(rd + 1) * batch]:
            p = int(p)
            i, j = pairs[p]
            slot, ev = ex.next_slot()
            items.append((store[p], int(n_query[p]), keypoints[i], keypoints[j], proj[i], proj[j],
                          slot[:4 * cap].view(4, cap), slot[4 * cap:4 * cap + 1].view(torch.int32)))
            after.append(ev)
            ex.commit()
        if items:
            engine.triangulate_batch(items, after)
        gathered, _ = ex.flush(getattr(engine, "streams", ()))
        points.index_copy_(0, dst[rd], gathered.reshape((world * batch, width)))
    counts = points[:n_pairs, 4 * cap].contiguous().view(torch.int32).cpu().long()      # the job's one host read
    if stats is not None:
        stats.update(exchange_ms=ex.exchange_ms() if ex.cuda else 0.0, collectives=ex.collectives, bytes_per_rank_per_collective=batch * width * 4)
    return points[:n_pairs, :4 * cap].view(n_pairs, 4, cap), counts


class HipVerifyEngine:
    """isfm.py:73-94 for one matched pair on the device: Lowe survivors -> keypoint gather -> cv2.findEssentialMat(RANSAC,
    prob 0.999, threshold 0.4) -> rows with mask == 1 -> cv2.recoverPose -> rows with mask > 0; returns how many correspondences
    are left (the number isfm.py prints) or -1 when no essential matrix was found / fewer than five survivors."""

    def __init__(self, K, ratio=0.70):
        self.K, self.ratio = K, ratio

    def verify(self, block, nq, kp0, kp1):
        from . import ops, ransac
        out_q, out_t, count = ops.ratio_compact(block[0, :nq], block[1, :nq].view(torch.float32), self.ratio)
        pts0, pts1 = ops.gather_matches(kp0, kp1, out_q, out_t, count)
        m = int(count.item())
        if m < 5:
            return -1
        pts0, pts1 = pts0[:m], pts1[:m]
        E, mask = ransac.find_essential_mat(pts0, pts1, self.K, 0.999, 0.4, return_device_mask=True)
        if E is None:
            return -1
        keep = ops.mask_indices(mask).long()                   # isfm.py:81-82 (rows of OpenCV's {0,1} mask)
        pts0, pts1 = pts0[keep], pts1[keep]
        _, _, _, mask = ransac.recover_pose(E[:3], pts0, pts1, self.K, return_device_mask=True)
        return int(ops.mask_indices(mask, nonzero=True).numel())   # isfm.py:84-86 (rows of the {0,255} mask)


def verify_pairs_sharded(store, n_query, pairs, keypoints, K, engine=None, group=None, ratio=0.70, partition=None, only=None):
    """isfm.py:73-94 over a matched pair list: every rank verifies ITS pairs (essential-matrix RANSAC + cheirality vote on
    the Lowe survivors of the gathered KNN blocks), the per-pair inlier counts — one integer per pair — are exchanged in one
    all-gather.  only: optional set of pair indices to verify (others report -2).  engine: `verify(block, nq, kp0, kp1) ->
    int` as HipVerifyEngine (the default).  Returns int64 [n_pairs] on every rank."""
    world, rank = _world_rank(group)
    n_pairs = len(pairs)
    engine = engine or HipVerifyEngine(K, ratio)
    partition = partition or contiguous_partition(n_pairs, world)
    per = max((len(part) for part in partition), default=0)
    local = torch.full((max(per, 1),), -2, dtype=torch.int64)
    for k, p in enumerate(partition[rank]):
        p = int(p)
        if only is None or p in only:
            i, j = pairs[p]
            local[k] = engine.verify(store[p], int(n_query[p]), keypoints[i], keypoints[j])
    dev = store.device
    gathered = torch.empty((world, max(per, 1)), dtype=torch.int64, device=dev)
    if dist.is_available() and dist.is_initialized():
        dist.all_gather_into_tensor(gathered.view(-1), local.to(dev), group=group)
    else:
        gathered[0].copy_(local)
    gathered = gathered.cpu()
    out = torch.full((n_pairs,), -2, dtype=torch.int64)
    for r, part in enumerate(partition):
        if len(part):
            out[torch.from_numpy(np.asarray(part, dtype=np.int64))] = gathered[r, :len(part)]
    return out


def knn2_train_split(des0, des1_shard, train_offset, knn2=None, merge=None, group=None):
    """SURVEY 8e's fallback for ONE pair whose train set is split over the ranks (a descriptor set that does not fit one
    device, or a single huge pair to be sped up): every rank holds all queries and the train rows
    [train_offset, train_offset + len(des1_shard)), computes its partial top-2, the partial results are all-gathered and
    merged 2 * world -> 2 by (distance, global train index) — associative, and the lower index wins ties exactly as in a
    single scan (cv2.BFMatcher keeps the earlier row).  knn2(des0, des1) -> (idx [nq,2] int32, dist [nq,2] float32), -1 /
    anything where a neighbour is missing; merge(gathered int32 [world][2][nq][2]) -> (idx, dist); defaults: the HIP kernels
    (sfm_knn2_l2_f32, sfm_knn_merge_top2: one kernel, a lane per query) — the gloo tests inject host engines.
    Returns (idx, dist) on every rank."""
    world, rank = _world_rank(group)
    if knn2 is None or merge is None:
        from . import ops
        knn2, merge = knn2 or ops.knn2, merge or ops.knn_merge_top2
    nq, dev = des0.shape[0], des0.device
    local = torch.empty((2, nq, 2), dtype=torch.int32, device=dev)
    if des1_shard.shape[0] > 0:
        idx, d = knn2(des0, des1_shard)
        local[0] = torch.where(idx >= 0, idx + int(train_offset), idx)
        local[1] = d.contiguous().view(torch.int32)
    else:
        local[0].fill_(-1)
        local[1].zero_()
    gathered = torch.empty((world, 2, nq, 2), dtype=torch.int32, device=dev)
    if dist.is_available() and dist.is_initialized():
        dist.all_gather_into_tensor(gathered.view(world * 2, nq, 2), local, group=group)
    else:
        gathered[0].copy_(local)
    return merge(gathered)
