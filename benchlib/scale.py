"""BASELINE configs[4] (`--workload c5`; part of the default run at N > 1), isfm.py's exhaustive loop (`--workload allpairs`) and the
dry run of the N-rank protocol over gloo (`--dry-run-dist`)."""
from .common import *  # noqa: F401,F403


def bench_c5(args, world, rank, dev):
    """BASELINE configs[4] as written: `--images` (256) images x 50 000 SIFT-like descriptors in TOTAL, sequential pairs
    (k, k+1) as in sfm.py:347 sharded over the ranks through the package's one multi-GPU code path —
    sharded.match_pairs_sharded (halo partition: a rank generates and holds only its block's images + one halo image;
    the KNN blocks of 8 pairs per RCCL all-gather, inside the timed region) followed by sharded.triangulate_pairs_sharded
    (DLT of every Lowe survivor on the owning rank, all-gather of the float32 x 4 points).  STRONG scaling: the job is the
    same 255 pairs whatever N.  Image k + 1 carries 30 % planted twins of image k; they must come back as nearest neighbours."""
    from sfm_mvs_amd import sharded
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from datagen import load_pose_csv
    n_img, n_desc, n_plant = max(2, args.images or 256), 50_000, 15_000
    pairs = sharded.sequential_pairs(n_img)

    # The exchange of a ONE-rank run goes through a real one-rank RCCL group (the same all_gather_into_tensor as at N > 1),
    # created for this leg when the process has none; if RCCL cannot be initialised the leg says "local copy".
    import torch.distributed as dist
    own_group, exchange_kind = False, "RCCL all_gather_into_tensor"
    if not (dist.is_available() and dist.is_initialized()):
        try:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29541")
            dist.init_process_group("nccl", device_id=dev, world_size=1, rank=0)
            own_group = True
            exchange_kind = "RCCL all_gather_into_tensor on a one-rank group created for this leg"
        except Exception as e:      # noqa: BLE001
            exchange_kind = f"local copy (no process group: {type(e).__name__})"
    try:
        return _bench_c5(args, world, rank, dev, sharded, pairs, n_img, n_desc, n_plant, exchange_kind)
    finally:
        if own_group:
            torch.cuda.synchronize()
            dist.destroy_process_group()


def _bench_c5(args, world, rank, dev, sharded, pairs, n_img, n_desc, n_plant, exchange_kind):
    from sfm_mvs_amd import ops
    from datagen import load_pose_csv

    def base(k):          # image k before its planted rows: a function of k alone, so every rank generates the same image
        g = torch.Generator(device=dev).manual_seed(100 + k)
        d = torch.randn((n_desc, 128), generator=g, device=dev).abs_().square_()
        d /= d.norm(dim=1, keepdim=True)
        d = torch.minimum(d, torch.tensor(0.2, device=dev))
        d /= d.norm(dim=1, keepdim=True)
        return (d * 512).round_().clamp_(0, 255), g

    def image(k):         # rows [0, n_plant) = noisy twins of rows >= n_plant of image k - 1 (rows no image overwrites)
        d, g = base(k)
        if k == 0:
            return d, None
        src = n_plant + torch.randperm(n_desc - n_plant, generator=g, device=dev)[:n_plant]
        d[:n_plant] = (base(k - 1)[0][src] + torch.randn((n_plant, 128), generator=g, device=dev).mul_(2).round_()).clamp_(0, 255)
        return d, src

    mine = sharded.halo_images(pairs, world, rank)
    imgs, planted = [None] * n_img, [None] * n_img
    for k in mine:
        imgs[k], planted[k] = image(k)
    g = torch.Generator(device="cpu").manual_seed(7)
    kps = [torch.rand((n_desc, 2), generator=g).mul_(900.0).to(dev) if k in mine else None for k in range(n_img)]
    _, P = load_pose_csv()
    proj = [P[k % len(P)] for k in range(n_img)]
    eng = sharded.HipMatchEngine(dev, 0.70, depth=PIPE_DEPTH)
    warm = [(mine[0], mine[1])] * 3 if len(mine) >= 2 else []
    if warm:                                                 # warm-up: streams, kernels, the collective
        wd = [imgs[k] if k in (mine[0], mine[1]) else None for k in range(n_img)]
        wpairs = [(mine[0], mine[1])] * world * 3
        wstore, wnq = sharded.match_pairs_sharded(wd, wpairs, n_desc=[n_desc] * n_img, engine=eng, device=dev, batch=EXCH_BATCH)
        sharded.triangulate_pairs_sharded(wstore, wnq, wpairs, kps, proj, batch=EXCH_BATCH)
        del wstore
    barrier_sync(world)
    st_m, st_t = {}, {}
    t0 = time.perf_counter()
    store, nq = sharded.match_pairs_sharded(imgs, pairs, n_desc=[n_desc] * n_img, engine=eng, device=dev, batch=EXCH_BATCH, stats=st_m)
    torch.cuda.synchronize()
    t_match = time.perf_counter() - t0
    pts, counts = sharded.triangulate_pairs_sharded(store, nq, pairs, kps, proj, batch=EXCH_BATCH, stats=st_t)
    barrier_sync(world)
    elapsed = max_over_ranks(time.perf_counter() - t0, world, dev)
    t_match = max_over_ranks(t_match, world, dev)
    # every rank holds every pair's block: planted twins recovered as nearest neighbours (checked for the pairs whose source
    # rows this rank knows, i.e. whose train image it generated)
    hits = tot = 0
    for p, (i, j) in enumerate(pairs):
        if planted[j] is not None:
            # query = image i rows `src`, train = image j rows [0, n_plant): twin of query row src[r] is train row r
            hits += int((store[p, 0, :, 0].index_select(0, planted[j]) == torch.arange(n_plant, device=dev, dtype=torch.int32)).sum().item())
            tot += n_plant
    n_pairs = len(pairs)
    out = {"metric": "descriptor-pair distances/sec over an image sequence (BF-KNN k=2 + Lowe ratio), pair-sharded with the match-record and 3-D point all-gathers",
           "value": n_pairs * n_desc * n_desc / elapsed, "unit": "distances/s", "n_gpus": world, "steps": n_pairs, "warmup": 3,
           "ms_per_step": elapsed / n_pairs * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
           "dtype": "f32 results; filter arithmetic exact-integer i8 MFMA (u8-integer descriptors)", "data": "synthetic",
           "config": {"workload": f"BASELINE configs[4]: {n_img} images x 50k SIFT-like descriptors in total, {n_pairs} sequential pairs sharded "
                                  f"{world}-way (halo partition), all-gather of the KNN blocks ({EXCH_BATCH} pairs per collective) and of the "
                                  f"triangulated points; exchange = {exchange_kind}; 30 % planted matches", "images": n_img, "descriptors": n_desc,
                      "exchange": exchange_kind,
                      "parallelism": f"pair-sharded x{world} (sharded.match_pairs_sharded + triangulate_pairs_sharded); {PIPE_DEPTH} pairs in flight per GPU"},
           "job_seconds": elapsed, "match_seconds": t_match, "triangulate_and_gather_seconds": elapsed - t_match,
           "images_resident_on_this_rank": len(mine),
           "pairs_per_rank": [hi - lo for lo, hi in (sharded.shard_range(len(pairs), world, r) for r in range(world))],
           "exchange": {"kind": exchange_kind, "match_records": st_m, "points": st_t,
                        "note": "device time between the events bracketing each all_gather_into_tensor (includes waiting for the batch's producers)"},
           "triangulated_points_total": int(counts.sum().item()),
           "planted_matches_recovered_as_nearest_neighbour": hits / max(tot, 1),
           "ratio_survivors_per_pair_mean": float(counts.float().mean().item())}
    # roofline of the dominant kernel at this shape: one 50k x 50k pair alone on the device, the filter launched PROF_REPEAT
    # times inside the library's event pair (as the headline leg does)
    if len(mine) >= 2:
        pm = ops.PairMatcher(n_desc, n_desc, dev, 0.70)
        a, b = imgs[mine[0]], imgs[mine[1]]
        pm.run(a, b)
        torch.cuda.synchronize()
        ops.profile_read(0), ops.profile_read(1)
        for _ in range(4):
            ops.profile_enable(PROF_REPEAT)
            pm.run(a, b)
            ops.profile_enable(False)
            torch.cuda.synchronize()
        f_ms, f_n = ops.profile_read(0)
        r_ms, r_n = ops.profile_read(1)
        mode = int(pm.stats[3].item())
        peak, unit, sus = (I8_MFMA_PEAK_TOPS, "TOP/s", I8_MFMA_SUSTAINED_TOPS) if mode == 4 else (BF16_MFMA_PEAK_TFLOPS, "TFLOP/s", F16_MFMA_SUSTAINED_TFLOPS)
        ach = n_desc * n_desc * FLOP_PER_DISTANCE / (f_ms / max(f_n, 1) * 1e-3) / 1e12
        out["roofline"] = {"bound": "mfma", "achieved": ach, "peak": peak, "unit": unit, "frac": ach / peak, "frac_of_sustained": ach / sus,
                           "kernel": "knn_filter_q4_kernel<0> (" + ("filter_i8_body" if mode == 4 else "16-bit body") + "), one 50k x 50k pair per launch",
                           "avg_launch_ms": f_ms / max(f_n, 1), "launches": f_n, "refine_avg_launch_ms": r_ms / max(r_n, 1),
                           "algorithmic_flop_per_launch": n_desc * n_desc * FLOP_PER_DISTANCE, "traffic": None,
                           "note": "256 integer ops per distance (SURVEY 8d); peak = dense int8 MFMA; sustained = profiles/r04_mfma_ceiling.md"}
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            from oracle import oracle as O
            cores = os.cpu_count() or 1
            qh, th = a.cpu().numpy(), b.cpu().numpy()
            probe = min(n_desc, 64 * cores)
            t1 = time.perf_counter()
            O.knn2(qh[:probe], th, nthreads=cores)
            rate = probe * n_desc / (time.perf_counter() - t1)
            rows = int(min(n_desc, max(probe, rate * 10.0 / n_desc)))
            t1 = time.perf_counter()
            wi, wd = O.knn2(qh[:rows], th, nthreads=cores)
            dt = time.perf_counter() - t1
            gi = store[0, 0, :rows].cpu().numpy() if pairs[0] == (mine[0], mine[1]) else None
            out["cpu_baseline"] = {"value": rows * n_desc / dt, "unit": "distances/s", "cores": cores, "kind": "port",
                                   "sample": f"the first {rows} query rows of pair 0 x its 50 000 train rows, once, oracle orc_knn2_l2_f32 "
                                             f"(OpenMP over query rows, {cores} threads), {dt:.1f} s",
                                   "indices_identical_to_hip_on_the_sample": None if gi is None else bool(np.array_equal(gi, wi))}
    return out


def bench_allpairs(args, world, rank, dev):
    """isfm.py:56-94 — EXHAUSTIVE matching: every image against every earlier one (`--images` 64 -> 2 016 pairs of 10 000
    x 10 000 SIFT-like descriptors), the pair grid dealt to the ranks by SURVEY 8e's 2-D block-cyclic split
    (sharded.block_cyclic_partition: a rank holds the image blocks of one process-grid row and column only), KNN + ratio
    through sharded.match_pairs_sharded with the all-gather of the KNN blocks inside the timed region.  STRONG scaling.
    Then isfm.py:80-94 (findEssentialMat RANSAC + recoverPose, the printed inlier count) on the pairs among the first
    `--verify-images` images through sharded.verify_pairs_sharded, with the oracle's counts beside them."""
    from sfm_mvs_amd import sharded
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from datagen import ring_scene
    import torch.distributed as dist
    n_img, n_desc, n_scene = max(2, args.images or 64), 10_000, 7_000
    xb = 4 * EXCH_BATCH                                          # pairs per collective: 5 MB blocks (a 10k-query pair is 160 KB; few, large collectives)
    pairs = sharded.all_pairs(n_img)
    part = sharded.block_cyclic_partition(pairs, n_img, world)
    mine = sharded.halo_images(pairs, world, rank, part)
    K, P, image = ring_scene(n_img, n_desc, n_scene, seed=11)
    kps, des = [None] * n_img, [None] * n_img
    for k in mine:
        kp, d, _ = image(k)
        kps[k], des[k] = torch.from_numpy(kp).to(dev), torch.from_numpy(d).to(dev)
    own_group, exchange_kind = False, "RCCL all_gather_into_tensor"
    if not (dist.is_available() and dist.is_initialized()):
        try:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29542")
            dist.init_process_group("nccl", device_id=dev, world_size=1, rank=0)
            own_group, exchange_kind = True, "RCCL all_gather_into_tensor on a one-rank group created for this leg"
        except Exception as e:      # noqa: BLE001
            exchange_kind = f"local copy (no process group: {type(e).__name__})"
    try:
        eng = sharded.HipMatchEngine(dev, 0.70, depth=PIPE_DEPTH)
        if len(mine) >= 2:                                       # warm-up: streams, kernels, the collective
            wp = [(mine[0], mine[1])] * (world * xb)
            sharded.match_pairs_sharded(des, wp, n_desc=[n_desc] * n_img, engine=eng, device=dev, batch=xb)
        barrier_sync(world)
        st = {}
        t0 = time.perf_counter()
        store, nq = sharded.match_pairs_sharded(des, pairs, n_desc=[n_desc] * n_img, engine=eng, device=dev, batch=xb, partition=part, stats=st)
        barrier_sync(world)
        elapsed = max_over_ranks(time.perf_counter() - t0, world, dev)
        # geometric verification (isfm.py:80-94) of the pairs among the first images: every rank verifies the ones it owns
        nv = min(n_img, max(2, args.verify_images))
        only = {p for p, (j, i) in enumerate(pairs) if i < nv and j < nv}
        need = sorted({i for p in only for i in pairs[p]})
        for k in need:                                           # (a rank verifies only pairs it owns: their images are resident)
            if kps[k] is None and any(int(p) in only for p in part[rank]):
                kp, _, _ = image(k)
                kps[k] = torch.from_numpy(kp).to(dev)
        t1 = time.perf_counter()
        counts = sharded.verify_pairs_sharded(store, nq, pairs, kps, K, partition=part, only=only)
        t_verify = time.perf_counter() - t1
    finally:
        if own_group:
            torch.cuda.synchronize()
            dist.destroy_process_group()
    n_pairs = len(pairs)
    loads = [len(x) for x in part]
    out = {"metric": "descriptor-pair distances/sec over an exhaustive pair list (BF-KNN k=2 + Lowe ratio), 2-D block-cyclic pair sharding",
           "value": n_pairs * n_desc * n_desc / elapsed, "unit": "distances/s", "n_gpus": world, "steps": n_pairs, "warmup": xb,
           "ms_per_step": elapsed / n_pairs * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
           "dtype": "f32 results; filter arithmetic exact-integer i8 MFMA (u8-integer descriptors)", "data": "synthetic",
           "config": {"workload": f"isfm.py:56-94 exhaustive matching: {n_img} images x {n_desc} SIFT-like descriptors ({n_scene} scene points seen by every "
                                  f"camera of a ring + clutter), all {n_pairs} pairs (j < i), block-cyclic over a {sharded.process_grid(world)[0]} x {sharded.process_grid(world)[1]} "
                                  f"process grid; exchange = {exchange_kind}", "images": n_img, "descriptors": n_desc, "pairs": n_pairs,
                      "parallelism": f"pair-sharded x{world} (sharded.block_cyclic_partition + match_pairs_sharded, 8 pairs per launch set, {xb} per collective)"},
           "job_seconds": elapsed, "images_resident_on_this_rank": len(mine), "pairs_per_rank": loads,
           "exchange": {"kind": exchange_kind, "match_records": st},
           "verification": {"what": f"isfm.py:80-94 on the {len(only)} pairs among the first {nv} images: findEssentialMat(RANSAC, 0.999, 0.4) + recoverPose; inliers left per pair",
                            "seconds": t_verify, "inliers": {f"{pairs[p][0]}-{pairs[p][1]}": int(counts[p]) for p in sorted(only)}}}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as O
        t1 = time.perf_counter()
        want, same_knn = {}, True
        for p in sorted(only):
            j, i = pairs[p]
            (kj, dj, _), (ki, di, _) = image(j), image(i)
            wi, wd = O.knn2(dj, di, nthreads=os.cpu_count() or 1)
            same_knn = same_knn and np.array_equal(store[p, 0].cpu().numpy(), wi) and np.array_equal(store[p, 1].cpu().numpy().view(np.float32), wd)
            q, t, _ = O.ratio_filter(wi, wd, 0.70)
            E, m = O.find_essential_mat(kj[q], ki[t], K, 0.999, 0.4)
            if E is None:
                want[p] = -1
                continue
            keep = m.ravel() == 1
            _, _, _, m2 = O.recover_pose(E, kj[q][keep], ki[t][keep], K)
            want[p] = int((m2.ravel() > 0).sum())
        dt = time.perf_counter() - t1
        out["verification"]["oracle_inliers_identical"] = all(int(counts[p]) == want[p] for p in only)
        out["verification"]["knn_blocks_identical_to_oracle"] = bool(same_knn)
        out["cpu_baseline"] = {"value": len(only) * n_desc * n_desc / dt, "unit": "distances/s", "cores": os.cpu_count() or 1, "kind": "port",
                               "sample": f"the same {len(only)} pairs end to end (KNN on all cores + sequential ratio / E-RANSAC / recoverPose), oracle, {dt:.1f} s"}
    return out



def bench_dry_run(args, world, rank):
    """--dry-run-dist: what the N-rank launch does around the kernels, on CPU tensors over gloo.  The knn leg's step protocol
    (next_slot -> fill -> commit -> flush, one all-gather per EXCH_BATCH pairs, barrier + max-over-ranks timing) and, for
    --workload c5, the strong-scaling partition (contiguous pair blocks + one halo image) run for real; a slot is filled with
    a (rank, pair) stamp instead of a KNN block — there is no CPU compute path — and every rank checks every gathered slot."""
    import torch.distributed as dist
    from sfm_mvs_amd import sharded
    dev = torch.device("cpu")
    nq = 64
    pbatch = max(1, min(8, args.pair_batch))
    steps = max(1, min(args.steps, 8))
    ex = sharded.BatchedExchange((2, nq, 2), torch.int32, dev, batch=EXCH_BATCH, nbuf=(args.pipe_depth or PIPE_DEPTH) + 1)
    assert ex.world == world == dist.get_world_size() and ex.rank == rank
    ok, serial = True, 0
    barrier_sync(world)
    t0 = time.perf_counter()
    for _ in range(steps):
        for b in range(pbatch):
            slot, _ = ex.next_slot()
            slot.fill_(rank * 1_000_000 + serial)
            serial += 1
            if ex.commit():
                got, filled = ex.flush(())
                base = serial - filled
                for r in range(world):
                    for k in range(filled):
                        ok = ok and bool((got[r, k] == r * 1_000_000 + base + k).all())
    if ex.fill:
        ex.flush(())
    barrier_sync(world)
    elapsed = max_over_ranks(time.perf_counter() - t0, world, dev)
    out = {"metric": "descriptor-pair distances/sec (BF-KNN k=2 + Lowe ratio)", "value": None, "unit": "distances/s", "dry_run": True,
           "n_gpus": dist.get_world_size(), "steps": steps, "warmup": 0, "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "none (dry run: no kernels)", "data": "rank-stamped slots",
           "config": {"workload": f"dry run of the {args.workload} leg's N-rank protocol on CPU tensors", "backend": dist.get_backend(),
                      "parallelism": f"pair-sharded x{world} + one all-gather of the match records per {EXCH_BATCH} pairs",
                      "exchange": {"collectives": ex.collectives, "pairs_per_collective": EXCH_BATCH, "ranks": dist.get_world_size(),
                                   "gathered_slots_verified": ok},
                      "launched_by": "bench.py self_launch" if os.environ.get("TORCHELASTIC_RUN_ID") else "external launcher"}}
    if args.workload == "c5" or (args.workload == "knn" and world > 1):
        # (default workload at N > 1: the real run measures BASELINE configs[4] beside the headline — main(); its partition is shown here)
        n_img = max(2, args.images or 256)
        pairs = sharded.sequential_pairs(n_img)
        lo, hi = sharded.shard_range(len(pairs), world, rank)
        held = sharded.halo_images(pairs, world, rank)
        counts = [None] * world
        dist.all_gather_object(counts, {"pairs": hi - lo, "images_held": len(held)})
        ok = ok and sum(c["pairs"] for c in counts) == len(pairs) and all(c["images_held"] == c["pairs"] + (1 if c["pairs"] else 0) for c in counts)
        if args.workload == "c5":
            out["scaling"] = "strong"
            out["config"]["partition"] = counts
        else:
            out["config"]["secondary"] = {"config5_images": n_img, "config5_pairs": len(pairs), "config5_pairs_per_rank": [c["pairs"] for c in counts],
                                          "config5_images_per_rank": [c["images_held"] for c in counts], "config5_scaling": "strong",
                                          "rccl_ranks": dist.get_world_size()}
        out["config"]["exchange"]["gathered_slots_verified"] = ok
    if not ok:
        raise SystemExit("dry run: a gathered slot did not carry its (rank, pair) stamp")
    return out


