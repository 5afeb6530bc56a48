"""BASELINE configs[1]: the headline leg (see bench.py's docstring for the step protocol)."""
from .common import *  # noqa: F401,F403


def bench_knn(args, world, rank, dev):
    from sfm_mvs_amd import ops
    nq, nt = args.nq, args.nt
    depth = max(1, args.pipe_depth or PIPE_DEPTH)
    pbatch = max(1, min(8, args.pair_batch))
    # DISTINCT pairs: a launch set matches `pbatch` different (query, train) images (seeds 2 (pbatch (N_SETS rank + s) + b)
    # and + 1), and N_SETS such sets rotate over the steps — the caches, the arithmetic-mode decision and the rescan counts
    # see different images in every slot of a launch set and in consecutive steps.  (Round 2 matched ONE pair eight times.)
    def image(seed, n):
        return torch.rand((n, 128), generator=torch.Generator().manual_seed(seed)).to(dev)
    sets = [[(image(2 * (pbatch * (N_SETS * rank + s) + b), nq), image(2 * (pbatch * (N_SETS * rank + s) + b) + 1, nt)) for b in range(pbatch)]
            for s in range(N_SETS)]
    q, t = sets[0][0]
    # Pairs are independent units.  A step = one pair; the pairs of consecutive steps are issued PAIR_BATCH per
    # launch set (one prep / filter / refine / scatter launch for the batch: a filter workgroup pays its prologue once per
    # batch, and there are PAIR_BATCH times fewer kernel boundaries), and launch sets are pipelined over PIPE_DEPTH streams
    # so that the low-occupancy tail of one (rescans, ordered scatter) and the prep pass of the next overlap a filter kernel.
    pipe = ops.BatchPipeline(nq, nt, dev, ratio=0.70, depth=depth, batch=pbatch)
    pm = pipe.matchers[0]
    import torch.distributed as dist
    exchange = dist.is_available() and dist.is_initialized() and not os.environ.get("SFM_BENCH_NOEX")
    ex = None
    if exchange:
        # The exchange through the package's one multi-GPU code path, sfm_mvs_amd.sharded.BatchedExchange (the
        # class match_pairs_sharded drives and the world-size-2 gloo tests cover): every rank ends up with every pair's
        # {trainIdx x2, distance x2} block (16 B per query).  EXCH_BATCH pairs are written straight into one batch buffer
        # and exchanged by ONE RCCL all-gather (fewer, larger collectives: a 160 KB all-gather per pair costs more in
        # launch + ring latency than the pair itself), issued from one stream in the same order on every rank; two batch
        # buffers alternate.
        from sfm_mvs_amd import sharded
        ex = sharded.BatchedExchange((2, nq, 2), torch.int32, dev, batch=EXCH_BATCH, nbuf=depth + 1)   # a launch set waits for the gather `depth + 1` sets back: all `depth` streams stay busy

    step_no = [0]

    def step():
        """One launch set: `pbatch` independent, distinct pairs through sfm_match_batch_l2_f32 on the next stream of the
        pipeline (+ at N > 1 the all-gather of their match records); consecutive steps take the next set of images."""
        pairs = sets[step_no[0] % N_SETS]
        step_no[0] += 1
        if ex is None:
            for qb, tb in pairs:
                pipe.submit(qb, tb, after=False)             # static inputs, nothing to wait for; the last one launches
            return
        for qb, tb in pairs:
            slot, free_ev = ex.next_slot()
            pipe.submit(qb, tb, after=free_ev if free_ev is not None else False, result=slot)
            if ex.commit():
                pipe.flush()
                ex.flush(pipe.streams)

    def drain():
        pipe.flush()
        if ex is not None and ex.fill > 0:
            ex.flush(pipe.streams)

    # Set-up, not steps.  (1) every stream is created and every matcher's kernels are loaded once (a HIP stream's first
    # launch costs milliseconds).  (2) The device is brought to its sustained clock: after an idle period the MI355X runs the
    # same launch set ~20 % slower and takes ~25 ms of load to ramp up (scripts/dev/dev_ramp.py of the round-5 tree: 43 -> 36 us per pair over the
    # first 200 launch sets), far longer than W warm-up steps; the path is a throughput path (thousands of pairs per job), so
    # the steady state is what is measured.  CLOCK_WARMUP_STEPS untimed steps (~60 ms of load), then the W warm-up steps.
    for st, pmx in zip(pipe.streams, pipe.matchers):
        with torch.cuda.stream(st):
            pmx.run(sets[0])
            pmx.run(sets[0][:1])
    torch.cuda.synchronize()
    # (3) streams that the runtime really serves concurrently (see PIPE_DEPTH): probe, replace, keep the fastest (one rank only: with an
    # exchange in the loop the ranks would have to agree on the collectives the probe issues)
    stream_probe_ms = pipe.tune_streams(sets, tries=STREAM_TRIES) if depth > 1 and ex is None else []
    # COLD figure: the same K steps right after an idle period, before the clock ramp (kernels and streams are loaded, the
    # device is not at its sustained clock) — what a caller that matches one batch now and then sees.
    time.sleep(0.5)
    barrier_sync(world)
    t0 = time.perf_counter()
    for i in range(args.steps):
        step()
    drain()
    barrier_sync(world)
    cold_elapsed = max_over_ranks(time.perf_counter() - t0, world, dev)
    for i in range(CLOCK_WARMUP_STEPS):                      # (a fixed count: every rank issues the same collectives)
        step()
        if i % 16 == 15:
            torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    drain()
    barrier_sync(world)
    if ex is not None:
        ex.exchange_ms()                                     # drop the warm-up collectives' timings
    # The timed region is ONLY step() calls (+ the closing exchange): barrier + device-wide sync on both sides.
    t0 = time.perf_counter()
    for i in range(args.steps):
        step()
    t_enq = time.perf_counter() - t0                         # host time to enqueue the K steps (reported, not the metric)
    drain()
    barrier_sync(world)
    elapsed = time.perf_counter() - t0
    elapsed = max_over_ranks(elapsed, world, dev)
    exchange_ms = ex.exchange_ms() if ex is not None else None       # device time inside the timed region's collectives
    exchange_calls = ex.collectives if ex is not None else 0
    # Roofline sampling, AFTER the timed region: the library brackets its kernels with HIP events on the launch stream
    # when profiling is on.  An event pair costs ~3.5 us of stream time, and with several pairs in flight a kernel's
    # event-to-event time also contains the neighbours' kernels it shares the chip with — so the sampled steps run
    # ALONE (pipeline drained before and after), and on them the (idempotent) filter kernel is launched PROF_REPEAT
    # times inside one event pair so that the event overhead is amortised.
    ops.profile_read(0), ops.profile_read(1)               # clear the slots
    pipe.synchronize()
    for i in range(PROF_SAMPLES):                           # one launch set (a whole pair batch) alone on the device
        ops.profile_enable(PROF_REPEAT)
        pm.run(sets[i % N_SETS])
        ops.profile_enable(False)
        torch.cuda.synchronize()
    drain()
    barrier_sync(world)
    filt_ms, filt_n = ops.profile_read(0)
    ops.profile_read(1)
    # the refine kernel's own time from launch sets with ONE filter launch (behind three back-to-back filter launches the
    # part's clock is at its lowest and the latency-bound refine reads 30-40 % long)
    for i in range(PROF_SAMPLES):
        ops.profile_enable(True)
        pm.run(sets[i % N_SETS])
        ops.profile_enable(False)
        torch.cuda.synchronize()
    ops.profile_read(0)
    ref_ms, ref_n = ops.profile_read(1)
    stats = pm.stats[0].cpu().tolist()

    # HBM-side bytes per launch of the dominant kernel come from PMC passes (rocprofv3 cannot be driven from inside the
    # process); the committed figure is stamped with the sha256 of the kernel source it was measured on and is
    # reported only while that source is unchanged.
    traffic, traffic_note = None, "no PMC figure for this shape"
    tpath = os.path.join(ROOT, "profiles", "knn_traffic.json")
    if os.path.exists(tpath) and (nq, nt) == (10000, 10000):
        tj = json.load(open(tpath))
        if tj.get("knn_hip_code_sha256") == knn_source_hash() and tj.get("pairs_per_launch", 1) == pbatch:
            traffic, traffic_note = tj.get("bytes_per_launch"), f"profiles/knn_traffic.json ({tj.get('source')})"
        else:
            traffic_note = "profiles/knn_traffic.json is stale (csrc/knn.hip or the pair batch changed since the PMC passes): not reported"
    value = world * pbatch * nq * nt * args.steps / elapsed      # every step matches pbatch pairs per GPU
    filt_avg_ms = filt_ms / max(filt_n, 1)
    algo_flop = pbatch * nq * nt * FLOP_PER_DISTANCE        # one filter launch covers the whole pair batch
    pair_flop = nq * nt * FLOP_PER_DISTANCE                 # ... a single-pair launch (the variant legs below) one pair
    achieved = algo_flop / (filt_avg_ms * 1e-3) / 1e12
    # MFMA work actually issued by the filter arithmetic the device chose (stats[3]): one fp16 product per fp32 product
    # (8 MFMAs per 32x32x128 tile) or the 3-product bf16 split (24)
    mode = stats[3]
    int_body = mode in (4, 5)        # v_mfma_i32_32x32x32_i8: exact u8 data (4) or float data QUANTISED to 8 bits (5) — priced against the int8 roof
    mfma_per_tile = {0: 8, 1: 8, 2: SPLIT_MFMA_PER_TILE}.get(mode, 8)
    # (+ the accumulator-init MFMA of every tile, v_mfma_f32_32x32x8_bf16: half the flops of a product MFMA)
    issued = pbatch * (nq / 32.0) * (nt / 32.0) * (mfma_per_tile + 0.5) * 2 * 32 * 32 * 16 / (filt_avg_ms * 1e-3) / 1e12
    if int_body:                     # 4 product MFMAs of K = 32 per 32 x 32 x 128 tile and group, one init MFMA per tile shared by 8 groups
        issued = pbatch * (nq / 32.0) * (nt / 32.0) * (4 + 1.0 / 8) * 2 * 32 * 32 * 32 / (filt_avg_ms * 1e-3) / 1e12
    peak_hl, unit_hl, sus_hl = ((I8_MFMA_PEAK_TOPS, "TOP/s", I8_MFMA_SUSTAINED_TOPS) if int_body
                                else (BF16_MFMA_PEAK_TFLOPS, "TFLOP/s", F16_MFMA_SUSTAINED_TFLOPS))
    mode_name = {0: "fp16 single product (inputs exact in fp16)", 1: "fp16 single product", 2: "bf16 hi+mid split (3 products)",
                 3: "fp32 MFMA", 4: "exact-integer i8 MFMA (u8-integer descriptors)",
                 5: "i8 MFMA on the descriptors QUANTISED to 8 bits (one grid per pair; the certificate uses the measured residual norms)"}.get(mode, str(mode))
    out = {
        "metric": "descriptor-pair distances/sec (BF-KNN k=2 + Lowe ratio)", "value": value, "unit": "distances/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "ms_per_pair": elapsed / (args.steps * pbatch) * 1e3, "host_enqueue_ms_per_step": t_enq / args.steps * 1e3,
        "cold_value": world * pbatch * nq * nt * args.steps / cold_elapsed, "cold_ms_per_step": cold_elapsed / args.steps * 1e3,
        "cold_note": "the same K steps timed after 0.5 s of idle, BEFORE the clock-ramp steps (kernels and streams loaded): `value` is the sustained rate, this the rate a cold device gives",
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": f"f32 results (bit-identical to the direct-form f32 reference); filter arithmetic on MFMA: {mode_name}; "
                 "f32 exact refine",
        "data": "synthetic",
        "config": {"workload": f"BASELINE configs[1]: 10k x 10k uniform[0,1) float32 128-D descriptors, BF-KNN k=2 + "
                               f"Lowe ratio 0.70; a step = one launch set = a batch of {pbatch} distinct pairs of that shape per GPU "
                               f"({pbatch}e8 distances; {pbatch} distinct pairs per launch set, {N_SETS} sets of images rotating over the steps)",
                   "nq": nq, "nt": nt, "dim": 128, "pairs_per_step": pbatch, "distinct_pairs_per_launch_set": pbatch, "image_sets": N_SETS,
                   "parallelism": f"pair-sharded x{world}" + (f" + one RCCL all-gather of the match records per {EXCH_BATCH} pairs" if world > 1 else "")
                                  + f"; independent pairs issued {pbatch} per launch set (sfm_match_batch_l2_f32), {depth} launch sets in flight per GPU (one HIP stream each)",
                   "pairs_per_launch": pbatch,
                   "cold_value": world * pbatch * nq * nt * args.steps / cold_elapsed, "cold_ms_per_step": cold_elapsed / args.steps * 1e3,
                   "cold_note": "`value` is the sustained rate; cold_value = the same K steps after 0.5 s of idle, before the clock ramp",
                   "stream_probe_ms_per_launch_set": stream_probe_ms,
                   "stream_probe_note": f"set-up, untimed: the pipeline's {depth} streams are probed and replaced up to {STREAM_TRIES - 1} times, the fastest set is kept "
                                        "(two launch sets overlap fully only when the runtime serves their streams concurrently: ~1 fresh pair in 24 does not)",
                   "setup": f"streams and kernels loaded, then {CLOCK_WARMUP_STEPS} untimed steps of the same workload (~60 ms: the device reaches its sustained clock) before the W warm-up steps"},
        "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak_hl, "unit": unit_hl,
                     "frac": achieved / peak_hl, "frac_of_sustained": achieved / sus_hl,
                     # the WHOLE step against the same roof: algorithmic work of a launch set / the timed region's time per step
                     # (prep + mode/split + filter + refine + scatter, `pipe_depth` launch sets in flight) — `frac` is the filter kernel alone
                     "frac_step": algo_flop / (elapsed / args.steps) / 1e12 / peak_hl,
                     "frac_step_note": "algorithmic ops of one launch set / ms_per_step / peak: the step as a whole, not its dominant kernel",
                     "sustained_note": ("a pure i8 MFMA stream on random bytes holds 3 619 TOPS on this part (power-limited clock), profiles/r04_mfma_ceiling.md" if int_body else
                                        "a pure fp16 MFMA stream on random operands holds 1 691 TFLOP/s on this part (clock 1.66 GHz: power-limited), profiles/r04_mfma_ceiling.md"),
                     "peak_note": ("dense int8 MFMA peak (the filter ran on v_mfma_i32_32x32x32_i8: 2x the 16-bit rate); against the dense fp16 peak of 2 500 the same figure is "
                                   f"{achieved / BF16_MFMA_PEAK_TFLOPS:.3f}" if int_body else "dense fp16 / bf16 MFMA peak"),
                     "traffic": traffic,
                     "traffic_unit": "bytes/launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE)", "traffic_source": traffic_note,
                     "algorithmic_bytes_per_launch": pbatch * (4 * 128 * (nq + nt) + 16 * nq),
                     "kernel": "knn_filter_q4_kernel<0>" + (" (filter_i8_body on 8-bit quantised operands)" if mode == 5 else " (filter_i8_body)" if mode == 4 else ""),
                     "avg_launch_ms": filt_avg_ms, "launches": filt_n, "pairs_per_launch": pbatch,
                     "algorithmic_flop_per_launch": algo_flop,
                     "issued_mfma_tflops": issued, "issued_frac_of_peak": issued / peak_hl,
                     "launch_sampling": f"HIP events around the filter kernel on {PROF_SAMPLES} launch sets run alone AFTER the timed region "
                                        f"(pipeline drained); on those the kernel is launched {PROF_REPEAT}x back-to-back inside "
                                        "the event pair (idempotent) so that the event overhead (~7 us per pair) is amortised",
                     "note": "algorithmic = 256 FLOP per distance (SURVEY 8d); issued = MFMA flops of the arithmetic mode that ran"},
        "kernels_ms": {"knn_filter": filt_avg_ms, "knn_refine": ref_ms / max(ref_n, 1)},
        "knn_stats": {"rescanned_queries": stats[0], "filter_workgroups": stats[1], "streams_per_query": stats[2],
                      "filter_mode": mode_name},
    }
    if ex is not None:
        out["exchange"] = {"ms_total_in_timed_region": exchange_ms, "ms_per_step": exchange_ms / args.steps, "ms_per_pair": exchange_ms / (args.steps * pbatch),
                           "collectives_since_start": exchange_calls, "pairs_per_collective": EXCH_BATCH,
                           "bytes_per_rank_per_collective": EXCH_BATCH * nq * 16,
                           "note": "device time between the events bracketing each all_gather_into_tensor on the issuing stream "
                                   "(includes waiting for the batch's producers); the pair kernels of the next batch overlap it"}
    # latency of ONE pair launched alone (batch of one, one stream), and of one whole batch, outside the timed region
    pm = pm1 = ops.PairMatcher(nq, nt, dev, ratio=0.70)
    for _ in range(3):
        pm1.run(q, t)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        pm1.run(q, t)
    torch.cuda.synchronize()
    out["pair_latency_ms_single_stream"] = (time.perf_counter() - t0) / 50 * 1e3
    bm0 = pipe.matchers[0]
    t0 = time.perf_counter()
    for _ in range(20):
        bm0.run(sets[0])
    torch.cuda.synchronize()
    out["batch_latency_ms_single_stream"] = (time.perf_counter() - t0) / 20 * 1e3
    same_as_single = True                                   # every pair of the batch against its own single-pair call
    for b, (qb, tb) in enumerate(sets[0]):
        pm1.run(qb, tb)
        same_as_single = same_as_single and bool(torch.equal(bm0.idx[b], pm1.idx) and torch.equal(bm0.dist[b], pm1.dist))
    pm1.run(q, t)
    out["batched_results_identical_to_single_pair_call"] = same_as_single
    if world == 1 and not args.no_extras:
        # SURVEY 8d's second input distribution at the same shape: SIFT-like integer descriptors (0..255, norm 512) with 30 %
        # planted matches — the Lowe mask is non-trivial and known (uniform random data passes the 0.70 test on ~0 rows), and
        # the filter takes its exact single-product path.  Same pipeline, untimed w.r.t. the headline value.
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from datagen import planted_pair
        # DISTINCT pairs here too: N_SETS launch sets of `pbatch` different planted pairs rotate over the steps.  Integer 0..255
        # data: filter="auto" takes the exact-integer body (v_mfma_i32_32x32x32_i8, stats[3] = 4) — the reference's real data
        # (cv2 SIFT output, sfm.py:246-252) — so this leg carries its own roofline against the i8 peak.
        rng_s = np.random.default_rng(0)
        sets_s, planted0 = [], None
        for s_ in range(N_SETS):
            cur = []
            for b_ in range(pbatch):
                qh_, th_, pl_ = planted_pair(rng_s, nq, nt, 0.3)
                if planted0 is None:
                    planted0 = pl_
                cur.append((torch.from_numpy(qh_).to(dev), torch.from_numpy(th_).to(dev)))
            sets_s.append(cur)

        def run_sets(n):
            for i in range(n):
                for qb, tb in sets_s[i % N_SETS]:
                    pipe.submit(qb, tb, after=False)
            pipe.flush(); pipe.synchronize()
        run_sets(CLOCK_WARMUP_STEPS // 2)                      # (the device is at its sustained clock already; kernels of this mode loaded)
        n_sets = 60
        t0 = time.perf_counter()
        run_sets(n_sets)
        dt = time.perf_counter() - t0
        ops.profile_read(0), ops.profile_read(1)
        for i in range(PROF_SAMPLES):
            ops.profile_enable(PROF_REPEAT)
            pipe.matchers[0].run(sets_s[i % N_SETS])
            ops.profile_enable(False)
            torch.cuda.synchronize()
        f8_ms, f8_n = ops.profile_read(0)
        r8_ms, r8_n = ops.profile_read(1)
        f8_avg = f8_ms / max(f8_n, 1)
        bm_s = pipe.matchers[0]
        bm_s.run(sets_s[0]); torch.cuda.synchronize()
        m = int(bm_s.count[0].item())
        got = dict(zip(bm_s.out_q[0, :m].cpu().tolist(), bm_s.out_t[0, :m].cpu().tolist()))
        mode_s = int(bm_s.stats[0, 3].item())
        ach8 = algo_flop / (f8_avg * 1e-3) / 1e12
        peak8 = I8_MFMA_PEAK_TOPS if mode_s == 4 else BF16_MFMA_PEAK_TFLOPS
        i8_traffic, i8_traffic_note = None, "no PMC figure for this shape"
        t8 = os.path.join(ROOT, "profiles", "knn_i8_traffic.json")
        if os.path.exists(t8) and (nq, nt) == (10000, 10000):
            tj8 = json.load(open(t8))
            if tj8.get("knn_hip_code_sha256") == knn_source_hash() and tj8.get("pairs_per_launch", 1) == pbatch:
                i8_traffic, i8_traffic_note = tj8.get("bytes_per_launch"), f"profiles/knn_i8_traffic.json ({tj8.get('source')})"
            else:
                i8_traffic_note = "profiles/knn_i8_traffic.json is stale (csrc/knn.hip or the pair batch changed since the PMC passes): not reported"
        out["sift_like"] = {"distances_per_sec": n_sets * pbatch * nq * nt / dt, "ms_per_pair": dt / (n_sets * pbatch) * 1e3, "ms_per_step": dt / n_sets * 1e3,
                            "filter_mode": {0: "fp16 single product (inputs exact in fp16)", 1: "fp16 single product", 2: "bf16 split",
                                            4: "exact-integer i8 MFMA (v_mfma_i32_32x32x32_i8, i32 scores)"}.get(mode_s),
                            "distinct_pairs_per_launch_set": pbatch, "image_sets": N_SETS,
                            "roofline": {"bound": "mfma", "achieved": ach8, "peak": peak8, "unit": "TOP/s" if mode_s == 4 else "TFLOP/s", "frac": ach8 / peak8,
                                         "frac_of_sustained": ach8 / (I8_MFMA_SUSTAINED_TOPS if mode_s == 4 else F16_MFMA_SUSTAINED_TFLOPS),
                                         "sustained_note": "a pure MFMA stream on random operands holds 3 619 TOPS (i8) / 1 691 TFLOP/s (fp16) on this part: the clock drops to 1.66-1.78 GHz (profiles/r04_mfma_ceiling.md)",
                                         "kernel": "knn_filter_q4_kernel<0> (filter_i8_body)" if mode_s == 4 else "knn_filter_q4_kernel<0>",
                                         "avg_launch_ms": f8_avg, "launches": f8_n, "pairs_per_launch": pbatch, "algorithmic_flop_per_launch": algo_flop, "traffic": i8_traffic,
                                         "traffic_unit": "bytes/launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE)", "traffic_source": i8_traffic_note,
                                         "algorithmic_bytes_per_launch": pbatch * (4 * 128 * (nq + nt) + 16 * nq),
                                         "note": "algorithmic = 256 integer ops per distance (SURVEY 8d, GEMM form 2 D); peak = dense int8 MFMA (MI355X_MICROARCH.md: ~5 P dense, 4 404 TOPS measured for 32x32)"},
                            "kernels_ms": {"knn_filter": f8_avg, "knn_refine": r8_ms / max(r8_n, 1)},
                            "rescanned_queries_pair0": int(bm_s.stats[0, 0].item()),
                            "ratio_survivors": m, "planted_matches": int(len(planted0)),
                            "planted_matches_among_survivors": int(sum(1 for a, b in planted0.tolist() if got.get(a) == b)),
                            "note": "SIFT-like descriptors (SURVEY 8d (ii)), 30 % planted twins with N(0, 2) integer noise; survivors = Lowe ratio 0.70; same pipeline as the headline value"}
        # boundary handing over HOST buffers: pinned H2D of both descriptor sets + the step + D2H of the results
        qh, th = q.cpu().pin_memory(), t.cpu().pin_memory()
        qd, td = torch.empty_like(q), torch.empty_like(t)
        ih, dh = torch.empty((nq, 2), dtype=torch.int32).pin_memory(), torch.empty((nq, 2), dtype=torch.float32).pin_memory()
        for _ in range(3):
            qd.copy_(qh, non_blocking=True); td.copy_(th, non_blocking=True)
            pm.run(qd, td)
            ih.copy_(pm.idx, non_blocking=True); dh.copy_(pm.dist, non_blocking=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            qd.copy_(qh, non_blocking=True); td.copy_(th, non_blocking=True)
            pm.run(qd, td)
            ih.copy_(pm.idx, non_blocking=True); dh.copy_(pm.dist, non_blocking=True)
        torch.cuda.synchronize()
        out["pcie_inclusive"] = {"distances_per_sec": nq * nt * 20 / (time.perf_counter() - t0),
                                 "note": "pinned-host descriptors in, results out, same stream (not the headline value)"}
        # the 16-bit body on the SAME data, box and pipeline (filter="noquant": what ran by default before the quantised body)
        pipe16 = ops.BatchPipeline(nq, nt, dev, ratio=0.70, depth=depth, batch=pbatch, filter="noquant")
        if depth > 1:
            pipe16.tune_streams(sets, tries=STREAM_TRIES)
        def run16(n):
            for i in range(n):
                for qb, tb in sets[i % N_SETS]:
                    pipe16.submit(qb, tb, after=False)
            pipe16.flush(); pipe16.synchronize()
        run16(40)
        t0 = time.perf_counter()
        run16(60)
        dt16 = time.perf_counter() - t0
        ops.profile_read(0), ops.profile_read(1)
        for i in range(PROF_SAMPLES):
            ops.profile_enable(PROF_REPEAT)
            pipe16.matchers[0].run(sets[i % N_SETS])
            ops.profile_enable(False)
            torch.cuda.synchronize()
        f16_ms, f16_n = ops.profile_read(0)
        r16_ms, r16_n = ops.profile_read(1)
        f16_avg = f16_ms / max(f16_n, 1)
        m16 = pipe16.matchers[0]
        m16.run(sets[0]); bm0.run(sets[0])
        torch.cuda.synchronize()
        same16 = bool(torch.equal(m16.result, bm0.result) and torch.equal(m16.count, bm0.count))
        out["fp16_body_variant"] = {"filter": "noquant", "distances_per_sec": 60 * pbatch * nq * nt / dt16, "ms_per_step": dt16 / 60 * 1e3,
                                    "filter_mode": int(m16.stats[0, 3].item()), "filter_avg_launch_ms": f16_avg, "refine_avg_launch_ms": r16_ms / max(r16_n, 1),
                                    "achieved_tflops": algo_flop / (f16_avg * 1e-3) / 1e12, "peak_tflops": BF16_MFMA_PEAK_TFLOPS,
                                    "frac": algo_flop / (f16_avg * 1e-3) / 1e12 / BF16_MFMA_PEAK_TFLOPS,
                                    "frac_of_sustained": algo_flop / (f16_avg * 1e-3) / 1e12 / F16_MFMA_SUSTAINED_TFLOPS,
                                    "results_identical_to_default": same16,
                                    "frac_step": algo_flop / (dt16 / 60) / 1e12 / BF16_MFMA_PEAK_TFLOPS,
                                    "note": "the same launch sets through the fp16 single-product body (same pipeline depth): the headline value runs the i8 MFMA "
                                            "body on 8-bit quantised operands instead; bit-identical results (tests/test_gpu_knn_q8.py::test_full_size_batch_quantised_equals_noquant)"}
        # ADVICE r04 / VERDICT r04 item 4: the headline ran the i8 body on QUANTISED operands, which only data of compact support
        # (uniform, beta) qualify for; Gaussian / heavy-tailed float descriptors take the fp16 body — that rate, on the same box and
        # shape, is the one to quote for float descriptors in general (u8 SIFT output: `sift_like`)
        out["config"]["general_float_value"] = out["fp16_body_variant"]["distances_per_sec"]
        out["config"]["general_float_note"] = ("distances/s of the same launch sets through filter = noquant (fp16 body): what float descriptors WITHOUT compact support "
                                               "(Gaussian, unit-norm, RootSIFT-like) get; `value` applies to uniform-like data, `sift_like` to the reference's real u8 input")
        del pipe16
        # the exact-f32-MFMA filter variant on the same inputs (identical results), for the fp32 roofline
        pm32 = ops.PairMatcher(nq, nt, dev, ratio=0.70, filter="f32")
        for _ in range(5):
            pm32.run(q, t)
        torch.cuda.synchronize()
        ops.profile_enable(True)
        for _ in range(20):
            pm32.run(q, t)
        f32_ms, f32_n = ops.profile_read(0)
        ops.profile_read(1)
        ops.profile_enable(False)
        same = bool(torch.equal(pm32.idx, pm.idx) and torch.equal(pm32.dist, pm.dist))
        f32_avg = f32_ms / max(f32_n, 1)
        out["fp32_filter_variant"] = {"kernel": "knn_filter_kernel (v_mfma_f32_32x32x2_f32)", "avg_launch_ms": f32_avg,
                                      "achieved_tflops": pair_flop / (f32_avg * 1e-3) / 1e12, "peak_tflops": FP32_MFMA_PEAK_TFLOPS,
                                      "frac": pair_flop / (f32_avg * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, "pairs_per_launch": 1,
                                      "results_identical_to_default": same}
        # ... and the 3-product bf16 split pinned (what the device picks for data outside fp16's comfortable range)
        pms = ops.PairMatcher(nq, nt, dev, ratio=0.70, filter="split")
        for _ in range(5):
            pms.run(q, t)
        torch.cuda.synchronize()
        ops.profile_enable(True)
        for _ in range(20):
            pms.run(q, t)
        sp_ms, sp_n = ops.profile_read(0)
        ops.profile_read(1)
        ops.profile_enable(False)
        sp_avg = sp_ms / max(sp_n, 1)
        out["bf16_split_variant"] = {"kernel": "knn_filter_q4_kernel<0>, split body (3 x v_mfma_f32_32x32x16_bf16 per product)",
                                     "avg_launch_ms": sp_avg, "achieved_tflops": pair_flop / (sp_avg * 1e-3) / 1e12,
                                     "frac": pair_flop / (sp_avg * 1e-3) / 1e12 / BF16_MFMA_PEAK_TFLOPS, "pairs_per_launch": 1,
                                     "issued_mfma_tflops": 3 * pair_flop / (sp_avg * 1e-3) / 1e12,
                                     "results_identical_to_default": bool(torch.equal(pms.idx, pm.idx) and torch.equal(pms.dist, pm.dist))}
    return out


