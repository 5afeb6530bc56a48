#!/usr/bin/env python
"""Hot-path benchmark (contract in the task prompt; workload = BASELINE.json configs[1]).

  python bench.py --gpus 1 --steps K --warmup W                      # config 2: 10k x 10k BF-KNN + ratio
  python bench.py --gpus N                                           # N > 1: starts its own N ranks (self_launch) ...
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N   # ... or runs as one rank of a launcher's N; pair-sharded, weak scaling
  python bench.py --gpus N --dry-run-dist                            # no GPU: the N-rank launch + exchange protocol over gloo

A "step" is one pass of the hot path over one batch of synthetic input: knnMatch(k=2) + Lowe ratio for a batch of
PAIR_BATCH independent image pairs of 10 000 x 10 000 128-D float32 descriptors per rank (one launch set of
sfm_match_batch_l2_f32), inputs resident in HBM, launch sets pipelined over PIPE_DEPTH streams.  For N > 1 every rank owns
its own pairs (the path shards by image pair) and the pairs' match records are exchanged by one RCCL
all-gather per EXCH_BATCH pairs (sfm_mvs_amd.sharded.BatchedExchange), inside the timed region.

Prints ONE JSON line on rank 0.  `roofline` is measured live with HIP events around the dominant
kernel (knn_filter_split2_kernel, v_mfma_f32_32x32x16_f16) via the library's sfm_profile_* hook, on
steps run after the timed region;
`cpu_baseline` times the CPU oracle (a port of OpenCV's batchDistance semantics) on a bounded
sample of the same workload with all host cores.
"""
import argparse
import json
import os
import sys
import time

# The HIP runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); two streams sharing one are serialised by
# each other's barrier packets.  A rank uses three launch-set streams + the stream that issues the collectives + RCCL's
# own: with 4 queues the exchange path lost 14 % (scripts/dev/dev_exchange.py of the round-5 tree).  Must be set before the runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
BF16_MFMA_PEAK_TFLOPS = 2500.0    # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_bf16, dense (no sparsity)
I8_MFMA_PEAK_TOPS = 5000.0        # dense int8 MFMA (v_mfma_i32_32x32x32_i8: 2x the bf16 rate; the guide measured 4 404 TOPS for 32x32)
F16_MFMA_SUSTAINED_TFLOPS, I8_MFMA_SUSTAINED_TOPS = 1691.0, 3619.0   # pure MFMA stream on RANDOM operands, measured: profiles/r04_mfma_ceiling.md
PROF_SAMPLES = 6                   # profiled steps, run alone AFTER the timed region
PIPE_DEPTH = 2                     # KNN launch sets in flight (one stream + workspace each).  Two are ~4 % faster than three (0.176 vs 0.185 ms) — but about one
                                   # fresh pair of streams in 24 is served one after the other by the runtime (0.22 ms, the one-stream figure), so the set-up
                                   # probes the pipeline's streams and keeps the fastest of STREAM_TRIES sets (ops.BatchPipeline.tune_streams; untimed;
                                   # profiles/r05_knn_pipe_depth.txt, scripts/dev/depth2_streams.py of the round-5 tree)
STREAM_TRIES = 3
SIFT_DEPTH = 3                     # SIFT frames in flight
N_SETS = 2                         # sets of PAIR_BATCH distinct image pairs rotating over the steps
PAIR_BATCH = 8                     # independent pairs per launch set = per step (sfm_match_batch_l2_f32): prologue / ramp / kernel boundaries once per batch
PROF_REPEAT = 3                    # filter launches per HIP-event pair on a profiled step (an event pair adds ~7 us to one)
EXCH_BATCH = 8                     # pairs per RCCL all-gather at N > 1
CLOCK_WARMUP_STEPS = 1600 // PAIR_BATCH   # untimed launch sets (~60 ms of load) before the warm-up steps: the device's clock ramp takes ~25 ms
SPLIT_MFMA_PER_TILE, F32_MFMA_PER_TILE = 24, 65
FLOP_PER_DISTANCE = 256           # GEMM form 2*D
HBM_PEAK_GBS = 8000.0
CPU_BASELINE_SECONDS = 4.0         # wall time of the all-cores oracle sample (cores x 4 s of CPU work)
FP64_VALU_PEAK_TFLOPS = 78.6


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="knn", choices=["knn", "tri", "ba", "sfm", "c5", "sift", "allpairs"])
    ap.add_argument("--nq", type=int, default=10000)
    ap.add_argument("--nt", type=int, default=10000)
    ap.add_argument("--images", type=int, default=None, help="workload c5: images in the sequence IN TOTAL (default 256 = BASELINE config 5); workload allpairs: images (default 64)")
    ap.add_argument("--verify-images", type=int, default=6, help="workload allpairs: essential-matrix RANSAC + recoverPose (isfm.py:80-94) on the pairs among the first N images")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--pipe-depth", type=int, default=None, help=f"launch sets (knn: default {PIPE_DEPTH}) / frames (sift: default {SIFT_DEPTH}) in flight per GPU (1 = one stream)")
    ap.add_argument("--pair-batch", type=int, default=PAIR_BATCH, help="independent pairs per launch set (1..8)")
    ap.add_argument("--from-pixels", action="store_true",
                    help="workload sfm: BASELINE configs[2] FROM PIXELS — 57 rendered full-size frames through img_downscale, cvtColor + SIFT and the "
                         "driver (pipeline.run_sfm_images), with a per-stage breakdown and the oracle twin from the same pixels")
    ap.add_argument("--dry-run-dist", action="store_true",
                    help="no GPU: run the N-rank launch, rendezvous, batched all-gather and timing protocol of the knn / c5 legs on CPU tensors over gloo "
                         "(the slots are filled with a rank-stamped pattern instead of KNN blocks); the JSON line carries dry_run: true and no throughput")
    return ap.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment (the driver's command line for the scaling
    curve): re-exec this script under torch.distributed.run, one rank per GPU, rendezvous on 127.0.0.1.  The children find
    WORLD_SIZE set and take the normal path; rank 0 prints the one JSON line on the inherited stdout.  GPU_MAX_HW_QUEUES and
    HSA_ENABLE_IPC_MODE_LEGACY are exported BEFORE any child touches HIP (sharded.py: streams sharing a hardware queue
    serialise, -14 %; the host driver supports dmabuf IPC only).  Returns the launcher's exit code."""
    import socket
    import subprocess
    if not args.dry_run_dist:
        have = torch.cuda.device_count()
        if have < args.gpus:
            print(f"[bench] --gpus {args.gpus} but this node shows {have} GPU(s): refusing to oversubscribe", file=sys.stderr)
            return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("GPU_MAX_HW_QUEUES", "8")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // args.gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"[bench] --gpus {args.gpus}: launching {args.gpus} ranks: {' '.join(cmd[2:9])} bench.py ...", file=sys.stderr)
    sys.stdout.flush()
    return subprocess.call(cmd, env=env)


def init_dist(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.dry_run_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("gloo", world_size=world, rank=rank)
    elif world > 1 or os.environ.get("SFM_BENCH_EXCHANGE"):     # (the env switch runs the N > 1 code path on one rank: dev/test)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local), world_size=world, rank=rank)
        assert dist.get_world_size() == world          # n_gpus in the JSON line = the RCCL group's rank count
    else:
        torch.cuda.set_device(0)
        local = 0
    if world != args.gpus and rank == 0:
        print(f"[bench] note: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    return world, rank, local


def barrier_sync(world):
    gpu = torch.cuda.is_available()
    if gpu:
        torch.cuda.synchronize()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        if gpu:
            torch.cuda.synchronize()


def max_over_ranks(x, world, dev):
    if world == 1:
        return x
    import torch.distributed as dist
    t = torch.tensor([x], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def cpu_knn_baseline(nq, nt, seed_q, seed_t):
    """Oracle (kind 'port') on all host cores, bounded to ~5 s of wall time (the default run must stay short)."""
    from oracle import oracle as O
    cores = os.cpu_count() or 1
    q = torch.rand((nq, 128), generator=torch.Generator().manual_seed(seed_q)).numpy()
    t = torch.rand((nt, 128), generator=torch.Generator().manual_seed(seed_t)).numpy()
    probe = min(nq, 64 * cores)
    O.knn2(q[:probe], t, nthreads=cores)                     # (first call: thread pool start-up, page faults)
    t0 = time.perf_counter()
    O.knn2(q[:probe], t, nthreads=cores)
    rate = probe * nt / (time.perf_counter() - t0)
    rows = int(min(nq, max(probe, rate * CPU_BASELINE_SECONDS / nt)))
    passes, dt = 0, 0.0
    t0 = time.perf_counter()
    while dt < CPU_BASELINE_SECONDS and passes < 1000:       # whole passes until the time budget is spent
        O.knn2(q[:rows], t, nthreads=cores)
        passes += 1
        dt = time.perf_counter() - t0
    out = {"value": passes * rows * nt / dt, "unit": "distances/s", "cores": cores, "kind": "port",
           "sample": f"{passes} pass(es) over the first {rows} of {nq} query rows x {nt} train rows of the same synthetic "
                     f"set, oracle orc_knn2_l2_f32 (direct-form f32, OpenMP over query rows, {cores} threads), {dt:.1f} s"}
    # also one thread, and torch.cdist + topk on the CPU as an independent sanity point (a few seconds each)
    r1 = min(nq, 256)
    t0 = time.perf_counter()
    O.knn2(q[:r1], t, nthreads=1)
    out["one_thread_distances_per_sec"] = r1 * nt / (time.perf_counter() - t0)
    try:
        r2 = min(nq, 2000)
        qt, tt = torch.from_numpy(q[:r2]), torch.from_numpy(t)
        t0 = time.perf_counter()
        d = torch.cdist(qt, tt)
        vals, idx = torch.topk(d, 2, dim=1, largest=False)
        out["torch_cdist_topk_distances_per_sec"] = r2 * nt / (time.perf_counter() - t0)
        out["torch_threads"] = torch.get_num_threads()
        wi, _ = O.knn2(q[:r2], t, nthreads=cores)
        out["torch_topk_first_neighbour_agreement"] = float((idx[:, 0].numpy() == wi[:, 0]).mean())
    except Exception as e:                                    # a sanity point only
        out["torch_cdist_topk_error"] = str(e)
    try:                                                      # SURVEY 8d baseline item 1: the reference's own operator, if the box has it
        import cv2
        cv2.setNumThreads(cores)
        r3 = min(nq, max(probe, 2000))
        t0 = time.perf_counter()
        m = cv2.BFMatcher().knnMatch(q[:r3], t, k=2)
        dt3 = time.perf_counter() - t0
        out["opencv"] = {"value": r3 * nt / dt3, "unit": "distances/s", "kind": "reference", "version": cv2.__version__,
                         "threads": cv2.getNumThreads(), "sample": f"cv2.BFMatcher().knnMatch on the first {r3} query rows, {dt3:.1f} s "
                                                                   "(includes building the DMatch lists, as sfm.py:260 pays for them)",
                         "first_neighbour_agreement_with_oracle": float(np.mean([a[0].trainIdx for a in m] == O.knn2(q[:r3], t, nthreads=cores)[0][:, 0]))}
    except ImportError:
        out["opencv"] = None                                  # cv2 is not installed on this box: the oracle (kind "port") is the baseline
    try:
        model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")]
        out["cpu_model"] = model[0] if model else None
    except OSError:
        pass
    return out


def knn_source_hash():
    """sha256 of the csrc/knn.hip CODE the LOADED library was built from (sfm_build_id(): baked in at build time; comments and
    whitespace do not count): what the PMC traffic stamps under profiles/ must carry to be reported."""
    from sfm_mvs_amd import _lib
    return _lib.knn_code_hash_of_binary()


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


def synth_correspondences(n, seed):
    """SURVEY 8d triangulation input: cameras 1, 2 of the reference's pose.csv, n DISTINCT points uniform in the
    bounding box of its sparse.ply, observations = projection + N(0, 0.3 px), float32."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from datagen import load_pose_csv
    K, P = load_pose_csv()
    rng = np.random.default_rng(seed)
    X = np.stack([rng.uniform(-6.3, 3.6, n), rng.uniform(-2.6, 5.0, n), rng.uniform(3.2, 13.0, n)], 1)
    Xh = np.c_[X, np.ones(n)].T
    out = []
    for Pm in (P[1], P[2]):
        x = Pm @ Xh
        out.append(((x[:2] / x[2]).T + rng.normal(0, 0.3, (n, 2))).astype(np.float32))
    return K, P[1], P[2], X, out[0], out[1]


# Algorithmic fp64 work of one triangulated point (cv2.triangulatePoints = 4x4 one-sided Jacobi SVD), counted, not
# estimated: the oracle runs the identical rotation sequence (results are bit-identical) and counts per point R applied
# rotations and S skipped pairs.  Per applied rotation: dot 7 + threshold 3 + (c, s) 22 + column update with norms 40 +
# V update 24 = 96 FLOP; per skipped pair 10; per point 100 for building A (32), the initial and final norms (60) and
# the float32 division (sqrt and divide counted as one FLOP each).
TRI_FLOP_ROT, TRI_FLOP_SKIP, TRI_FLOP_FIXED = 96, 10, 100


def tri_cpu_baseline_and_flops(P1, P2, x1, x2, n_cpu):
    from oracle import oracle as O
    a, b = np.ascontiguousarray(x1[:n_cpu].T), np.ascontiguousarray(x2[:n_cpu].T)
    O.jacobi_stats()
    t0 = time.perf_counter()
    want = O.triangulate(P1, P2, a, b, normalise_w=True)
    dt = time.perf_counter() - t0
    rot, skip, calls = O.jacobi_stats()
    flop_pt = TRI_FLOP_FIXED + TRI_FLOP_ROT * rot / calls + TRI_FLOP_SKIP * skip / calls
    base = {"value": n_cpu / dt, "unit": "points/s", "cores": 1, "kind": "port",
            "sample": f"the first {n_cpu} of the same correspondences, once, oracle orc_triangulate_dlt (sequential C, 1 thread), {dt:.1f} s"}
    return base, flop_pt, {"rotations_per_point": rot / calls, "skipped_pairs_per_point": skip / calls,
                           "sweeps_per_point": (rot + skip) / calls / 6.0}, want


def extras(dev):
    """The metric's other two legs, measured outside the timed region: triangulated points/s (with the oracle timed
    beside it and the fp64-VALU roofline from the COUNTED work) and the reprojection error of the HIP path relative to the
    oracle on the same inputs."""
    from sfm_mvs_amd import ops
    from oracle import oracle as O
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from datagen import decompose_P
    n = 1_000_000
    K, P1, P2, X, x1, x2 = synth_correspondences(n, seed=2)
    a = torch.from_numpy(np.ascontiguousarray(x1.T)).to(dev)
    b = torch.from_numpy(np.ascontiguousarray(x2.T)).to(dev)
    for _ in range(3):
        ops.triangulate(P1, P2, a, b, normalise_w=True)
    torch.cuda.synchronize()
    ops.profile_read(2)
    ops.profile_enable(True)
    iters = 20
    for _ in range(iters):
        X4 = ops.triangulate(P1, P2, a, b, normalise_w=True)
    ms, cnt = ops.profile_read(2)
    ops.profile_enable(False)
    tri_rate = n * cnt / (ms * 1e-3)
    # the same normalised result through the fast path (inverse iteration on A^T A instead of Jacobi sweeps)
    for _ in range(3):
        ops.triangulate(P1, P2, a, b, normalise_w="fast")
    ops.profile_read(2)
    ops.profile_enable(True)
    for _ in range(iters):
        X4f = ops.triangulate(P1, P2, a, b, normalise_w="fast")
    fms, fcnt = ops.profile_read(2)
    ops.profile_enable(False)
    same = float((X4f == X4).all(0).float().mean().item())
    maxrel = float(((X4f - X4).abs().amax(0) / X4.abs().amax(0)).max().item())
    # ... and the guarded fast path (bit-identical to the faithful one on every point: what the driver and the sharded path use)
    for _ in range(3):
        ops.triangulate(P1, P2, a, b, normalise_w="guarded")
    ops.profile_read(2)
    ops.profile_enable(True)
    for _ in range(iters):
        X4g = ops.triangulate(P1, P2, a, b, normalise_w="guarded")
    gms, gcnt = ops.profile_read(2)
    ops.profile_enable(False)
    guarded = {"pts_per_sec": n * gcnt / (gms * 1e-3), "ms_1e6": gms / gcnt, "hbm_GBs": 32.0 * n / (gms / gcnt * 1e-3) / 1e9,
               "bit_identical_to_faithful_path": bool(torch.equal(X4g.view(torch.int32), X4.view(torch.int32))),
               "note": "normalise_w=3: inverse iteration where the unit vector's float32 casts keep a margin of max(2^-40, 16 eps lambda1/lambda3) from a rounding boundary, compacted Jacobi pass for the rest (its ~30 us latency floor shows at 1e6 points)"}
    # the product path at the north-star size: 1e7 DISTINCT correspondences through the guarded kernel (what pipeline.Triangulation
    # and sharded.triangulate_pairs_sharded call), checked bit for bit against the faithful kernel on the same inputs
    n7 = 10_000_000
    _, P1b, P2b, _, y1, y2 = synth_correspondences(n7, seed=5)
    a7 = torch.from_numpy(np.ascontiguousarray(y1.T)).to(dev)
    b7 = torch.from_numpy(np.ascontiguousarray(y2.T)).to(dev)
    for _ in range(2):
        ops.triangulate(P1b, P2b, a7, b7, normalise_w="guarded")
    ops.profile_read(2)
    ops.profile_enable(True)
    for _ in range(5):
        X7g = ops.triangulate(P1b, P2b, a7, b7, normalise_w="guarded")
    g7ms, g7cnt = ops.profile_read(2)
    ops.profile_enable(False)
    X7 = ops.triangulate(P1b, P2b, a7, b7, normalise_w=True)
    product = {"pts_per_sec": n7 * g7cnt / (g7ms * 1e-3), "ms_1e7": g7ms / g7cnt, "hbm_GBs": 32.0 * n7 / (g7ms / g7cnt * 1e-3) / 1e9,
               "hbm_frac": 32.0 * n7 / (g7ms / g7cnt * 1e-3) / 1e9 / HBM_PEAK_GBS,
               "bit_identical_to_faithful_kernel_on_all_1e7_points": bool(torch.equal(X7g.view(torch.int32), X7.view(torch.int32)))}
    del a7, b7, X7, X7g
    n_cpu = 1_000_000
    base, flop_pt, work, want = tri_cpu_baseline_and_flops(P1, P2, x1, x2, n_cpu)
    got_cpu = X4[:, :n_cpu].cpu().numpy()
    tflops = flop_pt * tri_rate / 1e12
    # reprojection error vs oracle on the first 4000 points
    R, tv = decompose_P(K, P2)
    rvec = O.rodrigues_mat2vec(R)
    Xf = X4[:3, :4000].t().contiguous()
    out = ops.project_residual(torch.from_numpy(np.hstack([rvec, tv])[None]).to(dev), K, Xf, torch.from_numpy(x2[:4000]).to(dev))
    got = float(np.sqrt(out["sumsq"].item()) / 4000)
    ref, _ = O.reprojection_error(np.hstack([R, tv[:, None]]), K, np.ascontiguousarray(want[:3, :4000].T), x2[:4000])
    return {"triangulated_pts_per_sec": product["pts_per_sec"],
            "triangulated_pts_per_sec_note": "the product path (normalise_w=3, guarded: outputs bit-identical to the OpenCV-order Jacobi kernel on every "
                                             "point, checked here on all 1e7) at 1e7 distinct correspondences; the faithful kernel itself: "
                                             "triangulated_pts_per_sec_faithful_kernel (1e6 points, with its roofline and the oracle beside it)",
            "triangulate_product_path_1e7": product,
            "triangulated_pts_per_sec_faithful_kernel": tri_rate, "triangulate_1e6_ms": ms / cnt,
            "triangulate": {"workload": "1e6 DISTINCT correspondences: pose.csv cameras 1, 2, points uniform in the sparse.ply bounding box, sigma 0.3 px",
                            "cpu_baseline": base,
                            "roofline": {"bound": "fp64-valu", "achieved": tflops, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                                         "frac": tflops / FP64_VALU_PEAK_TFLOPS, "flop_per_point": flop_pt, "counted_work": work,
                                         "kernel": "triangulate_kernel<4>", "avg_launch_ms": ms / cnt,
                                         "note": "FLOP counted on the oracle's identical rotation sequence (96 per applied rotation, 10 per "
                                                 "skipped pair, 100 fixed); a wave runs to its slowest lane's sweep count, so issued > algorithmic"},
                            "hbm_GBs": 32.0 * n / (ms / cnt * 1e-3) / 1e9, "hbm_frac": 32.0 * n / (ms / cnt * 1e-3) / 1e9 / HBM_PEAK_GBS,
                            "points_bit_identical_to_oracle": float((got_cpu == want).all(0).mean()),
                            "max_rel_diff_vs_oracle": float((np.abs(got_cpu - want).max(0) / np.abs(want).max(0)).max())},
            "triangulate_hbm_GBs": 32.0 * n / (ms / cnt * 1e-3) / 1e9,
            "triangulate_guarded": guarded,
            "triangulate_fast": {"pts_per_sec": n * fcnt / (fms * 1e-3), "ms_1e6": fms / fcnt,
                                 "hbm_GBs": 32.0 * n / (fms / fcnt * 1e-3) / 1e9, "hbm_frac": 32.0 * n / (fms / fcnt * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                 "points_bit_identical_to_faithful_path": same, "max_rel_diff": maxrel},
            "reproj_error_hip": got, "reproj_error_oracle": ref, "reproj_error_rel_diff": abs(got - ref) / ref}


def bench_tri(args, world, rank, dev):
    from sfm_mvs_amd import ops
    n = 10_000_000
    K, P1, P2, X, x1, x2 = synth_correspondences(n, seed=2 + rank)
    a = torch.from_numpy(np.ascontiguousarray(x1.T)).to(dev)
    b = torch.from_numpy(np.ascontiguousarray(x2.T)).to(dev)
    for _ in range(args.warmup):
        ops.triangulate(P1, P2, a, b, normalise_w=True)
    barrier_sync(world)
    ops.profile_read(2)
    ops.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ops.triangulate(P1, P2, a, b, normalise_w=True)
    barrier_sync(world)
    elapsed = max_over_ranks(time.perf_counter() - t0, world, dev)
    ms, cnt = ops.profile_read(2)
    ops.profile_enable(False)
    gbs = 32.0 * n / (ms / cnt * 1e-3) / 1e9
    for _ in range(2):
        ops.triangulate(P1, P2, a, b, normalise_w="fast")
    ops.profile_read(2)
    ops.profile_enable(True)
    for _ in range(max(args.steps, 3)):
        Xf = ops.triangulate(P1, P2, a, b, normalise_w="fast")
    fms, fcnt = ops.profile_read(2)
    ops.profile_enable(False)
    Xs = ops.triangulate(P1, P2, a, b, normalise_w=True)
    for _ in range(2):
        ops.triangulate(P1, P2, a, b, normalise_w="guarded")
    ops.profile_read(2)
    ops.profile_enable(True)
    for _ in range(max(args.steps, 3)):
        Xg = ops.triangulate(P1, P2, a, b, normalise_w="guarded")
    gms, gcnt = ops.profile_read(2)
    ops.profile_enable(False)
    guarded = {"pts_per_sec": n / (gms / gcnt * 1e-3), "avg_launch_ms": gms / gcnt, "hbm_GBs": 32.0 * n / (gms / gcnt * 1e-3) / 1e9,
               "hbm_frac": 32.0 * n / (gms / gcnt * 1e-3) / 1e9 / HBM_PEAK_GBS,
               "bit_identical_to_faithful_path": bool(torch.equal(Xg.view(torch.int32), Xs.view(torch.int32))),
               "note": "normalise_w=3 (what the driver and the sharded path use): fast path guarded by a conditioning-aware rounding-boundary margin + compacted Jacobi pass"}
    fast = {"pts_per_sec": n / (fms / fcnt * 1e-3), "avg_launch_ms": fms / fcnt, "hbm_GBs": 32.0 * n / (fms / fcnt * 1e-3) / 1e9,
            "hbm_frac": 32.0 * n / (fms / fcnt * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "points_bit_identical_to_faithful_path": float((Xf == Xs).all(0).float().mean().item()),
            "note": "normalise_w=2: inverse iteration on A^T A (LDL^T) instead of OpenCV's Jacobi sweeps, same float32 result"}
    out = {"fast_path": fast, "guarded_path": guarded, "metric": "triangulated points/sec (DLT, cv2.triangulatePoints)", "value": world * n * args.steps / elapsed,
           "unit": "points/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f64", "data": "synthetic",
           "config": {"workload": "north-star synthetic: 1e7 DISTINCT correspondences, pose.csv cameras 1,2, points uniform in the sparse.ply "
                                  "bounding box, sigma 0.3 px", "n": n}}
    if rank == 0 and not args.no_cpu_baseline:
        n_cpu = 2_000_000
        base, flop_pt, work, want = tri_cpu_baseline_and_flops(P1, P2, x1, x2, n_cpu)
        tflops = flop_pt * n / (ms / cnt * 1e-3) / 1e12
        out["cpu_baseline"] = base
        out["roofline"] = {"bound": "fp64-valu", "achieved": tflops, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                           "frac": tflops / FP64_VALU_PEAK_TFLOPS, "traffic": None, "flop_per_point": flop_pt, "counted_work": work,
                           "kernel": "triangulate_kernel<4>", "avg_launch_ms": ms / cnt, "hbm_GBs": gbs, "hbm_frac": gbs / HBM_PEAK_GBS,
                           "note": "fp64-VALU bound (one-sided Jacobi); FLOP counted on the oracle's identical rotation sequence"}
        out["points_bit_identical_to_oracle"] = float((Xs[:, :n_cpu].cpu().numpy() == want).all(0).mean())
    else:
        out["roofline"] = {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": None,
                           "kernel": "triangulate_kernel<4>", "avg_launch_ms": ms / cnt}
    return out


# fp64 work of the dense sweep per observation.  ALGORITHMIC figure (what `achieved` is computed from, fixed across rounds): the
# fused form of the reference's Jacobian — cv2.projectPoints' dR/dr . X products — 101 fused + 37 plain = 239 ~ 240 FLOP (round 3's
# ISA count of the observation's own arithmetic; SURVEY 8d says ~250; round 2 quoted 420 for the unfused form that multiplied the
# structural zeros).  ISSUED by the round-4 kernel, whole camera loop incl. the per-wave fold (ISA count: v_fma / v_fmac = 2,
# v_mul / v_add = 1; PMC SQ_INSTS_VALU agrees: 161): 230 FLOP in 160 vector instructions per observation (round 3, PMC: 180) — the
# rotation derivative as a cross product with the rotated point, the camera table fetched once per camera instead of per point.
BA_FLOP_PER_OBS = 240
BA_ISSUED_FLOP_PER_OBS = 230
BA_VALU_PER_OBS = 160
BA_FLOP_NOTE = ("achieved = ALGORITHMIC FLOP (240 per observation: the fused form of the reference's dR/dr Jacobian, fixed across rounds) / kernel time; "
                "the round-4 kernel ISSUES 230 FLOP in 160 vector instructions per observation, fold included (round 3: 180 instructions, PMC)")


def c4_problem(dev, seed, ncam=500, npt=200_000):
    """BASELINE configs[3]: cameras on a ring, points in the unit ball, dense visibility, sigma 0.5 px, 1 % perturbed
    cameras.  Observations are synthesised on the device with the library's own projection."""
    from sfm_mvs_amd import ops
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from datagen import load_pose_csv, ring_cameras
    K, _ = load_pose_csv()
    g = torch.Generator(device="cpu").manual_seed(seed)
    cams = torch.from_numpy(ring_cameras(ncam)).to(dev)
    X = torch.randn((npt, 3), generator=g)
    X = (X / X.norm(dim=1, keepdim=True).clamp(min=1.0) * torch.rand((npt, 1), generator=g).clamp(min=0.2)).to(dev)
    obs = torch.empty((ncam, npt, 2), dtype=torch.float32, device=dev)
    zero = torch.zeros((npt, 2), device=dev)
    for c in range(ncam):
        obs[c] = ops.project_residual(cams[c:c + 1], K, X, zero, want_proj=True)["proj"]
    obs += 0.5 * torch.randn(obs.shape, device=dev)
    cams_p = cams * (1 + 0.01 * torch.randn(cams.shape, device=dev, dtype=torch.float64))
    return K, cams_p, X, obs


def c4_cpu_baseline(K, cams_p, X, obs, ns):
    """The oracle's sweep (sequential C, one thread: its accumulation order is the reference order) on a bounded slice of
    the same problem: all cameras x the first `ns` points, residual + all four block sets."""
    from oracle import oracle as O
    ncam = cams_p.shape[0]
    cam_idx = np.repeat(np.arange(ncam, dtype=np.int32), ns)
    pt_idx = np.tile(np.arange(ns, dtype=np.int32), ncam)
    ch, Xh, oh = cams_p.cpu().numpy(), X[:ns].cpu().numpy(), obs[:, :ns].reshape(-1, 2).cpu().numpy()
    t1 = time.perf_counter()
    O.project_residual(ch, K, Xh, oh, cam_idx, pt_idx)
    dt = time.perf_counter() - t1
    return {"value": ncam * ns / dt, "unit": "observations/s", "cores": 1, "kind": "port",
            "sample": f"all {ncam} cameras x the first {ns} points of the same problem ({ncam * ns} observations), once, oracle "
                      f"orc_project_residual (sequential C, 1 thread), {dt:.1f} s"}


def extra_c4(dev):
    """configs[3] inside the default run (so that the driver's own bench record carries it): 3 timed sweeps of the 500 x 200k
    dense residual / J^T J sweep with its roofline and the 1-thread oracle beside it, and the reprojection-error operator
    (A5, sfm.py:79-100) at 10^6 points."""
    from sfm_mvs_amd import ops
    K, cams_p, X, obs = c4_problem(dev, 3)
    ncam, npt = cams_p.shape[0], X.shape[0]
    nobs = ncam * npt
    ops.ba_dense_sweep(cams_p, K, X, obs)
    torch.cuda.synchronize()
    ops.profile_read(3)
    ops.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(3):
        ops.ba_dense_sweep(cams_p, K, X, obs)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 3
    ms, cnt = ops.profile_read(3)
    ops.profile_enable(False)
    k_ms = ms / cnt
    gbs = 8.2 * nobs / (k_ms * 1e-3) / 1e9
    out = {"workload": "BASELINE configs[3]: 500 cameras x 200k points dense (1e8 observations), sigma 0.5 px; 3 sweeps",
           "value": nobs / wall, "unit": "observations/s", "ms_per_sweep": wall * 1e3,
           "roofline": {"bound": "fp64-valu", "achieved": BA_FLOP_PER_OBS * nobs / (k_ms * 1e-3) / 1e12, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": BA_FLOP_PER_OBS * nobs / (k_ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS, "kernel": "ba_dense_kernel", "avg_launch_ms": k_ms,
                        "flop_per_observation": BA_FLOP_PER_OBS, "issued_flop_per_observation": BA_ISSUED_FLOP_PER_OBS, "valu_instructions_per_observation": BA_VALU_PER_OBS,
                        "valu_issue_frac_at_peak_clock": BA_VALU_PER_OBS * nobs / 64 * 4 / (1024 * 2.4e9) / (k_ms * 1e-3),
                        "flop_note": BA_FLOP_NOTE,
                        "hbm_GBs_at_8.2_B_per_obs": gbs, "hbm_frac": gbs / HBM_PEAK_GBS, "traffic": None},
           "cpu_baseline": c4_cpu_baseline(K, cams_p, X, obs, 8000)}
    del obs
    # A5 at scale: ReprojectionError of 10^6 points in one camera (projection + f32 diff + fixed-shape fp64 reduction)
    n = 1_000_000
    g = torch.Generator(device="cpu").manual_seed(5)
    Xb = (torch.randn((n, 3), generator=g) * 0.3).to(dev)
    ob = ops.project_residual(cams_p[:1], K, Xb, torch.zeros((n, 2), device=dev), want_proj=True)["proj"] + 0.5 * torch.randn((n, 2), device=dev)
    for _ in range(3):
        ops.project_residual(cams_p[:1], K, Xb, ob, want_proj=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        r = ops.project_residual(cams_p[:1], K, Xb, ob, want_proj=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    out["reprojection_error_1e6"] = {"points_per_sec": n / dt, "ms_per_call": dt * 1e3, "hbm_GBs_at_28_B_per_point": 28.0 * n / dt / 1e9,
                                     "error": float(np.sqrt(r["sumsq"].item()) / n),
                                     "note": "sfm_project_residual, single camera: cam table + residual kernel + two-level fixed-shape fold (12 B X + 8 B obs in, 8 B proj out)"}
    return out


def extra_other_workloads(args, dev):
    """Short runs of the other workloads inside the default run, so that the driver's own bench record carries them
    (each is the `--workload X` leg with fewer steps; a leg that fails is reported, it never takes the headline line down):
    configs[4] (the full 255-pair job on this one GPU), SIFT frames/s, the 57-camera driver."""
    import copy
    out = {}
    for key, fn, over in (("config5", bench_c5, {}), ("allpairs", bench_allpairs, {"images": 32, "verify_images": 4}),
                          ("sift", bench_sift, {"steps": 30, "warmup": 5}), ("sfm57", bench_sfm, {"steps": 2, "warmup": 1}),
                          ("sfm57_from_pixels", bench_sfm_pixels, {"steps": 3, "warmup": 1, "images": 57})):
        a = copy.copy(args)
        for k, v in over.items():
            setattr(a, k, v)
        try:
            r = fn(a, 1, 0, dev)
            keep = ("metric", "value", "unit", "ms_per_step", "config", "roofline", "cpu_baseline", "parity", "frame_latency_ms_single_stream",
                    "job_seconds", "match_seconds", "triangulate_and_gather_seconds", "exchange", "triangulated_points_total", "verification",
                    "images_resident_on_this_rank", "pairs_per_rank", "planted_matches_recovered_as_nearest_neighbour", "scaling", "profile",
                    "ms_per_registered_camera")
            out[key] = {k: r[k] for k in keep if k in r}
        except Exception as e:      # noqa: BLE001 — an extra, not the measurement
            out[key] = {"error": f"{type(e).__name__}: {e}"}
        torch.cuda.synchronize()
    return out


def bench_ba(args, world, rank, dev):
    from sfm_mvs_amd import ops
    ncam, npt = 500, 200_000
    K, cams_p, X, obs = c4_problem(dev, 3 + rank, ncam, npt)
    for _ in range(max(args.warmup, 1)):
        ops.ba_dense_sweep(cams_p, K, X, obs)
    barrier_sync(world)
    ops.profile_enable(True)
    steps = args.steps
    t0 = time.perf_counter()
    for _ in range(steps):
        ops.ba_dense_sweep(cams_p, K, X, obs)
    barrier_sync(world)
    elapsed = max_over_ranks(time.perf_counter() - t0, world, dev)
    ms, cnt = ops.profile_read(3)
    ops.profile_enable(False)
    nobs = ncam * npt
    gbs = 8.2 * nobs / (ms / cnt * 1e-3) / 1e9
    # the solver built on the sweep (outside the timed region): Schur products and a few Levenberg-Marquardt iterations
    from sfm_mvs_amd import ba
    xr = torch.randn((ncam, 6), dtype=torch.float64, device=dev)
    vr = torch.randn((npt, 3), dtype=torch.float64, device=dev)
    ops.ba_schur_wt(cams_p, K, X, xr), ops.ba_schur_w(cams_p, K, X, vr)
    ops.profile_read(5)
    ops.profile_enable(True)
    for _ in range(3):
        ops.ba_schur_wt(cams_p, K, X, xr), ops.ba_schur_w(cams_p, K, X, vr)
    sms, scnt = ops.profile_read(5)
    ops.profile_enable(False)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    _, _, hist = ba.bundle_adjust_schur(cams_p, K, X, obs, iters=4)
    torch.cuda.synchronize()
    lm_s = time.perf_counter() - t1
    solver = {"schur_product_ms": sms / max(scnt, 1), "schur_pairs_per_sec": nobs / (sms / max(scnt, 1) * 1e-3),
              "lm_iterations": len(hist) - 1, "lm_seconds": lm_s, "cost_start": hist[0], "cost_end": hist[-1],
              "cost_noise_floor": 2.0 * nobs * 0.25,
              "note": "Schur-complement LM (sfm_mvs_amd.ba.bundle_adjust_schur): PCG on the reduced camera system, "
                      "S x = B x - W C^-1 W^T x with W never formed (sfm_ba_schur_wt / sfm_ba_schur_w)"}
    cpu = c4_cpu_baseline(K, cams_p, X, obs, 40000) if rank == 0 and not args.no_cpu_baseline else None
    return {"solver": solver, "cpu_baseline": cpu,
            "metric": "BA observations/sec (residual + J^T J sweep)", "value": world * nobs * steps / elapsed,
            "unit": "observations/s", "n_gpus": world, "steps": steps, "warmup": args.warmup,
            "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE configs[3]: 500 cameras x 200k points dense, sigma 0.5 px", "ncam": ncam, "npt": npt},
            "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                         "traffic": None, "kernel": "ba_dense_kernel", "avg_launch_ms": ms / cnt,
                         "fp64_valu_TFLOPs": BA_FLOP_PER_OBS * nobs / (ms / cnt * 1e-3) / 1e12, "flop_per_observation": BA_FLOP_PER_OBS,
                         "fp64_valu_frac": BA_FLOP_PER_OBS * nobs / (ms / cnt * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS,
                         "issued_flop_per_observation": BA_ISSUED_FLOP_PER_OBS, "valu_instructions_per_observation": BA_VALU_PER_OBS, "flop_note": BA_FLOP_NOTE,
                         "fp64_valu_peak_TFLOPs": FP64_VALU_PEAK_TFLOPS}}


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


def bench_sift(args, world, rank, dev):
    """cv2 SIFT detectAndCompute on frames of the reference's working size (sfm.py:40 halves the
    1936 x 1296 photographs to 968 x 648).  No dataset on the box: procedural frames (tests/datagen.scene_image), a
    different one per rank, resident in HBM as uint8.  One step = one frame: scale space, extrema, orientations,
    ordering, descriptors; keypoints and descriptors stay in HBM (they feed the matcher)."""
    from sfm_mvs_amd import ops, sift
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from datagen import scene_image
    w, h = 968, 648
    g_host = scene_image(w, h, 3 + rank)
    gray = torch.as_tensor(g_host).to(dev)
    sift_depth = max(1, args.pipe_depth or SIFT_DEPTH)
    pipe = sift.SiftPipeline(w, h, dev, depth=sift_depth)
    eng = pipe.engines[0]
    for _ in range(max(2, args.warmup)):
        pipe.submit(gray, after=False)
    barrier_sync(world)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pipe.submit(gray, after=False)
    barrier_sync(world)
    elapsed = max_over_ranks(time.perf_counter() - t0, world, dev)
    nkp = int(eng.count[0].item())
    t1 = time.perf_counter()
    for _ in range(10):
        eng.launch(gray)
    torch.cuda.synchronize()
    single_ms = (time.perf_counter() - t1) / 10 * 1e3
    ops.profile_read(6); ops.profile_read(7)
    ops.profile_enable(True)
    for _ in range(5):
        eng.launch(gray)
    pyr_ms, pyr_n = ops.profile_read(6)
    des_ms, des_n = ops.profile_read(7)
    ops.profile_enable(False)
    # algorithmic HBM bytes of the scale-space build: every blur reads one float plane and writes two (Gaussian + DoG),
    # the base blur reads and writes one, the 2x upsample writes one (2x decimation is a strided read of the next blur)
    n_oct = int(round(np.log2(min(2 * w, 2 * h)) - 2)) + 1
    px = [((2 * w) >> o) * ((2 * h) >> o) for o in range(n_oct)]
    pyr_bytes = sum(p * 12 * 5 for p in px) + px[0] * 8 + px[0] * 4 + w * h
    gbs = pyr_bytes / (pyr_ms / pyr_n * 1e-3) / 1e9
    out = {"metric": "SIFT detectAndCompute frames/sec (968 x 648 uint8 frames)", "value": world * args.steps / elapsed, "unit": "frames/s",
           "n_gpus": world, "steps": args.steps, "warmup": max(2, args.warmup), "ms_per_step": elapsed / args.steps * 1e3,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "SURVEY 8f-1: SIFT (3 layers/octave, 0.04, 10, 1.6) on 968 x 648 procedural frames, one frame per step",
                      "keypoints_per_frame": nkp, "octaves": n_oct,
                      "parallelism": f"frame-sharded x{world}; {sift_depth} frames in flight per GPU"},
           "frame_latency_ms_single_stream": single_ms,
           "keypoints_per_sec": world * nkp * args.steps / elapsed,
           "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": None,
                        "kernel": "scale space: gauss_blur_fixed_kernel<N> x 46 + upsample (one event pair around all of them)",
                        "avg_launch_ms": pyr_ms / pyr_n, "algorithmic_bytes": pyr_bytes},
           "descriptor_kernel": {"avg_launch_ms": des_ms / des_n, "keypoints_per_sec": nkp / (des_ms / des_n * 1e-3),
                                 "note": "VALU/latency bound: per-cell raster walks in the sequential algorithm's float32 order"}}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as orc
        t0 = time.perf_counter()
        kpo, deso = orc.sift(g_host)
        dt = time.perf_counter() - t0
        kp = eng.keypoints[:nkp].cpu().numpy()
        des = eng.descriptors[:nkp].cpu().numpy()
        out["cpu_baseline"] = {"value": 1.0 / dt, "unit": "frames/s", "cores": 1, "kind": "port",
                               "sample": "the same 968 x 648 frame, once, oracle/sift_oracle.c (sequential C)"}
        out["parity"] = {"keypoints_bit_identical": bool(len(kpo) == nkp and np.array_equal(kp.view(np.int32), kpo.view(np.int32))),
                         "descriptors_bit_identical": bool(len(kpo) == nkp and np.array_equal(des, deso))}
    return out


def bench_sfm(args, world, rank, dev):
    """BASELINE configs[2] on Gustav GEOMETRY (the images are not available): the incremental driver over the 57
    cameras of the reference's pose.csv, features rendered from the reference's own cloud."""
    from sfm_mvs_amd import pipeline as pl
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from datagen import decompose_P, gustav_scene
    K, P, feats, ids = gustav_scene(57, seed=3)
    pl.run_sfm(feats[:4], K)                      # warm-up (allocator, first launches)
    barrier_sync(world)
    times = []
    for _ in range(max(1, min(args.steps, 5))):
        t0 = time.perf_counter()
        out = pl.run_sfm(feats, K)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    got = out["posearr"][9:].reshape(-1, 3, 4)
    dR = max(np.abs(decompose_P(K, got[k])[0] - decompose_P(K, P[k])[0]).max() for k in range(57))
    dt = max(np.linalg.norm(decompose_P(K, got[k])[1] - decompose_P(K, P[k])[1]) / max(1.0, np.linalg.norm(decompose_P(K, P[k])[1]))
             for k in range(57))
    sec = float(np.median(times))
    parity = {"max_abs_dR_vs_planted_pose_csv_cameras": float(dR), "max_rel_dt_vs_planted_pose_csv_cameras": float(dt),
              "max_frame_reproj_error": float(max(out["errors"])), "cloud_points": int(len(out["Xtot"]))}
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # north_star's closing bar: the same driver run FREE with every numeric operator replaced by the CPU oracle (the
        # sequential restatement of the cv2 calls): poses, cloud and per-frame errors of all 55 registrations, HIP vs that twin
        from oracle import oracle as O
        from oracle_backend import oracle_pipeline_backend
        t0 = time.perf_counter()
        want = pl.run_sfm(feats, K, be=oracle_pipeline_backend(O))
        dt_cpu = time.perf_counter() - t0
        n = 57
        dP = np.abs(out["posearr"] - want["posearr"])[9:].reshape(n, 12).max(1) / np.abs(want["posearr"][9:]).reshape(n, 12).max(1)
        dE = [abs(a - b) / b for a, b in zip(out["errors"], want["errors"])]
        parity["vs_oracle_twin_free_running_57_frames"] = {
            "same_shapes": bool(out["posearr"].shape == want["posearr"].shape and out["Xtot"].shape == want["Xtot"].shape),
            "max_rel_diff_P": float(dP.max()), "max_rel_diff_frame_error": float(max(dE)),
            "max_rel_diff_cloud": float(np.abs(out["Xtot"] - want["Xtot"]).max() / np.abs(want["Xtot"]).max()),
            "cloud_bit_identical": bool(np.array_equal(out["Xtot"], want["Xtot"])), "tolerance": 1e-4,
            "note": "sfm.py:341-409 run free on both sides; the LM sweep's 28 sums follow one fixed tree in csrc/ransac.hip and oracle/solvers_oracle.c"}
        cpu = {"value": dt_cpu, "unit": "s", "cores": 1, "kind": "port",
               "sample": "the whole 57-camera run once: the same driver with every operator replaced by the oracle (sequential C, 1 thread; KNN included)"}
    return {"cpu_baseline": cpu, "metric": "end-to-end incremental SfM, 57 cameras (s)", "value": sec, "unit": "s", "n_gpus": world,
            "steps": len(times), "warmup": 1, "ms_per_step": sec * 1e3, "higher_is_better": False, "scaling": "replicas",
            "vs_baseline": None, "dtype": "f32/f64", "data": "synthetic (Gustav geometry: pose.csv cameras x sparse.ply points)",
            "config": {"workload": "BASELINE configs[2] on synthetic Gustav geometry", "images": 57,
                       "features_per_image": int(np.mean([len(f[0]) for f in feats]))},
            "parity": parity}


def bench_sfm_pixels(args, world, rank, dev):
    """BASELINE configs[2] from PIXELS (sfm.py:301-409 at full length; the Gustav photographs are not available): 57 frames of
    1936 x 1296 rendered along the reference's own camera path (pose.csv) around textured 3-D structure (tests/datagen.py:
    gustav_views) -> img_downscale (pyrDown, sfm.py:40) -> cvtColor + SIFT (sfm.py:243-252) -> knnMatch + ratio -> findEssentialMat /
    recoverPose -> triangulatePoints -> solvePnPRansac per frame.  `value` = the free-running wall time of the whole job, frames
    handed over as host uint8 arrays (the boundary of sfm.py:301: cv2.imread); beside it ONE profiled run (device drained at
    every stage boundary) for the per-stage breakdown and the count of host waits, and the oracle twin from the same pixels."""
    from sfm_mvs_amd import pipeline as pl
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from datagen import decompose_P, gustav_views
    n_img = max(3, min(args.images or 57, 57))
    t0 = time.perf_counter()
    images, K, P = gustav_views(n_img, seed=5)                 # (set-up: rendered with torch on the GPU, handed over as NumPy frames)
    t_render = time.perf_counter() - t0
    pl.run_sfm_images(images[:4], K, downscale=2)              # warm-up: allocator, SIFT pipelines, first launches
    barrier_sync(world)
    times = []
    for _ in range(max(1, min(args.steps, 3))):
        t0 = time.perf_counter()
        out = pl.run_sfm_images(images, K, downscale=2)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    sec = float(np.median(times))
    prof = pl.DriverProfile()
    outp = pl.run_sfm_images(images, K, downscale=2, profile=prof)
    feats = out["features"]
    nfeat = [int(len(f[0])) for f in feats]
    got = out["posearr"][9:].reshape(-1, 3, 4)
    # the planted cameras are pose.csv's: first camera at the origin, unit first baseline — the gauge recoverPose fixes too
    dR = max(np.abs(decompose_P(K, got[k])[0] - decompose_P(K, P[k])[0]).max() for k in range(n_img))
    dC = max(np.linalg.norm(-decompose_P(K, got[k])[0].T @ decompose_P(K, got[k])[1] + decompose_P(K, P[k])[0].T @ decompose_P(K, P[k])[1]) for k in range(n_img))
    parity = {"profiled_run_identical_to_free_run": bool(np.array_equal(outp["posearr"], out["posearr"]) and np.array_equal(outp["Xtot"], out["Xtot"])),
              "max_abs_dR_vs_planted_cameras": float(dR), "max_camera_centre_error_vs_planted (first baseline = 1)": float(dC),
              "max_frame_reproj_error_px": float(max(out["errors"])), "cloud_points": int(len(out["Xtot"])),
              "note": "planted = the reference's pose.csv cameras the frames were rendered from; the reconstruction sees them only through pixels "
                      "(SIFT localisation noise, planar structure), so this is accuracy of the whole chain, not bit parity — that is vs_oracle_twin"}
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # the twin: the oracle's pyrDown -> cvtColor -> SIFT on the same frames (one thread per frame, ctypes releases the GIL),
        # then the same driver with every operator replaced by the oracle, run free
        from concurrent.futures import ThreadPoolExecutor
        from oracle import oracle as O
        from oracle_backend import oracle_pipeline_backend
        t0 = time.perf_counter()
        def cpu_features(im):
            kp, des = O.sift(O.bgr2gray(O.pyrdown(im)))
            return np.ascontiguousarray(kp[:, :2]), des
        workers = max(1, min(n_img, (os.cpu_count() or 1)))
        with ThreadPoolExecutor(workers) as ex:
            feats_o = list(ex.map(cpu_features, images))
        t_feat = time.perf_counter() - t0
        same_feat = all(np.array_equal(_host(a[0]).view(np.int32), b[0].view(np.int32)) and np.array_equal(_host(a[1]), b[1]) for a, b in zip(feats, feats_o))
        t0 = time.perf_counter()
        small_o = [O.pyrdown(im) for im in images]
        want = pl.run_sfm(feats_o, K, images=small_o, be=oracle_pipeline_backend(O))
        t_drv = time.perf_counter() - t0
        dP = np.abs(out["posearr"] - want["posearr"])[9:].reshape(n_img, 12).max(1) / np.abs(want["posearr"][9:]).reshape(n_img, 12).max(1)
        dE = [abs(a - b) / b for a, b in zip(out["errors"], want["errors"])]
        parity["vs_oracle_twin_from_pixels"] = {
            "features_bit_identical_all_frames": bool(same_feat), "same_shapes": bool(out["posearr"].shape == want["posearr"].shape and out["Xtot"].shape == want["Xtot"].shape),
            "max_rel_diff_P": float(dP.max()), "max_rel_diff_frame_error": float(max(dE)),
            "max_rel_diff_cloud": float(np.abs(out["Xtot"] - want["Xtot"]).max() / np.abs(want["Xtot"]).max()) if out["Xtot"].shape == want["Xtot"].shape else None,
            "cloud_bit_identical": bool(np.array_equal(out["Xtot"], want["Xtot"])), "colours_identical": bool(np.array_equal(out["colorstot"], want["colorstot"])),
            "tolerance": 1e-4}
        cpu = {"value": t_feat + t_drv, "unit": "s", "cores": workers, "kind": "port",
               "sample": f"the whole job once: oracle pyrDown + cvtColor + SIFT of the {n_img} frames on {workers} threads ({t_feat:.1f} s), then the driver with "
                         f"every operator replaced by the oracle, sequential ({t_drv:.1f} s)"}
    return {"cpu_baseline": cpu, "metric": f"end-to-end incremental SfM from pixels, {n_img} frames of 1936 x 1296 (s)", "value": sec, "unit": "s", "n_gpus": world,
            "steps": len(times), "warmup": 1, "ms_per_step": sec * 1e3, "higher_is_better": False, "scaling": "replicas",
            "vs_baseline": None, "dtype": "u8 pixels -> f32 features -> f32/f64 geometry", "data": "synthetic (rendered along the reference's pose.csv camera path; surrogate for the Gustav II Adolf photographs)",
            "config": {"workload": "BASELINE configs[2] from pixels (surrogate frames)", "images": n_img, "frame": [1936, 1296], "working_size": [968, 648],
                       "features_per_image_mean": int(np.mean(nfeat)), "features_per_image_min_max": [min(nfeat), max(nfeat)], "render_seconds_setup": t_render},
            "ms_per_registered_camera": sec * 1e3 / n_img,
            "profile": prof.report(n_img - 2), "parity": parity}


def _host(x):
    return x.detach().cpu().numpy() if torch.is_tensor(x) else np.asarray(x)


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


COMPACT_MAX = 5500                 # bytes of the ONE stdout line (the driver keeps an 8 KB tail of stdout + stderr: round 5's 22.5 KB line was not parsed)
FULL_JSON = os.path.join(ROOT, "gpurun_out", "bench_full.json")


def _finite(x):
    """Strict JSON: NaN / Infinity become null."""
    if isinstance(x, float):
        return x if x == x and abs(x) != float("inf") else None
    if isinstance(x, dict):
        return {str(k): _finite(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_finite(v) for v in x]
    if isinstance(x, (np.floating, np.integer)):
        return _finite(x.item())
    return x


def _sig(x, n=6):
    """Numbers to n significant digits (the compact line only; the full record keeps every digit)."""
    if isinstance(x, bool) or not isinstance(x, float):
        return x
    return float(f"{x:.{n}g}")


def _clip(s, n):
    return s if not isinstance(s, str) or len(s) <= n else s[:n - 3].rstrip() + "..."


def _pick(d, keys, clip=120, digits=9):
    return {k: _clip(_sig(d[k], digits), clip) for k in keys if isinstance(d, dict) and k in d}


def compact_line(out):
    """The ONE stdout line: the contract's fields + `roofline` + `cpu_baseline`, numbers and short labels only.
    Everything else — `extra`, variants, notes — goes to gpurun_out/bench_full.json.  Always < COMPACT_MAX bytes: optional
    blocks are dropped, last first, until it fits."""
    top = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "data", "dry_run")
    c = {k: _clip(_sig(out[k], 10), 160) for k in top if k in out}
    c["dtype"] = _clip(out.get("dtype"), 200)
    cfg = out.get("config") or {}
    c["config"] = _pick(cfg, ("workload",), clip=330)
    c["config"].update(_pick(cfg, ("nq", "nt", "dim", "pairs_per_step", "images", "descriptors", "pairs", "frame", "working_size", "cameras", "points", "observations",
                                   "backend", "launched_by", "cold_value", "general_float_value", "rccl_ranks"), clip=80))
    c["config"].update(_pick(cfg, ("parallelism",), clip=240))
    for k in ("exchange", "partition"):
        if k in cfg:
            c["config"][k] = cfg[k]
    if isinstance(cfg.get("secondary"), dict):
        c["config"]["secondary"] = {k: _sig(v) for k, v in cfg["secondary"].items() if v is not None and (not isinstance(v, str) or k.endswith("_scaling"))}   # numbers (+ the one-word scaling kind)
    if isinstance(out.get("roofline"), dict):
        c["roofline"] = _pick(out["roofline"], ("bound", "achieved", "peak", "unit", "frac", "frac_step", "frac_of_sustained", "traffic", "algorithmic_bytes_per_launch",
                                                 "algorithmic_bytes", "kernel", "avg_launch_ms", "launches"), clip=100)
    if isinstance(out.get("cpu_baseline"), dict):
        cb = out["cpu_baseline"]
        c["cpu_baseline"] = _pick(cb, ("value", "unit", "cores", "kind"), clip=40)
        c["cpu_baseline"]["sample"] = _clip(cb.get("sample"), 260)
        c["cpu_baseline"].update(_pick(cb, ("one_thread_distances_per_sec", "torch_cdist_topk_distances_per_sec", "cpu_model"), clip=60))
        c["cpu_baseline"]["opencv"] = _pick(cb["opencv"], ("value", "unit", "kind", "version", "threads")) if isinstance(cb.get("opencv"), dict) else None
    for k in ("exchange", "parity", "job_seconds", "kernels_ms", "cold_value", "cold_ms_per_step"):      # small, optional: dropped first if the line is too long
        if k in out:
            v = out[k]
            c[k] = {kk: _sig(vv) for kk, vv in v.items() if not isinstance(vv, (str, dict, list))} if isinstance(v, dict) else _sig(v)
    c["full_record"] = "gpurun_out/bench_full.json"
    c = _finite(c)
    for drop in ("kernels_ms", "parity", "exchange", "cold_ms_per_step", "cold_value", "job_seconds"):
        if len(json.dumps(c, allow_nan=False)) < COMPACT_MAX:
            break
        c.pop(drop, None)
    if len(json.dumps(c, allow_nan=False)) >= COMPACT_MAX:
        c["config"].pop("secondary", None)
    line = json.dumps(c, allow_nan=False)
    assert len(line) < COMPACT_MAX and "\n" not in line, len(line)
    return line


def emit(out, json_fd):
    """Full record -> gpurun_out/bench_full.json (never stdout / stderr: the driver's 8 KB tail holds both); compact line -> stdout."""
    try:
        os.makedirs(os.path.dirname(FULL_JSON), exist_ok=True)
        with open(FULL_JSON, "w") as f:
            json.dump(_finite(out), f, indent=1, allow_nan=False)
            f.write("\n")
    except OSError as e:
        print(f"[bench] could not write {FULL_JSON}: {e}", file=sys.stderr)
    os.write(json_fd, (compact_line(out) + "\n").encode())


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    # stdout carries exactly ONE line (the JSON): libraries that write to the C-level stdout (RCCL prints a version banner
    # from its own stdio buffer at exit, gloo its connection notes) are sent to stderr for the whole run, the JSON goes to
    # the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if args.dry_run_dist:
        world, rank, _ = init_dist(args)
        out = bench_dry_run(args, world, rank)
        import torch.distributed as dist
        if rank == 0:
            emit(out, json_fd)
        dist.destroy_process_group()
        return
    world, rank, local = init_dist(args)
    dev = torch.device("cuda", local)
    import sfm_mvs_amd
    sfm_mvs_amd.lib()      # fail loudly if the HIP extension is missing
    if args.workload == "knn":
        out = bench_knn(args, world, rank, dev)
        multi = world > 1 or bool(os.environ.get("SFM_BENCH_EXCHANGE"))       # (the env switch: the N > 1 code path on one rank, dev / test)
        if rank == 0 and not multi:
            if not args.no_extras:
                out["extra"] = extras(dev)
                out["extra"]["config4"] = extra_c4(dev)
                out["extra"].update(extra_other_workloads(args, dev))
                # The driver's record keeps `config` in full but only the KEYS of everything else: the other legs' headline figures
                # ride in config.secondary so that they are driver-timed values too, each measured in this run.
                ex = out["extra"]
                def val(*path):
                    d = ex
                    for k in path:
                        d = d.get(k) if isinstance(d, dict) else None
                    return d
                pix = ex.get("sfm57_from_pixels", {})
                out["config"]["secondary"] = {
                    "sift_like_u8_distances_per_sec": out.get("sift_like", {}).get("distances_per_sec"),
                    "sift_like_u8_ms_per_step": out.get("sift_like", {}).get("ms_per_step"),
                    "fp16_body_distances_per_sec": out.get("fp16_body_variant", {}).get("distances_per_sec"),
                    "config5_one_gpu_distances_per_sec": val("config5", "value"), "config5_job_seconds": val("config5", "job_seconds"),
                    "allpairs_distances_per_sec": val("allpairs", "value"),
                    "triangulated_points_per_sec_1e7": val("triangulate_product_path_1e7", "pts_per_sec"),
                    "ba_dense_observations_per_sec": val("config4", "value"), "ba_dense_fp64_valu_frac": val("config4", "roofline", "frac"),
                    "sift_frames_per_sec": val("sift", "value"),
                    "sfm57_from_features_seconds": val("sfm57", "value"),
                    "sfm57_from_pixels_seconds": pix.get("value"), "sfm57_from_pixels_ms_per_camera": pix.get("ms_per_registered_camera"),
                    "sfm57_from_pixels_host_waits_per_camera": (pix.get("profile") or {}).get("host_syncs", {}).get("per_registered_camera"),
                    "sfm57_from_pixels_vs_oracle_twin_max_rel_diff_P": ((pix.get("parity") or {}).get("vs_oracle_twin_from_pixels") or {}).get("max_rel_diff_P"),
                    "note": "the other legs of this same run (details under `extra`, `sift_like`, `fp16_body_variant`); frames of the sfm legs are SURROGATES of the Gustav photographs"}
            if not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_knn_baseline(args.nq, args.nt, 0, 1)
        elif multi and not args.no_extras:
            # The driver's scaling command (`bench.py --gpus N`, N > 1): the headline stays config 2, weak-scaled (so that the N = 1 point of
            # SCALE equals BENCH), and the SAME run also measures BASELINE configs[4] — 256 images x 50k descriptors, the 255 sequential
            # pairs split over the N ranks (+ one halo image each), both all-gathers (match records, triangulated points) over the
            # RCCL group — reported under config.secondary.  Every rank takes part (collectives).
            import copy
            a5 = copy.copy(args)
            a5.images = args.images or 256
            try:
                r5 = bench_c5(a5, world, rank, dev)
            except Exception as e:      # noqa: BLE001 — the headline must survive a failure of the secondary leg
                r5 = {"error": f"{type(e).__name__}: {e}"}
            if rank == 0:
                import torch.distributed as dist
                out["config5"] = r5
                out["config"]["secondary"] = {
                    "config5_distances_per_sec": r5.get("value"), "config5_job_seconds": r5.get("job_seconds"),
                    "config5_match_seconds": r5.get("match_seconds"), "config5_ms_per_pair": r5.get("ms_per_step"),
                    "config5_images": a5.images, "config5_pairs": a5.images - 1, "config5_pairs_per_rank": r5.get("pairs_per_rank"),
                    "config5_exchange_ms_match_records": ((r5.get("exchange") or {}).get("match_records") or {}).get("exchange_ms"),
                    "config5_exchange_ms_points": ((r5.get("exchange") or {}).get("points") or {}).get("exchange_ms"),
                    "config5_scaling": "strong", "rccl_ranks": dist.get_world_size(), "config5_error": r5.get("error")}
    elif args.workload == "tri":
        out = bench_tri(args, world, rank, dev)
    elif args.workload == "sfm":
        out = bench_sfm_pixels(args, world, rank, dev) if args.from_pixels else bench_sfm(args, world, rank, dev)
    elif args.workload == "c5":
        out = bench_c5(args, world, rank, dev)
    elif args.workload == "sift":
        out = bench_sift(args, world, rank, dev)
    elif args.workload == "allpairs":
        out = bench_allpairs(args, world, rank, dev)
    else:
        out = bench_ba(args, world, rank, dev)
    if rank == 0:
        emit(out, json_fd)
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
