"""Constants and helpers shared by the legs of bench.py."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
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


