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

from benchlib.common import *                      # noqa: E402,F401,F403  (constants, barrier_sync, max_over_ranks, cpu_knn_baseline)
from benchlib.knn import bench_knn                 # noqa: E402
from benchlib.geometry import bench_ba, bench_tri, extra_c4, extras      # noqa: E402
from benchlib.scale import bench_allpairs, bench_c5, bench_dry_run       # noqa: E402
from benchlib.features import bench_sfm, bench_sfm_pixels, bench_sift    # noqa: E402
from benchlib.line import COMPACT_MAX, FULL_JSON, compact_line, emit     # noqa: E402,F401


SECONDARY_LEG_TIMEOUT = float(os.environ.get("SFM_BENCH_C5_TIMEOUT", "240"))          # seconds the configs[4] leg of a default N > 1 run may take before the watchdog reports the headline alone


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
            import threading
            a5 = copy.copy(args)
            a5.images = args.images or 256
            # The headline must survive the secondary leg, whatever happens to it.  An exception on one rank is caught below; a HANG —
            # ranks that disagree on a collective: this leg has never run on more than one rank of real hardware — is not an exception,
            # so a watchdog on every rank ends the process after SECONDARY_LEG_TIMEOUT seconds, rank 0 printing the headline line first.
            done = threading.Event()

            def watchdog():
                if done.wait(SECONDARY_LEG_TIMEOUT):
                    return
                if rank == 0:
                    out["config"]["secondary"] = {"config5_error": f"timed out after {SECONDARY_LEG_TIMEOUT} s (the headline above was measured before it started)",
                                                  "config5_images": a5.images, "config5_pairs": a5.images - 1, "rccl_ranks": world}
                    emit(out, json_fd)
                os._exit(0)
            threading.Thread(target=watchdog, daemon=True).start()
            try:
                r5 = bench_c5(a5, world, rank, dev)
            except Exception as e:      # noqa: BLE001 — the headline must survive a failure of the secondary leg
                r5 = {"error": f"{type(e).__name__}: {e}"}
            done.set()
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
