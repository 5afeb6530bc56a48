"""SIFT (`--workload sift`) and BASELINE configs[2]: the 57-camera driver from features (`--workload sfm`) and from pixels (`--from-pixels`)."""
from .common import *  # noqa: F401,F403


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


