"""Host-side mirror of the reference's helper functions and driver (sfm.py), on top of the HIP back-end.

Same names, argument meaning, shapes and quirks as the reference so that a user of sfm.py finds the same
interface: `find_features(img0, img1)` (sfm.py:242-270; `match_features` is its matcher half), `Triangulation` (:45-56), `PnP`
(:60-76), `ReprojectionError` (:79-100), `common_points` (:215-239), `to_ply` (:169-201) and the
sliding-window driver (:274-423) as `run_sfm`.  An image is represented by its keypoint coordinates + 128-D
descriptors ("features"); `img_downscale` (:35-42) and `features_from_images` (the cvtColor + SIFT half of
find_features, :243-252) produce them from pixels on the GPU, `run_sfm_images` chains both.

Behavioural quirks that are load-bearing for parity are reproduced and marked `# quirk`.
"""
import os

import numpy as np
import torch

from . import cv2compat as cv2
from . import ops

RATIO = 0.70          # sfm.py:264


class Backend:
    """What the helper functions run on: the cv2-named facade `cv` plus the fused device operators.  The default
    is the HIP back-end; tests substitute a CPU twin built on the oracle to diff the whole driver."""

    def __init__(self, cv=None, match=None, reproj=None):
        self.cv = cv or cv2
        self.match = match or _match_hip
        self.reproj = reproj or _reproj_hip


_default = None


def _be(be):
    global _default
    if be is not None:
        return be
    if _default is None:
        _default = Backend()
    return _default


def _match_hip(feat0, feat1):
    kp0, des0 = feat0
    kp1, des1 = feat1
    dev = torch.device("cuda")

    def up(a):      # features may already live in HBM (features_from_images(..., on_device=True))
        return a.to(dev, torch.float32).contiguous() if torch.is_tensor(a) else torch.as_tensor(np.ascontiguousarray(a, np.float32)).to(dev)

    d0, d1 = up(des0), up(des1)
    idx, dist = ops.knn2(d0, d1)                                   # bf.knnMatch(des0, des1, k=2)  sfm.py:260
    out_q, out_t, count = ops.ratio_compact(idx, dist, RATIO)      # m.distance < 0.70*n.distance   sfm.py:262-265
    p0, p1 = ops.gather_matches(up(kp0), up(kp1), out_q, out_t, count)
    m = int(count.item())
    return p0[:m].cpu().numpy(), p1[:m].cpu().numpy()              # sfm.py:267-268


def match_features(feat0, feat1, be=None):
    """Matcher half of sfm.py:242-270 on precomputed features.  feat = (kp (n,2) float32 keypoint coordinates,
    des (n,128) float32), NumPy or CUDA tensors.  Returns pts0, pts1 (M,2) float32 in ascending queryIdx order."""
    return _be(be).match(feat0, feat1)


def find_features(img0, img1, be=None):
    """sfm.py:242-270 with the reference's signature: two BGR (or already grey) uint8 frames in, the matched pixel
    coordinates `pts0, pts1` ((M,2) float32, ascending queryIdx) out — cvtColor (:243-244), SIFT detectAndCompute
    (:246-252), BFMatcher.knnMatch k=2 (:259-260), the 0.70 ratio loop (:262-265) and the keypoint gather (:267-268),
    all on the device; the features never visit the host.  (Feature tuples are still accepted and go to match_features:
    the driver computes every image's features once instead of twice as the reference does.)"""
    if isinstance(img0, (tuple, list)) and isinstance(img1, (tuple, list)):
        return match_features(img0, img1, be)
    b = _be(be)
    if b.match is not _match_hip:            # a substituted backend (the tests' CPU twin): its own cv2 facade end to end
        cv = b.cv
        feats = []
        for img in (img0, img1):
            img = np.asarray(img)
            gray = cv.cvtColor(img, cv.COLOR_BGR2GRAY) if img.ndim == 3 else img
            kp, des = cv.xfeatures2d.SIFT_create().detectAndCompute(gray, None)
            feats.append((np.float32([k.pt for k in kp]).reshape(-1, 2), des))
        return b.match(feats[0], feats[1])
    f0, f1 = features_from_images([img0, img1], depth=2, on_device=True)
    return _match_hip(f0, f1)


def img_downscale(img, downscale, be=None):
    """sfm.py:35-42: `int(downscale / 2)` successive cv2.pyrDown calls (the reference's downscale = 2 halves once)."""
    cv = _be(be).cv
    for _ in range(int(downscale / 2)):
        img = cv.pyrDown(img)
    return img


def features_from_images(images, depth=3, on_device=False):
    """The detector half of find_features (sfm.py:243-252) for a whole sequence: BGR uint8 frames -> the `features` list
    run_sfm takes, [(kp (n, 2) float32 pixel coordinates in cv2's keypoint order, des (n, 128) float32)].
    Frames are uploaded once and go through cvtColor + SIFT on `depth` streams; one device->host copy per frame, or
    none with on_device=True (the features stay in HBM as torch tensors, which the matcher takes as they are)."""
    from . import sift as _sift
    dev = torch.device("cuda")
    pipes, pending, feats = {}, [], []

    def collect():
        st, eng = pending.pop(0)
        st.synchronize()
        n = eng.check_capacity()
        if on_device:
            feats.append((eng.keypoints[:n, :2].contiguous(), eng.descriptors[:n].clone()))
        else:
            feats.append((eng.keypoints[:n, :2].cpu().numpy(), eng.descriptors[:n].cpu().numpy()))

    try:
        for img in images:
            if not torch.is_tensor(img):
                img = np.ascontiguousarray(img, np.uint8)
            h, w = img.shape[:2]
            pipe = pipes.get((w, h))
            if pipe is None:
                pipe = pipes[(w, h)] = _sift_pipeline(w, h, dev, depth)
            if len(pending) == depth:
                collect()
            d = img if torch.is_tensor(img) else torch.as_tensor(img).to(dev)      # (a frame already in HBM — run_sfm_images' downscaled ones — is taken as it is)
            gray = _sift.bgr2gray(d) if d.dim() == 3 else d
            _, st, eng = pipe.submit(gray)
            gray.record_stream(st)
            pending.append((st, eng))
        while pending:
            collect()
    finally:
        #  also when collect() / check_capacity / the device raised: a cached pipeline left BUSY would make every later
        # call of this frame size build a private one — 3 x 268 MB and a device-draining stream probe in the middle of a job
        for pipe in pipes.values():
            _sift_pipeline_done(pipe)
    return feats


_SIFT_PIPES = {}      # (w, h, depth, device index) -> sift.SiftPipeline: workspaces (3 x 268 MB at 968 x 648) and streams are kept between runs —
#                       new streams every run mean cold per-stream allocator pools, i.e. hipMalloc calls (device-wide waits) in the middle of the job
_SIFT_PIPES_BUSY = set()
_SIFT_PIPES_LOCK = __import__("threading").Lock()


def _sift_pipeline(w, h, dev, depth):
    """A SiftPipeline for this frame size, taken from the cache when nobody else holds it (give it back with _sift_pipeline_done);
    a second concurrent user gets a private, uncached one."""
    from . import sift as _sift
    key = (int(w), int(h), int(depth), torch.device(dev).index)
    with _SIFT_PIPES_LOCK:
        pipe = _SIFT_PIPES.get(key)
        if pipe is not None and id(pipe) not in _SIFT_PIPES_BUSY:
            _SIFT_PIPES_BUSY.add(id(pipe))
            return pipe
    with _SIFT_PIPES_LOCK:
        js = _SIFT_PIPES.get(("job streams", torch.device(dev).index, int(depth)))
    # the cached pipeline of a size runs on the device's probed feature streams when they exist (_job_streams); a second,
    # concurrent user's private pipeline gets streams of its own
    # (a private pipeline is built while its size's cached one is busy, possibly from a producer thread in the middle of a job: plain
    #  streams, no probe — ops.shared_streams(probe=False))
    fresh = _sift.SiftPipeline(w, h, dev, depth=depth,
                               streams=js[0] if js is not None and pipe is None else (ops.shared_streams("sift", depth, dev, probe=False) if pipe is not None else None))
    with _SIFT_PIPES_LOCK:
        if pipe is None:                                  # first of its size: cache it (a handful of frame sizes at most)
            for old in [k for k in _SIFT_PIPES if isinstance(k[0], int) and id(_SIFT_PIPES[k]) not in _SIFT_PIPES_BUSY][:max(0, len(_SIFT_PIPES) - 5)]:
                _SIFT_PIPES.pop(old)
            _SIFT_PIPES[key] = fresh
        _SIFT_PIPES_BUSY.add(id(fresh))
    return fresh


def _sift_pipeline_done(pipe):
    with _SIFT_PIPES_LOCK:
        _SIFT_PIPES_BUSY.discard(id(pipe))


def _job_streams(dev, depth):
    """(feature streams [depth], chain stream) of this device, created ONCE and probed to reach different hardware queues
    (ops.independent_streams).  HIP deals streams onto its few hardware queues in creation order and two streams on one queue run
    one after the other: created late in a process that already made dozens of streams, the four streams of a from-pixels job
    collide about every other time (the job then takes 78 instead of 63 ms: the chain's kernels queue behind a frame's).
    Call it from the thread that owns the device context before any producer thread starts (the probe drains the device)."""
    dev = torch.device(dev)
    key = ("job streams", dev.index, int(depth))
    with _SIFT_PIPES_LOCK:
        js = _SIFT_PIPES.get(key)
    if js is None:
        # the chain's kernels are tiny and each is waited for by the host: on a HIGH-priority stream they are dispatched ahead
        # of the feature streams' queued workgroups instead of behind them
        chain = ops.independent_streams(1, dev, priority=-1)[0]
        feat = ops.independent_streams(int(depth), dev, avoid=[chain])
        ops.release_probe_scratch(dev)
        js = (feat, chain)
        with _SIFT_PIPES_LOCK:
            js = _SIFT_PIPES.setdefault(key, js)
    return js


class FeatureStream:
    """sfm.py:301-302 (read + img_downscale) and :243-252 (cvtColor + SIFT) AHEAD of the driver.

    The driver is sequential and host-bound — RANSAC hypotheses, Levenberg-Marquardt algebra, a dozen host waits per camera —
    and leaves the GPU mostly idle; feature extraction is the opposite.  Round 4 ran them one after the other (all frames' SIFT,
    then the chain).  Here a producer THREAD uploads frame k, halves it, and runs cvtColor + SIFT on its own streams (`depth`
    frames in flight, at most `lookahead` frames ahead of the consumer); `stream[i]` blocks until frame i's (keypoints (n, 2),
    descriptors (n, 128)) are in HBM.  Same kernels on the same pixels: results are bit-identical to the serial order.
    The library calls release the GIL (ctypes), torch releases it around copies and launches."""

    lazy = True      # (_HipEngine: no up-front pass over every frame's shape)

    def __init__(self, images, downscale=2, depth=3, lookahead=8):
        import threading
        self.images, self.downscale, self.depth, self.lookahead = images, downscale, int(depth), max(int(lookahead), int(depth) + 1)
        self.dev = torch.device("cuda", torch.cuda.current_device())
        self.small = [None] * len(images)             # the halved frames (HBM): colour lookup at the end of the run
        self._res, self._err, self._wanted, self._stop = {}, None, 1, False
        self._cv = threading.Condition()
        self.chain_stream = _job_streams(self.dev, self.depth)[1]     # (made here, in the caller's thread: the probe drains the device)
        self._thread = threading.Thread(target=self._produce, name="sfm-feature-stream", daemon=True)
        self._thread.start()

    def __len__(self):
        return len(self.images)

    def __getitem__(self, i):
        if i < 0:
            i += len(self.images)
        with self._cv:
            if i > self._wanted:
                self._wanted = i
                self._cv.notify_all()
            while i not in self._res and self._err is None:
                self._cv.wait()
            if i not in self._res:
                raise self._err
            return self._res[i]

    def __iter__(self):
        return (self[i] for i in range(len(self.images)))

    def close(self):
        """Stop producing (a consumer that gives up mid-sequence must not leave the producer waiting for it) and join the thread."""
        with self._cv:
            self._stop = True
            self._cv.notify_all()
        self._thread.join()

    def _produce(self):
        from . import sift as _sift
        pipes, pending = {}, []                       # pending: (frame, stream, engine) in submission order
        try:
            torch.cuda.set_device(self.dev)

            def collect():
                k, st, eng = pending.pop(0)
                st.synchronize()
                n = eng.check_capacity()
                with torch.cuda.stream(st):
                    kp, des = eng.keypoints[:n, :2].contiguous(), eng.descriptors[:n].clone()
                st.synchronize()
                with self._cv:
                    self._res[k] = (kp, des)
                    self._cv.notify_all()

            for k, im in enumerate(self.images):
                with self._cv:                        # stay at most `lookahead` frames ahead of what the driver has asked for
                    while k > self._wanted + self.lookahead and not self._stop:
                        self._cv.wait()
                    if self._stop:
                        break
                h, w = im.shape[0], im.shape[1]
                for _ in range(int(self.downscale / 2)):
                    h, w = (h + 1) // 2, (w + 1) // 2
                pipe = pipes.get((w, h))
                if pipe is None:
                    pipe = pipes[(w, h)] = [_sift_pipeline(w, h, self.dev, self.depth), 0]
                while len(pending) >= self.depth or any(e is pipe[0].engines[pipe[1] % self.depth] for _, _, e in pending):
                    collect()
                slot = pipe[1] % self.depth
                pipe[1] += 1
                st, eng = pipe[0].streams[slot], pipe[0].engines[slot]
                with torch.cuda.stream(st):
                    d = im.to(self.dev) if torch.is_tensor(im) else torch.as_tensor(np.ascontiguousarray(im, np.uint8)).to(self.dev)
                    for _ in range(int(self.downscale / 2)):
                        d = _sift.pyrdown(d)
                    self.small[k] = d
                    eng.launch(_sift.bgr2gray(d) if d.dim() == 3 else d)
                pending.append((k, st, eng))
            while pending:
                collect()
        except BaseException as e:      # noqa: BLE001 — handed to the consumer
            with self._cv:
                self._err = e
                self._cv.notify_all()
        finally:
            for pipe in pipes.values():
                try:
                    pipe[0].synchronize()             # nothing of this thread's is in flight when the pipelines go back
                except Exception:      # noqa: BLE001
                    pass
                _sift_pipeline_done(pipe[0])
            with self._cv:
                if self._err is None and len(self._res) < len(self.images):
                    self._err = ops.SfmHipError("FeatureStream was closed before every frame was produced")
                self._cv.notify_all()


def run_sfm_images(images, K, downscale=2, log=None, be=None, bundle_adjustment=False, gtol_thresh=0.5, profile=None):
    """sfm.py's main loop from pixels: img_downscale (:40), cvtColor + SIFT (:243-252) and the driver (:274-423).
    `images`: BGR uint8 frames in sequence order; K is scaled by the caller as in sfm.py:20-26.
    profile: a DriverProfile — the run is then a PROFILED one (the device is drained at every stage boundary)."""
    import time
    if be is None and profile is None:
        # the product path: features are produced AHEAD of the sequential driver by a second host thread (FeatureStream)
        feats = FeatureStream(images, downscale)
        try:
            # the chain runs on the device's probed high-priority stream, beside the feature streams (_job_streams)
            cur = torch.cuda.current_stream()
            hp = feats.chain_stream
            hp.wait_stream(cur)
            with torch.cuda.stream(hp):
                out = run_sfm(feats, K, images=feats.small, log=log, bundle_adjustment=bundle_adjustment, gtol_thresh=gtol_thresh)
            cur.wait_stream(hp)
            out["features"] = list(feats)
        finally:
            feats.close()
        return out
    # a PROFILED run (or a substituted backend) keeps the stages apart: every frame's preprocessing, then every frame's SIFT, then the chain
    import time
    t_all = t0 = time.perf_counter()
    if be is None:
        # HIP path: a frame crosses PCIe once (the full-size upload); the halved frame stays in HBM for cvtColor + SIFT and for
        # the colour lookup at the end of the run (round 4 downloaded every halved frame and uploaded it again for SIFT)
        from . import sift as _sift
        dev = torch.device("cuda")
        small = []
        for im in images:
            d = im.to(dev) if torch.is_tensor(im) else torch.as_tensor(np.ascontiguousarray(im, np.uint8)).to(dev)
            for _ in range(int(downscale / 2)):
                d = _sift.pyrdown(d)
            small.append(d)
    else:
        small = [img_downscale(im, downscale, be) for im in images]
    if profile is not None:
        torch.cuda.synchronize()
        profile.add("img_downscale (sfm.py:40: upload + pyrDown)", time.perf_counter() - t0, len(images))
        t0 = time.perf_counter()
    feats = features_from_images(small, on_device=be is None)
    if profile is not None:
        torch.cuda.synchronize()
        profile.add("cvtColor + SIFT detectAndCompute (sfm.py:243-252)", time.perf_counter() - t0, len(images))
    out = run_sfm(feats, K, images=small, log=log, be=be, bundle_adjustment=bundle_adjustment, gtol_thresh=gtol_thresh, profile=profile)
    if profile is not None:
        torch.cuda.synchronize()
        profile.wall = time.perf_counter() - t_all
    out["features"] = feats
    return out


class DriverProfile:
    """Where a driver run's wall time goes, stage by stage, and how often the host WAITS for the device.

    A profiled run drains the device at every stage boundary (torch.cuda.synchronize() before and after each engine call), so
    its stages add up to more than the free-running time: a breakdown, not the pipeline's speed.  `host_syncs` counts, over
    the driver loop only, the library-side waits (sfm_host_sync_count: the RANSAC entry points read hypothesis scores back
    chunk by chunk) and the Python-side ones (.item() / .cpu() / .tolist() / .numpy() of a device tensor, synchronize());
    the drains the profiler itself adds are not counted."""

    def __init__(self):
        self.seconds, self.calls = {}, {}
        self.host_syncs_python = self.host_syncs_library = 0
        self.pnp_host_us = None
        self.wall = None

    def add(self, stage, dt, n=1):
        self.seconds[stage] = self.seconds.get(stage, 0.0) + dt
        self.calls[stage] = self.calls.get(stage, 0) + n

    def report(self, cameras):
        tot = sum(self.seconds.values())
        return {"stage_ms": {k: round(v * 1e3, 3) for k, v in sorted(self.seconds.items(), key=lambda kv: -kv[1])},
                "stage_calls": dict(self.calls), "profiled_total_ms": round(tot * 1e3, 3),
                "profiled_wall_ms": None if self.wall is None else round(self.wall * 1e3, 3),
                "between_stages_ms": None if self.wall is None else round((self.wall - tot) * 1e3, 3),
                "between_stages_note": "Python between the operator calls (pose algebra, state hand-over, the final cloud / colour gather and their downloads) + the profiler's own drains",
                "host_syncs": {"python_side": self.host_syncs_python, "library_side": self.host_syncs_library,
                               "per_registered_camera": round((self.host_syncs_python + self.host_syncs_library) / max(cameras, 1), 2)},
                "solve_pnp_ransac_host_us_per_call": self.pnp_host_us}


class _count_python_syncs:
    """Context manager: counts Tensor.item / .cpu / .tolist / .numpy on DEVICE tensors and the synchronize() calls while active
    (diagnostics for DriverProfile; the patched methods call straight through)."""

    def __init__(self, prof):
        self.prof, self.saved = prof, []

    def __enter__(self):
        prof = self.prof

        def wrap_tensor(name):
            orig = getattr(torch.Tensor, name)

            def f(t, *a, **k):
                if t.is_cuda and not prof._muted:
                    prof.host_syncs_python += 1
                return orig(t, *a, **k)
            self.saved.append((torch.Tensor, name, orig))
            setattr(torch.Tensor, name, f)

        def wrap_fn(obj, name):
            orig = getattr(obj, name)

            def f(*a, **k):
                if not prof._muted:
                    prof.host_syncs_python += 1
                return orig(*a, **k)
            self.saved.append((obj, name, orig))
            setattr(obj, name, f)
        prof._muted = False
        for n in ("item", "cpu", "tolist", "numpy"):
            wrap_tensor(n)
        wrap_fn(torch.cuda, "synchronize")
        wrap_fn(torch.cuda.Stream, "synchronize")
        wrap_fn(torch.cuda.Event, "synchronize")
        return self

    def __exit__(self, *exc):
        for obj, name, orig in reversed(self.saved):
            setattr(obj, name, orig)
        return False


class _ProfiledEngine:
    """An engine whose operator calls are timed one by one (device drained before and after each)."""
    _STAGES = {"match": "knnMatch + ratio + gather (sfm.py:259-268)", "essential": "findEssentialMat (sfm.py:307)",
               "recover_pose": "recoverPose (sfm.py:311)", "triangulate": "triangulatePoints (sfm.py:53-54)",
               "error": "ReprojectionError (sfm.py:79-100)", "pnp": "solvePnPRansac + inlier gathers (sfm.py:67-76)",
               "associate": "common_points (sfm.py:215-239)", "take": "row gathers", "errors": "error download"}

    def __init__(self, eng, prof):
        self._eng, self._prof = eng, prof

    def __getattr__(self, name):
        attr = getattr(self._eng, name)
        stage = self._STAGES.get(name)
        if stage is None or not callable(attr):
            return attr
        prof = self._prof

        def timed(*a, **k):
            import time
            prof._muted = True
            torch.cuda.synchronize()
            prof._muted = False
            t0 = time.perf_counter()
            r = attr(*a, **k)
            prof._muted = True
            torch.cuda.synchronize()
            prof._muted = False
            prof.add(stage, time.perf_counter() - t0)
            return r
        return timed


def Triangulation(P1, P2, pts1, pts2, K, repeat, be=None):
    """sfm.py:45-56."""
    cv2 = _be(be).cv
    if not repeat:
        points1 = np.transpose(pts1)
        points2 = np.transpose(pts2)
    else:
        points1, points2 = pts1, pts2
    cloud = cv2.triangulatePoints(P1, P2, points1, points2)
    cloud = cloud / cloud[3]                                       # float32 division, sfm.py:54
    return points1, points2, cloud


def PnP(X, p, K, d, p_0, initial, be=None):
    """sfm.py:60-76."""
    cv2 = _be(be).cv
    if initial == 1:
        X = X[:, 0, :]
        p = p.T
        p_0 = p_0.T
    ret, rvecs, t, inliers = cv2.solvePnPRansac(X, p, K, d, cv2.SOLVEPNP_ITERATIVE)   # quirk 1: 5th positional = rvec
    R, _ = cv2.Rodrigues(rvecs)
    if inliers is not None:
        p = p[inliers[:, 0]]
        X = X[inliers[:, 0]]
        p_0 = p_0[inliers[:, 0]]
    return R, t, p, X, p_0


def _reproj_hip(r, t, K, Xf, obs):
    """Fused projection + squared-error sum on the device → (sumsq, projected float32 (N,2))."""
    dev = torch.device("cuda")
    cams = torch.as_tensor(np.hstack([r.ravel(), np.asarray(t, np.float64).ravel()])[None]).to(dev)
    out = ops.project_residual(cams, K, torch.as_tensor(Xf).to(dev), torch.as_tensor(np.ascontiguousarray(obs)).to(dev))
    return float(out["sumsq"].item()), out["proj"].cpu().numpy()


def ReprojectionError(X, pts, Rt, K, homogenity, be=None):
    """sfm.py:79-100: ||float32(proj) - float32(pts)||_F / N in ONE fused sweep on the device."""
    b = _be(be)
    cv2 = b.cv
    R = Rt[:3, :3]
    t = Rt[:3, 3]
    r, _ = cv2.Rodrigues(R)
    if homogenity == 1:
        X = cv2.convertPointsFromHomogeneous(X.T)
    obs = np.float32(pts.T if homogenity == 1 else pts)
    if np.asarray(X).dtype == np.float64:
        # float64 object points (the bundle-adjusted cloud, sfm.py:384): cv2.projectPoints works in double and returns double;
        # only then are p and pts cast to float32 (sfm.py:90-91) — a float32 round trip of X would be as large as the
        # "Minimized error" itself
        Xd = np.ascontiguousarray(np.asarray(X, np.float64).reshape(-1, 3))
        if b.reproj is _reproj_hip:
            p64 = ops.project_points_f64(r, t, K, torch.as_tensor(Xd).to("cuda")).cpu().numpy()
        else:
            p64, _ = cv2.projectPoints(Xd, r, t, K, distCoeffs=None)
        p = np.float32(np.asarray(p64, np.float64).reshape(-1, 2))
        d = p.astype(np.float64) - obs.reshape(-1, 2).astype(np.float64)             # cv2.norm(float32, float32, NORM_L2): double accumulation
        return float(np.sqrt(np.sum(d * d))) / len(p), X, p
    Xf = np.ascontiguousarray(np.asarray(X, np.float32).reshape(-1, 3))
    sumsq, p = b.reproj(r, t, K, Xf, obs)
    tot_error = float(np.sqrt(sumsq)) / len(p)                     # quirk 3: Frobenius norm / N
    return tot_error, X, p


def OptimReprojectionError(x, be=None):
    """sfm.py:104-136: residual vector of the reference's bundle adjustment.  x = [Rt 12 | K 9 | p 2N | X 3N], the 2-D
    block is found with the reference's `int(rest * 0.4)` split, the residual is (p - proj)^2 / N flattened (2N,).
    The projection runs on the device; the print() of sfm.py:132 is dropped."""
    b = _be(be)
    cv2 = b.cv
    x = np.asarray(x, np.float64)
    Rt = x[0:12].reshape((3, 4))
    K = x[12:21].reshape((3, 3))
    rest = int(len(x[21:]) * 0.4)
    p = x[21:21 + rest].reshape((2, int(rest / 2))).T
    X = x[21 + rest:].reshape((int(len(x[21 + rest:]) / 3), 3))
    r, _ = cv2.Rodrigues(Rt[:3, :3])
    # fp64 end to end, like the reference's cv2.projectPoints on float64 points: SciPy's 2-point finite differences
    # (relative step ~1.5e-8) would vanish in a float32 round trip of X or of the projection
    if b.reproj is _reproj_hip:
        p2d = ops.project_points_f64(r, Rt[:3, 3], K, torch.as_tensor(np.ascontiguousarray(X, np.float64)).to("cuda")).cpu().numpy()
    else:
        p2d, _ = cv2.projectPoints(np.ascontiguousarray(X, np.float64), r, Rt[:3, 3], K, distCoeffs=None)
        p2d = np.asarray(p2d, np.float64).reshape(-1, 2)
    num_pts = len(p)
    return (((p - p2d) ** 2).ravel()) / num_pts


def BundleAdjustment(points_3d, temp2, Rtnew, K, r_error, be=None):
    """sfm.py:138-157 (disabled by default in the reference, sfm.py:33): SciPy's trust-region least squares with its
    default 2-point finite-difference Jacobian over [Rt | K | 2-D points | 3-D points], residuals from the device.
    Returns X (N,3), p (N,2), Rt (3,4) exactly as the reference unpacks them."""
    from scipy.optimize import least_squares
    opt_variables = np.hstack((np.asarray(Rtnew, np.float64).ravel(), np.asarray(K, np.float64).ravel()))
    opt_variables = np.hstack((opt_variables, np.asarray(temp2, np.float64).ravel()))
    opt_variables = np.hstack((opt_variables, np.asarray(points_3d, np.float64).ravel()))
    res = least_squares(fun=lambda v: OptimReprojectionError(v, be), x0=opt_variables, gtol=r_error)
    cv = res.x
    Rt = cv[0:12].reshape((3, 4))
    rest = int(len(cv[21:]) * 0.4)
    p = cv[21:21 + rest].reshape((2, int(rest / 2))).T
    X = cv[21 + rest:].reshape((int(len(cv[21 + rest:]) / 3), 3))
    return X, p, Rt


def common_points(pts1, pts2, pts3):
    """sfm.py:215-239.  quirk 2: a row of pts2 "equals" pts1[i] when x OR y is bit-equal; first hit wins;
    duplicates in indx2 allowed; the complement is mask-and-compress of pts2 / pts3.  Runs on the device
    (sfm_common_points) when a GPU is present; the NumPy form below is the same definition for host-only use."""
    pts1 = np.asarray(pts1)
    pts2 = np.asarray(pts2)
    pts3 = np.asarray(pts3)
    if torch.cuda.is_available() and pts1.dtype == np.float32 and pts2.dtype == np.float32 and len(pts1) and len(pts2):
        dev = torch.device("cuda")
        i1, i2, keep = ops.common_points(torch.from_numpy(np.ascontiguousarray(pts1)).to(dev),
                                         torch.from_numpy(np.ascontiguousarray(pts2)).to(dev))
        keep = keep.cpu().numpy()
        return (i1.cpu().numpy().astype(np.int64), i2.cpu().numpy().astype(np.int64), pts2[keep].reshape(-1, 2),
                pts3[keep].reshape(-1, 2))
    hit = (pts2[None, :, 0] == pts1[:, None, 0]) | (pts2[None, :, 1] == pts1[:, None, 1])
    any_hit = hit.any(1)
    indx1 = np.flatnonzero(any_hit)
    indx2 = hit.argmax(1)[any_hit]
    keep = np.ones(len(pts2), bool)
    keep[indx2] = False
    return indx1, indx2, pts2[keep].reshape(-1, 2), pts3[keep].reshape(-1, 2)


def to_ply(path, point_cloud, colors, densify=False):
    """sfm.py:169-201 byte-for-byte: x200, keep dist < mean(dist)+300 about the centroid, tab-indented header
    whose last line also indents the first vertex (quirk 9), colour columns written as B G R."""
    pts = point_cloud.reshape(-1, 3) * 200
    verts = np.hstack([pts, colors.reshape(-1, 3)])
    centred = verts[:, :3] - np.mean(verts[:, :3], axis=0)
    dist = np.sqrt(centred[:, 0] ** 2 + centred[:, 1] ** 2 + centred[:, 2] ** 2)
    verts = verts[np.where(dist < np.mean(dist) + 300)]
    lines = ["format ascii 1.0", "element vertex %d" % len(verts), "property float x", "property float y",
             "property float z", "property uchar blue", "property uchar green", "property uchar red", "end_header"]
    name = "dense.ply" if densify else "sparse.ply"
    with open(path + "/Point_Cloud/" + name, "w") as f:
        f.write("ply\n" + "".join("\t\t" + ln + "\n" for ln in lines) + "\t\t")
        np.savetxt(f, verts, "%f %f %f %d %d %d")
    return len(verts)


# ---------------------------------------------------------------------------------------------------------------------
# The incremental driver (sfm.py:274-423, bundle_adjustment=False: the reference's default), written ONCE over a small
# engine interface.  An engine owns the array type the per-frame state lives in and supplies the operators:
#     match(i, j)            matched keypoints of images i, j                       (find_features, sfm.py:347)
#     essential(a, b)        essential matrix + rows of its {0,1} mask              (sfm.py:307-309)
#     recover_pose(E, a, b)  R, t + rows of the {0,255} cheirality mask             (sfm.py:311-313)
#     triangulate(Pa, Pb, a, b)   (n,3) float32 cloud, w normalised as sfm.py:54    (Triangulation + convertPointsFromHomogeneous)
#     error(X, obs, Rt)      handle of ||proj - obs||_F / n                         (ReprojectionError)
#     pnp(X, p, p0)          R, t and the inlier rows of p, X, p0                   (PnP)
#     associate(pts1, pts_)  indx1, indx2, rows of pts_ not associated              (common_points)
#     take(x, rows), host(x), errors(handles)
# `_HipEngine` keeps everything in HBM (device tensors from kernel to kernel, a handful of host synchronisations per
# frame); `_ArrayEngine` runs the NumPy helper mirrors above over a Backend — the HIP one, or the CPU twin the tests build
# on the oracle.  One frame is one call of register_next(): the tests drive it frame by frame with the other engine's
# state (teacher forcing) to separate per-frame parity from the drift of the free-running chain.
# ---------------------------------------------------------------------------------------------------------------------
class FrameState:
    """What the driver carries from frame to frame (sfm.py:404-409: only the last two cameras, quirk 7): the two projection
    matrices (host, float64), the matches of the last pair in the engine's array type, and — for the very first
    registration only — the bootstrap pair's PnP-filtered cloud and points (afterwards the cloud is re-triangulated from
    ALL ratio matches, quirk 6)."""
    __slots__ = ("P1", "P2", "pts0", "pts1", "cloud0")

    def __init__(self, P1, P2, pts0, pts1, cloud0=None):
        self.P1, self.P2, self.pts0, self.pts1, self.cloud0 = P1, P2, pts0, pts1, cloud0

    def on(self, engine):
        """The same state in another engine's array type."""
        c = engine.array
        return FrameState(self.P1.copy(), self.P2.copy(), c(self.pts0), c(self.pts1), None if self.cloud0 is None else c(self.cloud0))


def _np(x):
    return x.detach().cpu().numpy() if torch.is_tensor(x) else np.asarray(x)


class _ArrayEngine:
    """NumPy arrays in, NumPy arrays out, through the helper mirrors (Triangulation, PnP, ReprojectionError, common_points)
    of a Backend."""

    def __init__(self, features, K, be=None):
        self.features, self.K, self.be = features, np.asarray(K, np.float64), _be(be)

    def array(self, x):
        return np.ascontiguousarray(_np(x))

    host = staticmethod(_np)

    def match(self, i, j):
        return match_features(self.features[i], self.features[j], self.be)

    def essential(self, a, b):
        cv = self.be.cv
        E, mask = cv.findEssentialMat(a, b, self.K, method=cv.RANSAC, prob=0.999, threshold=0.4, mask=None)
        return E, np.flatnonzero(mask.ravel() == 1)              # quirk 4: {0,1} mask

    def recover_pose(self, E, a, b):
        _, R, t, mask = self.be.cv.recoverPose(E, a, b, self.K)
        return R, t, np.flatnonzero(mask.ravel() > 0)            # quirk 4: {0,255} mask

    def take(self, x, rows):
        return x[rows]

    def triangulate(self, Pa, Pb, a, b):
        _, _, cloud = Triangulation(Pa, Pb, a, b, self.K, repeat=False, be=self.be)
        return self.be.cv.convertPointsFromHomogeneous(cloud.T)[:, 0, :]

    def error(self, X, obs, Rt):
        return ReprojectionError(X, obs, Rt, self.K, homogenity=0, be=self.be)[0]

    def pnp(self, X, p, p0):
        return PnP(X, p, self.K, np.zeros((5, 1), dtype=np.float32), p0, initial=0, be=self.be)

    def associate(self, pts1, pts_):
        indx1, indx2, _, _ = common_points(pts1, pts_, pts_)
        keep = np.ones(len(pts_), bool)
        keep[indx2] = False
        return indx1, indx2, np.flatnonzero(keep)

    def errors(self, handles):
        return [float(h) for h in handles]


class _HipEngine:
    """Device tensors from kernel to kernel.  What crosses to the host per frame: one survivor count, one association count
    and the few scalars solvePnPRansac returns; error sums, clouds and colour-lookup points come back once, at the end."""

    def __init__(self, features, K):
        self.features, self.K = features, np.asarray(K, np.float64)
        self.dev = torch.device("cuda")
        self.cache, self.pre, self.pre_m = {}, {}, []
        # The matches of consecutive frames do not depend on the pose chain: pairs of equal shape are matched up front, 8 per
        # launch set (sfm_match_batch_l2_f32: bit-identical to per-pair calls), their survivor counts come back in ONE
        # download; pairs of a shape that occurs once are matched when the driver asks for them.
        shapes = {}
        for k in range(len(features) - 1 if not getattr(features, "lazy", False) else 0):     # (a FeatureStream is still being produced: frames of a real sequence never share a shape anyway)
            shapes.setdefault((len(features[k][0]), len(features[k + 1][0])), []).append(k)
        self.keep_all = set()
        for plist in shapes.values():
            if len(plist) >= 2:
                self.keep_all.update(plist)
                self.keep_all.update(k + 1 for k in plist)
        counts = []
        for (nq, nt), plist in shapes.items():
            if len(plist) < 2 or nq == 0 or nt == 0:
                continue
            bm = ops.BatchMatcher(nq, nt, self.dev, ratio=RATIO, batch=min(8, len(plist)))
            for c0 in range(0, len(plist), 8):
                chunk = plist[c0:c0 + 8]
                bm.run([(self._feat(k)[1], self._feat(k + 1)[1]) for k in chunk])
                oq, ot, cn = bm.out_q[:len(chunk)].clone(), bm.out_t[:len(chunk)].clone(), bm.count[:len(chunk)].clone()
                for b, k in enumerate(chunk):
                    self.pre[k] = (oq[b], ot[b], cn[b], len(counts) + b)
                counts.extend([cn[b] for b in range(len(chunk))])
        self.pre_m = torch.cat(counts).cpu().tolist() if counts else []

    def array(self, x):
        return x.to(self.dev).contiguous() if torch.is_tensor(x) else torch.as_tensor(np.ascontiguousarray(x)).to(self.dev)

    host = staticmethod(_np)

    def _feat(self, i):   # every image's features are uploaded once
        if i not in self.cache:
            kp, des = self.features[i]
            up = lambda a: a.to(self.dev, torch.float32).contiguous() if torch.is_tensor(a) else torch.as_tensor(np.ascontiguousarray(a, np.float32)).to(self.dev)
            self.cache[i] = (up(kp), up(des))
            if i - 2 not in self.keep_all:
                self.cache.pop(i - 2, None)
        return self.cache[i]

    def match(self, i, j):
        (kp0, d0), (kp1, d1) = self._feat(i), self._feat(j)
        if i in self.pre and j == i + 1:
            out_q, out_t, count, slot = self.pre[i]
            m = int(self.pre_m[slot])
        else:
            idx, dist = ops.knn2(d0, d1)
            out_q, out_t, count = ops.ratio_compact(idx, dist, RATIO)
            m = int(count.item())
        p0, p1 = ops.gather_matches(kp0, kp1, out_q, out_t, count)
        return p0[:m], p1[:m]

    def essential(self, a, b):
        from . import ransac
        E, mask = ransac.find_essential_mat(a, b, self.K, 0.999, 0.4, return_device_mask=True)
        return E, ops.mask_indices(mask)                           # quirk 4: rows of the {0,1} mask

    def recover_pose(self, E, a, b):
        from . import ransac
        _, R, t, mask = ransac.recover_pose(E, a, b, self.K, return_device_mask=True)
        return R, t, ops.mask_indices(mask, nonzero=True)          # quirk 4: rows of the {0,255} mask

    def take(self, x, rows):
        if torch.is_tensor(rows) and rows.dtype in (torch.int32, torch.int64) and rows.dim() == 1:
            return x.index_select(0, rows)                         # one launch (x[rows.long()] is two: the index conversion, then the gather)
        return x[rows.long() if torch.is_tensor(rows) else rows]

    def triangulate(self, Pa, Pb, a, b):
        nw = "guarded" if cv2.TRIANGULATE_ROWS == 4 else True      # (bit-identical to the faithful path, ~8x faster)
        return ops.triangulate(Pa, Pb, a.t(), b.t(), rows=cv2.TRIANGULATE_ROWS, normalise_w=nw)[:3].t().contiguous()

    def error(self, X, obs, Rt):
        from . import hostgeom as hg
        cams = torch.as_tensor(np.hstack([hg.rodrigues_mat2vec(Rt[:3, :3]), Rt[:3, 3]])[None]).to(self.dev)
        return ops.project_residual(cams, self.K, X, obs.contiguous(), want_proj=False)["sumsq"], len(obs)

    lazy_pnp = True       # register_next may ask for the pose and the inlier COUNT only (gather=False)

    def pnp(self, X, p, p0, gather=True):
        """sfm.py:60-76.  gather=False: (R, t, inliers (k, 1) int32 in HBM, None, None) — sfm.py:73-75 gathers p, X and p_0 by the
        inliers, but the reference's loop (sfm.py:341-409) reads none of the three after the call: the driver takes the pose and the
        inlier count, and the three gathers (and the p_0 argument's own gather) are not launched.  The bootstrap pair, whose filtered
        points survive (sfm.py:326-329), calls with gather=True."""
        from . import hostgeom as hg
        from . import ransac
        ok, rvec, t, inl = ransac.solve_pnp_ransac(X, p, self.K, return_device_inliers=True)
        R = hg.rodrigues_vec2mat(rvec)
        if not gather:
            return R, t, (inl if inl is not None else p), None, None
        if inl is not None:
            sel = inl[:, 0]
            p, X, p0 = p.index_select(0, sel), X.index_select(0, sel), p0.index_select(0, sel)
        return R, t, p, X, p0

    def associate(self, pts1, pts_):
        # association + complement list, ONE host read for both counts (round 5: two)
        indx1, indx2, keep, c1 = ops.common_points(pts1, pts_, raw=True)
        rest, c2 = ops.mask_indices(keep, nonzero=True, raw=True)
        m, k = torch.cat([c1, c2]).tolist()
        return indx1[:m], indx2[:m], rest[:k]

    def errors(self, handles):
        if not handles:
            return []
        dev_h = [h for h in handles if isinstance(h, tuple)]           # (a bundle-adjusted frame's error is already a float)
        sv = iter(torch.cat([h for h, _ in dev_h]).cpu().numpy() if dev_h else [])   # one download for the whole sequence
        return [float(np.sqrt(next(sv))) / h[1] if isinstance(h, tuple) else float(h) for h in handles]


def make_engine(features, K, be=None, device_resident=None):
    """The engine run_sfm would use: HBM-resident on the HIP back-end (be=None) unless device_resident=False, NumPy arrays
    over `be` otherwise."""
    if device_resident is None:
        device_resident = be is None
    if device_resident:
        if be is not None:
            raise ops.SfmHipError("run_sfm: device_resident needs the HIP back-end")
        return _HipEngine(features, K)
    return _ArrayEngine(features, K, be)


def bootstrap_pair(eng):
    """Images 0 and 1 (sfm.py:304-339): essential matrix, pose, first cloud, its error, and the PnP call whose inlier
    filtering of the points survives.  Returns (state, first error handle, P1, P2)."""
    K = eng.K
    Rt0 = np.hstack([np.eye(3), np.zeros((3, 1))])
    P1 = K @ Rt0
    a, b = eng.match(0, 1)
    E, rows = eng.essential(a, b)
    a, b = eng.take(a, rows), eng.take(b, rows)
    R, t, rows = eng.recover_pose(E, a, b)
    a, b = eng.take(a, rows), eng.take(b, rows)
    Rt1 = np.empty((3, 4))
    Rt1[:, :3] = R @ Rt0[:, :3]
    Rt1[:, 3] = Rt0[:, 3] + Rt0[:, :3] @ np.asarray(t, np.float64).ravel()
    P2 = K @ Rt1
    X = eng.triangulate(P1, P2, a, b)
    first = eng.error(X, b, Rt1)
    _, _, b_in, X_in, _ = eng.pnp(X, b, a)                      # (its pose is discarded, sfm.py:326)
    return FrameState(P1, P2, a, b_in, X_in), first


def register_next(eng, state, i, bundle_adjustment=False, gtol_thresh=0.5):
    """One iteration of sfm.py:341-409: register image i + 2 against the cloud of images i, i + 1.  Returns
    (next state, dict(P, error handle, cloud (n,3), lookup (n,2) points of the new cloud in image i + 2, pnp inlier count)).
    bundle_adjustment=True is the reference's `if bundle_adjustment:` branch (sfm.py:378-388, off by default, sfm.py:33):
    SciPy's least_squares over [Rt | K | 2-D points | 3-D points] of the NEW cloud with the device's fp64 projection as
    residual (BundleAdjustment above), then the refined Rt, points and observations replace the frame's — P for the next
    frame, the cloud, the colour-lookup points and the error (ReprojectionError with homogenity = 0 on the float64
    results, projected in float64 as sfm.py:384 does) — exactly where the reference replaces them.  The pose ARRAY keeps
    the camera as PnP found it: sfm.py:375 appends Pnew to posearr BEFORE the `if bundle_adjustment:` branch, so pose.csv
    holds the pre-adjustment matrices (out["P_pose"]); out["P"] is what the next frame triangulates with."""
    K = eng.K
    pts_, pts2 = eng.match(i + 1, i + 2)
    cloud = state.cloud0 if state.cloud0 is not None else eng.triangulate(state.P1, state.P2, state.pts0, state.pts1)
    indx1, indx2, rest = eng.associate(state.pts1, pts_)
    new1, new2 = eng.take(pts_, rest), eng.take(pts2, rest)
    if getattr(eng, "lazy_pnp", False):       # (HBM-resident engine: the filtered p / X / p_0 of sfm.py:73-75 are never read below)
        R, t, p_in, _, _ = eng.pnp(eng.take(cloud, indx1), eng.take(pts2, indx2), None, gather=False)
    else:
        R, t, p_in, _, _ = eng.pnp(eng.take(cloud, indx1), eng.take(pts2, indx2), eng.take(pts_, indx2))
    Rt = np.hstack((np.asarray(R, np.float64), np.asarray(t, np.float64).reshape(3, 1)))
    P = K @ Rt
    X = eng.triangulate(state.P2, P, new1, new2)
    out = dict(P=P, P_pose=P, error=eng.error(X, new2, Rt), cloud=X, lookup=new2, pnp_inliers=len(p_in))
    if bundle_adjustment:
        be = getattr(eng, "be", None)
        # sfm.py:380: points_3d is the (N,1,3) float32 cloud ReprojectionError(homogenity=1) handed back, temp2 the (2,N)
        # transposed view Triangulation returned — raveled, that is all x then all y, which is how sfm.py:110 unpacks them
        Xb, pb, Rt = BundleAdjustment(np.asarray(eng.host(X))[:, None, :], np.asarray(eng.host(new2)).T, Rt, K, gtol_thresh, be=be)
        P = K @ Rt                                                                   # sfm.py:381
        err, _, _ = ReprojectionError(Xb, pb, Rt, K, homogenity=0, be=be)            # sfm.py:384 ("Minimized error")
        out = dict(P=P, P_pose=out["P_pose"], error=err, cloud=Xb, lookup=pb, pnp_inliers=len(p_in), ba_error_before=out["error"])
    return FrameState(state.P2.copy(), P.copy(), pts_, pts2), out


def run_sfm(features, K, images=None, log=None, be=None, device_resident=None, bundle_adjustment=False, gtol_thresh=0.5, profile=None):
    """The reference's driver, sfm.py:274-423; `bundle_adjustment` / `gtol_thresh` are its globals of sfm.py:33,337 (default:
    no bundle adjustment, as shipped).
    features: list of (kp (n,2) float32, des (n,128) float32) per image, in sequence order.
    images:   optional list of HxWx3 uint8 arrays for the colour lookup (sfm.py:393-394).
    Returns dict(posearr (9+12*n_cam,), Xtot (m,3), colorstot (m,3), errors [per-frame], first_error).
    On the HIP back-end (be=None) the per-frame state stays in HBM (`_HipEngine`); device_resident=False, or a substituted
    backend (the tests' CPU twin), runs the same driver over NumPy arrays (`_ArrayEngine`): same operators, same results."""
    if profile is not None:
        return _run_sfm_profiled(features, K, images, log, be, device_resident, bundle_adjustment, gtol_thresh, profile)
    eng = make_engine(features, K, be, device_resident)
    return _drive(eng, features, images, log, bundle_adjustment, gtol_thresh)


def _run_sfm_profiled(features, K, images, log, be, device_resident, bundle_adjustment, gtol_thresh, profile):
    """run_sfm with every engine call timed (DriverProfile) and the host's waits for the device counted."""
    import ctypes
    import time
    from . import _lib
    L = _lib.lib()
    buf = (ctypes.c_double * 10)()
    L.sfm_pnp_profile_read(buf, 1)
    lib0 = int(L.sfm_host_sync_count())
    with _count_python_syncs(profile):
        profile._muted = True
        torch.cuda.synchronize()
        profile._muted = False
        t0 = time.perf_counter()
        eng = make_engine(features, K, be, device_resident)
        profile._muted = True
        torch.cuda.synchronize()
        profile._muted = False
        profile.add("engine set-up: uploads + the up-front batched knnMatch of equally shaped pairs", time.perf_counter() - t0)
        out = _drive(_ProfiledEngine(eng, profile), features, images, log, bundle_adjustment, gtol_thresh)
    profile.host_syncs_library = int(L.sfm_host_sync_count()) - lib0
    L.sfm_pnp_profile_read(buf, 0)
    calls = max(buf[0], 1.0)
    names = ("copy_in", "epnp_hypotheses_host", "device_scoring_and_wait", "mask_and_inlier_bookkeeping", "dlt_init_host", "lm_sweeps_device_and_wait", "lm_algebra_host")
    profile.pnp_host_us = {"calls": int(buf[0]), **{n: round(buf[1 + i] / calls, 2) for i, n in enumerate(names)},
                           "hypothesis_chunks_per_call": round(buf[8] / calls, 2), "lm_sweeps_per_call": round(buf[9] / calls, 2)}
    return out


def _drive(eng, features, images, log, bundle_adjustment, gtol_thresh):
    say = log or (lambda *a: None)
    state, first = bootstrap_pair(eng)
    poses = [eng.K.ravel(), state.P1.ravel(), state.P2.ravel()]
    handles, clouds, lookups = [first], [], []
    for i in range(len(features) - 2):
        state, out = register_next(eng, state, i, bundle_adjustment, gtol_thresh)
        poses.append(out["P_pose"].ravel())                                          # sfm.py:375: before the bundle-adjustment branch
        handles.append(out["error"])
        clouds.append(out["cloud"])
        lookups.append(out["lookup"])
        if log is not None:
            say("Reprojection Error: ", eng.errors([out["error"]])[0])
    errs = eng.errors(handles)
    Xtot = np.vstack([np.zeros((1, 3))] + [eng.host(c) for c in clouds])             # quirk 8: leading zero row
    cols = [np.zeros((1, 3))]
    if images is not None and len(lookups) and torch.is_tensor(images[0]) and all(torch.is_tensor(p) for p in lookups):
        # frames and lookup points both in HBM: gather on the device, ONE download (same truncation toward zero, quirk 10)
        picked = []
        for i, pts in enumerate(lookups):
            reg = pts.to(torch.int32).long()
            picked.append(images[i + 2][reg[:, 1], reg[:, 0]].reshape(-1, 3))
        cols.append(torch.cat(picked).cpu().numpy().astype(np.float64))
    else:
        for i, pts in enumerate(lookups):
            reg = np.array(eng.host(pts), dtype=np.int32)                            # quirk 10: truncation toward zero
            img = None if images is None else (images[i + 2].cpu().numpy() if torch.is_tensor(images[i + 2]) else images[i + 2])
            cols.append(img[reg[:, 1], reg[:, 0]].reshape(-1, 3) if img is not None else np.zeros((len(reg), 3)))
    return dict(posearr=np.hstack(poses), Xtot=Xtot, colorstot=np.vstack(cols), errors=errs[1:], first_error=errs[0])


def save_pose_csv(path, posearr):
    """sfm.py:423: one value per line, numpy's default '%.18e'."""
    np.savetxt(path, posearr, delimiter="\n")
