"""Seeded synthetic inputs shared by the CPU and GPU tests (SURVEY.md §8c/§8d fixtures)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def sift_like(rng, n):
    """Integer-valued 0..255 float32 descriptors with ||d|| ~ 512, like cv2 SIFT output."""
    d = np.abs(rng.standard_normal((n, 128))) ** 2
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d = np.minimum(d, 0.2)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return np.clip(np.rint(d * 512), 0, 255).astype(np.float32)


def planted_pair(rng, nq, nt, frac=0.3, noise=2.0):
    """SIFT-like query/train sets where `frac` of the queries have a planted noisy twin in train.
    Returns q, t, planted (array of (query, train) pairs)."""
    q, t = sift_like(rng, nq), sift_like(rng, nt)
    k = int(frac * min(nq, nt))
    qi = rng.permutation(nq)[:k]
    ti = rng.permutation(nt)[:k]
    t[ti] = np.clip(q[qi] + np.rint(rng.normal(0, noise, (k, 128))), 0, 255).astype(np.float32)
    return q, t, np.stack([qi, ti], 1)


def load_pose_csv():
    """K (3x3) and the 57 projection matrices of the reference's Gustav run (sfm.py:423 format)."""
    v = np.loadtxt(os.path.join(GOLDEN, "pose.csv"))
    K = v[:9].reshape(3, 3)
    P = v[9:].reshape(-1, 3, 4)
    return K, P


def sparse_points():
    """Subset of the reference's sparse.ply vertices in world units (file stores them x200)."""
    z = np.load(os.path.join(GOLDEN, "sparse_ply_subset.npz"))
    return z["verts"][:, :3] / 200.0


def project(P, X):
    x = P @ np.hstack([X, np.ones((len(X), 1))]).T
    return (x[:2] / x[2]).T, x[2]


def gustav_pair(k, n, sigma, seed):
    """Cameras k, k+1 of pose.csv looking at reference cloud points visible in both; pixel noise sigma.
    Returns K, P1, P2, X (n,3) float64 truth, x1, x2 (n,2) float32."""
    K, P = load_pose_csv()
    rng = np.random.default_rng(seed)
    X = sparse_points()
    x1, z1 = project(P[k], X)
    x2, z2 = project(P[k + 1], X)
    ok = (z1 > 0.5) & (z2 > 0.5) & (x1[:, 0] > 0) & (x1[:, 0] < 968) & (x1[:, 1] > 0) & (x1[:, 1] < 648) \
        & (x2[:, 0] > 0) & (x2[:, 0] < 968) & (x2[:, 1] > 0) & (x2[:, 1] < 648)
    sel = np.flatnonzero(ok)
    if len(sel) < n:   # widen with jittered copies so every pair yields n points
        extra = X[rng.choice(sel, n - len(sel))] + rng.normal(0, 0.02, (n - len(sel), 3))
        Xs = np.vstack([X[sel], extra])
    else:
        Xs = X[rng.permutation(sel)[:n]]
    x1, _ = project(P[k], Xs)
    x2, _ = project(P[k + 1], Xs)
    x1 = (x1 + rng.normal(0, sigma, x1.shape)).astype(np.float32)
    x2 = (x2 + rng.normal(0, sigma, x2.shape)).astype(np.float32)
    return K, P[k], P[k + 1], Xs, x1, x2


def decompose_P(K, P):
    """[R|t] = K^-1 P (valid for pose.csv: P = K [R|t])."""
    Rt = np.linalg.solve(K, P)
    return Rt[:, :3], Rt[:, 3]


def ring_cameras(ncam, radius=8.0):
    """Cameras on a ring looking at the origin → (ncam,6) rvec,tvec (config-4 geometry)."""
    from scipy.spatial.transform import Rotation
    cams = np.zeros((ncam, 6))
    for i in range(ncam):
        a = 2 * np.pi * i / ncam
        C = np.array([radius * np.cos(a), 0.3 * np.sin(3 * a), radius * np.sin(a)])
        zc = -C / np.linalg.norm(C)
        xc = np.cross([0, 1, 0], zc)
        xc /= np.linalg.norm(xc)
        yc = np.cross(zc, xc)
        R = np.stack([xc, yc, zc])
        cams[i, :3] = Rotation.from_matrix(R).as_rotvec()
        cams[i, 3:] = -R @ C
    return cams


def ba_problem(ncam, npt, sigma, seed, perturb=0.01):
    """Dense config-4 style problem: returns K, cams (perturbed), X float32, obs (ncam,npt,2) float32."""
    rng = np.random.default_rng(seed)
    K, _ = load_pose_csv()
    cams = ring_cameras(ncam)
    X = rng.normal(0, 1, (npt, 3))
    X /= np.maximum(1.0, np.linalg.norm(X, axis=1, keepdims=True) / rng.uniform(0.2, 1.0, (npt, 1)))
    from scipy.spatial.transform import Rotation
    obs = np.empty((ncam, npt, 2), np.float32)
    for c in range(ncam):
        R = Rotation.from_rotvec(cams[c, :3]).as_matrix()
        Xc = X @ R.T + cams[c, 3:]
        obs[c, :, 0] = K[0, 0] * Xc[:, 0] / Xc[:, 2] + K[0, 2]
        obs[c, :, 1] = K[1, 1] * Xc[:, 1] / Xc[:, 2] + K[1, 2]
    obs += rng.normal(0, sigma, obs.shape).astype(np.float32)
    cams_p = cams * (1 + perturb * rng.standard_normal(cams.shape))
    Xp = (X * (1 + perturb * rng.standard_normal(X.shape))).astype(np.float32)
    return K, cams_p, Xp, obs


def gustav_scene(n_images, seed=0, clutter=200, desc_noise=1.5, pix_noise=0.0):
    """Synthetic 'Gustav-geometry' sequence: the first n_images cameras of pose.csv look at
    the reference's own cloud; every visible point yields a keypoint (its projection) with a persistent
    SIFT-like descriptor (+ small per-view integer noise), plus random clutter features.
    Returns K, P (n,3,4), features [(kp (m,2) f32, des (m,128) f32)], point ids per feature."""
    K, P = load_pose_csv()
    rng = np.random.default_rng(seed)
    X = sparse_points()
    base = sift_like(rng, len(X))
    feats, ids = [], []
    for k in range(n_images):
        x, z = project(P[k], X)
        vis = np.flatnonzero((z > 0.5) & (x[:, 0] > 1) & (x[:, 0] < 967) & (x[:, 1] > 1) & (x[:, 1] < 647))
        kp = (x[vis] + rng.normal(0, pix_noise, (len(vis), 2))).astype(np.float32) if pix_noise > 0 else x[vis].astype(np.float32)
        des = np.clip(base[vis] + np.rint(rng.normal(0, desc_noise, (len(vis), 128))), 0, 255).astype(np.float32)
        ckp = rng.uniform([1, 1], [967, 647], (clutter, 2)).astype(np.float32)
        cdes = sift_like(rng, clutter)
        order = rng.permutation(len(vis) + clutter)
        feats.append((np.vstack([kp, ckp])[order], np.vstack([des, cdes])[order]))
        ids.append(np.hstack([vis, -np.ones(clutter, int)])[order])
    return K, P[:n_images], feats, ids


def scene_image(w, h, seed, shift=(0.0, 0.0)):
    """Procedural grayscale uint8 test image (no dataset on the box): overlapping soft blobs, hard-edged rectangles,
    oriented gratings and low-pass noise, so SIFT finds corners, blobs and textured regions over several octaves.
    `shift` translates the content (in pixels) to build overlapping view pairs."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    xx = xx + shift[0]
    yy = yy + shift[1]
    span = max(w, h)
    img = np.full((h, w), 90.0)
    for _ in range(int(60 * w * h / (256 * 192)) + 8):
        cx, cy = rng.uniform(-0.2, 1.2) * w, rng.uniform(-0.2, 1.2) * h
        s = np.exp(rng.uniform(np.log(1.5), np.log(span / 12)))
        ang, ecc = rng.uniform(0, np.pi), rng.uniform(0.5, 1.0)
        u = (xx - cx) * np.cos(ang) + (yy - cy) * np.sin(ang)
        v = -(xx - cx) * np.sin(ang) + (yy - cy) * np.cos(ang)
        img += rng.uniform(-70, 70) * np.exp(-(u * u + (v / ecc) ** 2) / (2 * s * s))
    for _ in range(int(14 * w * h / (256 * 192)) + 3):
        cx, cy = rng.uniform(0, w), rng.uniform(0, h)
        a, b = rng.uniform(3, span / 8), rng.uniform(3, span / 8)
        ang = rng.uniform(0, np.pi)
        u = (xx - cx) * np.cos(ang) + (yy - cy) * np.sin(ang)
        v = -(xx - cx) * np.sin(ang) + (yy - cy) * np.cos(ang)
        img += rng.uniform(-50, 50) * ((np.abs(u) < a) & (np.abs(v) < b))
    for _ in range(3):
        ang, f, ph = rng.uniform(0, np.pi), rng.uniform(0.05, 0.4), rng.uniform(0, 6.28)
        cx, cy, s = rng.uniform(0, w), rng.uniform(0, h), span / 6
        img += 15 * np.sin((xx * np.cos(ang) + yy * np.sin(ang)) * f + ph) * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * s * s))
    # band-limited texture that translates with the content: a sum of random plane waves
    for _ in range(40):
        ang, f, ph, a = rng.uniform(0, np.pi), rng.uniform(0.2, 1.2), rng.uniform(0, 6.28), rng.uniform(1, 5)
        img += a * np.sin((xx * np.cos(ang) + yy * np.sin(ang)) * f + ph)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def layered_views(n_views, w, h, seed, f=None, baseline=0.25, depths=(10.0, 6.0, 4.0)):
    """A synthetic image SEQUENCE with known geometry, for driving the pipeline from pixels: textured fronto-parallel
    layers at `depths`, seen by a camera that translates along +x by `baseline` per view (no rotation), so layer l moves
    by f * baseline / depth_l pixels per view and nearer layers occlude farther ones.
    Returns (images: list of (h, w, 3) uint8 BGR, K (3, 3), P: list of 3x4 projection matrices)."""
    f = float(f if f is not None else 0.9 * w)
    K = np.array([[f, 0, w / 2.0], [0, f, h / 2.0], [0, 0, 1.0]])
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    shapes = []
    for l in range(1, len(depths)):
        s = []
        for _ in range(5):
            s.append((rng.uniform(-0.1, 1.3) * w, rng.uniform(0.05, 0.95) * h, rng.uniform(0.08, 0.2) * w, rng.uniform(0.1, 0.3) * h))
        shapes.append(s)
    images, P = [], []
    for k in range(n_views):
        b = k * baseline
        img = scene_image(w, h, seed * 17 + 1, shift=(f * b / depths[0], 0.0)).astype(np.float64)
        for l in range(1, len(depths)):
            d = f * b / depths[l]
            tex = scene_image(w, h, seed * 17 + 1 + l, shift=(d, 0.0))
            mask = np.zeros((h, w), bool)
            for cx, cy, a, bb in shapes[l - 1]:
                mask |= (np.abs(xx + d - cx) < a) & (np.abs(yy - cy) < bb)
            img = np.where(mask, tex, img)
        g = np.clip(np.rint(img), 0, 255).astype(np.uint8)
        images.append(np.stack([g, g, g], -1))
        Rt = np.hstack([np.eye(3), np.array([[-b], [0.0], [0.0]])])
        P.append(K @ Rt)
    return images, K, P


def ring_scene(n_images, n_desc, n_scene, seed=0, desc_noise=1.5, pix_noise=0.3):
    """An exhaustive-matching scene (isfm.py:56-94): `n_images` cameras on a ring (K of pose.csv) look at `n_scene` points of
    the unit ball; every image holds the projections of ALL scene points (+ pixel noise) with a persistent SIFT-like
    descriptor (+ integer noise per view) and n_desc - n_scene clutter features, shuffled per image.  Every image has exactly
    n_desc features, so the pairs share one shape.  Image k is a function of (seed, k) alone: every rank of a sharded run
    generates identical images without communicating.  Returns K, P [n,3,4], image(k) -> (kp (n_desc,2) f32, des (n_desc,128)
    f32, ids (n_desc,) scene-point id or -1)."""
    K, _ = load_pose_csv()
    cams = ring_cameras(n_images)
    rng = np.random.default_rng(seed)
    X = rng.normal(size=(n_scene, 3))
    X = X / np.maximum(np.linalg.norm(X, axis=1, keepdims=True), 1.0) * rng.uniform(0.2, 1.0, (n_scene, 1))
    base = sift_like(rng, n_scene)
    P = np.empty((n_images, 3, 4))
    for k in range(n_images):
        th = np.linalg.norm(cams[k, :3])
        kx = cams[k, :3] / th if th > 0 else np.zeros(3)
        Kx = np.array([[0, -kx[2], kx[1]], [kx[2], 0, -kx[0]], [-kx[1], kx[0], 0]])
        R = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * (Kx @ Kx)
        P[k] = K @ np.c_[R, cams[k, 3:]]

    def image(k):
        r = np.random.default_rng([seed, 1000 + k])
        x, _ = project(P[k], X)
        kp = (x + r.normal(0, pix_noise, x.shape)).astype(np.float32)
        des = np.clip(base + np.rint(r.normal(0, desc_noise, base.shape)), 0, 255).astype(np.float32)
        nc = n_desc - n_scene
        ckp = r.uniform([1, 1], [967, 647], (nc, 2)).astype(np.float32)
        cdes = sift_like(r, nc)
        order = r.permutation(n_desc)
        return np.vstack([kp, ckp])[order], np.vstack([des, cdes])[order], np.hstack([np.arange(n_scene), -np.ones(nc, int)])[order]

    return K, P, image


def fast_texture(n, seed, device="cpu"):
    """An n x n uint8 texture (torch tensor on `device`) for the rendered views: 1/f noise (FFT-shaped Gaussian noise: structure
    at every scale, so SIFT finds blobs in several octaves), a few hundred random rectangles (corners and edges; a 2-D difference
    array summed twice: O(n^2) whatever their number) — non-periodic content, unlike tiling a small image (repeated structure
    would fail the ratio test).  The random draws come from NumPy (seeded), the arithmetic runs in torch float64."""
    import torch
    rng = np.random.default_rng(seed)
    dev = torch.device(device)
    fy, fx = torch.fft.fftfreq(n, dtype=torch.float64, device=dev)[:, None], torch.fft.rfftfreq(n, dtype=torch.float64, device=dev)[None, :]
    rad = torch.sqrt(fx * fx + fy * fy)
    rad[0, 0] = 1.0
    re, im = rng.standard_normal((n, n // 2 + 1)), rng.standard_normal((n, n // 2 + 1))
    spec = torch.complex(torch.from_numpy(re).to(dev), torch.from_numpy(im).to(dev)) / rad ** 1.3
    spec[0, 0] = 0.0
    spec = spec * torch.exp(-(rad / 0.18) ** 2)              # nothing above ~0.18 cycles per texel: survives the view's resampling
    img = torch.fft.irfft2(spec, s=(n, n))
    img = 55.0 * img / img.std()
    m = 12 * n // 32
    x0, y0 = rng.integers(0, n, m), rng.integers(0, n, m)
    ww, hh = rng.integers(4, max(5, n // 10), m), rng.integers(4, max(5, n // 10), m)
    x1, y1 = np.minimum(x0 + ww, n), np.minimum(y0 + hh, n)
    amp = rng.uniform(-45, 45, m)
    diff = np.zeros((n + 1, n + 1))
    np.add.at(diff, (y0, x0), amp)
    np.add.at(diff, (y0, x1), -amp)
    np.add.at(diff, (y1, x0), -amp)
    np.add.at(diff, (y1, x1), amp)
    img = img + torch.cumsum(torch.cumsum(torch.from_numpy(diff).to(dev), 0), 1)[:n, :n]
    return torch.clamp(torch.round(img + 120.0), 0, 255).to(torch.uint8)


def gustav_views(n_images=57, w=968, h=648, seed=0, n_planes=7, scale=2, tex=1024, device=None):
    """The reference's camera path SEEN THROUGH PIXELS (BASELINE configs[2]; the Gustav II Adolf photographs are not available):
    the `n_images` first cameras of pose.csv (sfm.py:423's own output) look at a scene of `n_planes` two-sided textured quads
    placed through the bounding box of the reference's cloud (sparse.ply) — real 3-D structure with parallax and occlusion, so
    that SIFT -> knnMatch -> findEssentialMat / recoverPose -> triangulate -> solvePnPRansac (sfm.py:301-409) has something to
    reconstruct — inside a far textured box (no empty pixels).  Frames are rendered at scale x (w, h) — sfm.py:40 halves every
    photograph before anything else — by exact ray / plane intersection (per plane a homography of the pixel grid), nearest hit
    wins, bilinear texture lookup.  Test-data generation, not product: torch float64 on `device` (the GPU when there is one —
    57 full-size frames take a second there and minutes in NumPy); the SAME uint8 frames then feed the HIP path and the oracle.
    Returns (images: list of (scale h, scale w, 3) uint8 BGR NumPy arrays, K for the DOWNSCALED frames as in pose.csv, P [n, 3, 4])."""
    import torch
    dev = torch.device(device if device is not None else ("cuda" if torch.cuda.is_available() else "cpu"))
    K, P = load_pose_csv()
    P = np.asarray(P[:n_images], np.float64)
    X = sparse_points()
    lo, hi = np.percentile(X, 10, axis=0), np.percentile(X, 90, axis=0)
    ctr, ext = 0.5 * (lo + hi), float(np.linalg.norm(hi - lo))
    rng = np.random.default_rng(seed)
    planes = []
    for p in range(n_planes):
        nrm = rng.standard_normal(3)
        nrm /= np.linalg.norm(nrm)
        u = np.cross(nrm, rng.standard_normal(3))
        u /= np.linalg.norm(u)
        v = np.cross(nrm, u)
        o = ctr + rng.uniform(-0.25, 0.25, 3) * ext
        half = rng.uniform(0.35, 0.6) * ext
        planes.append((o, u, v, nrm, half))
    big = 6.0 * ext
    for ax in range(3):
        for sgn in (-1.0, 1.0):
            nrm = np.zeros(3); nrm[ax] = sgn
            u = np.zeros(3); u[(ax + 1) % 3] = 1.0
            v = np.cross(nrm, u)
            planes.append((ctr + nrm * big, u, v, nrm, 1.2 * big))
    texs = torch.stack([fast_texture(tex, seed * 101 + p, dev) for p in range(len(planes))]).to(torch.float32)
    W, H = scale * w, scale * h
    Ks = K.copy()
    Ks[:2] *= scale                                          # the full-size frame's intrinsics (sfm.py:20-26 divides them back)
    Kinv = np.linalg.inv(Ks)
    xs = torch.arange(W, dtype=torch.float64, device=dev)[None, :]
    ys = torch.arange(H, dtype=torch.float64, device=dev)[:, None]
    images = []
    for k in range(n_images):
        R, t = decompose_P(K, P[k])
        Cc = -R.T @ t
        M = R.T @ Kinv                                       # world direction of pixel (x, y): M @ [x, y, 1]; camera depth of a hit = its ray parameter
        best = torch.full((H, W), float("inf"), dtype=torch.float64, device=dev)
        ta = torch.zeros((H, W), dtype=torch.float64, device=dev)
        tb = torch.zeros((H, W), dtype=torch.float64, device=dev)
        ti = torch.zeros((H, W), dtype=torch.int64, device=dev)
        for pi, (o, u, v, nrm, half) in enumerate(planes):
            dn, du, dv = nrm @ M, u @ M, v @ M               # linear forms in (x, y, 1)
            den = dn[0] * xs + dn[1] * ys + dn[2]
            s = float(nrm @ (o - Cc)) / torch.where(den.abs() < 1e-12, torch.full_like(den, 1e-12), den)
            a = float((Cc - o) @ u) + s * (du[0] * xs + du[1] * ys + du[2])
            b = float((Cc - o) @ v) + s * (dv[0] * xs + dv[1] * ys + dv[2])
            hit = (s > 0.05) & (a.abs() < half) & (b.abs() < half) & (s < best)
            best = torch.where(hit, s, best)
            sc = (tex - 1) / (2.0 * half)
            ta = torch.where(hit, (a + half) * sc, ta)
            tb = torch.where(hit, (b + half) * sc, tb)
            ti = torch.where(hit, torch.full_like(ti, pi), ti)
        x0 = torch.clamp(torch.floor(ta).long(), 0, tex - 2)
        y0 = torch.clamp(torch.floor(tb).long(), 0, tex - 2)
        fx, fy = torch.clamp(ta - x0, 0, 1).float(), torch.clamp(tb - y0, 0, 1).float()
        g = (texs[ti, y0, x0] * (1 - fx) * (1 - fy) + texs[ti, y0, x0 + 1] * fx * (1 - fy)
             + texs[ti, y0 + 1, x0] * (1 - fx) * fy + texs[ti, y0 + 1, x0 + 1] * fx * fy)
        g = torch.clamp(torch.round(g), 0, 255).to(torch.uint8).cpu().numpy()
        images.append(np.stack([g, g, g], -1))
    return images, K, P
