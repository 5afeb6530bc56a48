"""An INDEPENDENT float64 SIFT (NumPy / SciPy) — the SIFT analogue of tests/np_solvers.py.

Shares nothing with oracle/sift_oracle.c or csrc/sift.hip: the scale space is scipy.ndimage.gaussian_filter1d, extrema come from
maximum / minimum filters, the sub-pixel fit is numpy.linalg.solve on central differences, histograms are numpy.add.at.  It follows
the published algorithm (Lowe 2004) with the conventions of OpenCV's implementation that the reference calls through
cv2.xfeatures2d.SIFT_create().detectAndCompute (sfm.py:246-252): 2x bilinear base image lifted from an assumed 0.5 px blur to
sigma 1.6, 3 layers per octave, contrast threshold 0.04 (/ layers), edge threshold 10, 36-bin orientation histogram with the
[1 4 6 4 1] / 16 smoothing and the 0.8 peak ratio, 4 x 4 x 8 descriptor with trilinear spreading, 0.2 clipping, x 512 to bytes,
keypoint angles clockwise.  Everything is float64 and vectorised per keypoint; nothing is rounded the way a float32 SIMD loop
rounds — agreement with the oracle is therefore statistical (positions within half a pixel, scales within 10 %, descriptor cosine
>= 0.97 on matched keypoints), which is what tests/test_oracle_sift.py asserts.
"""
import numpy as np
from scipy.ndimage import gaussian_filter1d, maximum_filter, minimum_filter

SIGMA, LAYERS, CONTRAST, EDGE = 1.6, 3, 0.04, 10.0
BORDER, MAX_STEPS = 5, 5


def _blur(img, sigma):
    # separable Gaussian, mirror border (= BORDER_REFLECT_101), support +-4 sigma (OpenCV: ksize = round(8 sigma + 1) | 1)
    r = int(round(sigma * 8 + 1)) | 1
    t = (r // 2) / sigma
    return gaussian_filter1d(gaussian_filter1d(img, sigma, axis=1, mode="mirror", truncate=t), sigma, axis=0, mode="mirror", truncate=t)


def _upsample2(g):
    # bilinear, pixel centres at half-integers: destination i samples the source at (i + 0.5) / 2 - 0.5, clamped at the border
    def axis(n):
        s = (np.arange(2 * n) + 0.5) / 2 - 0.5
        i0 = np.floor(s).astype(int)
        f = s - i0
        return np.clip(i0, 0, n - 1), np.clip(i0 + 1, 0, n - 1), f
    y0, y1, fy = axis(g.shape[0])
    x0, x1, fx = axis(g.shape[1])
    rows = g[y0] * (1 - fy)[:, None] + g[y1] * fy[:, None]
    return rows[:, x0] * (1 - fx)[None, :] + rows[:, x1] * fx[None, :]


def scale_space(gray):
    base = _blur(_upsample2(gray.astype(np.float64)), np.sqrt(max(SIGMA ** 2 - 4 * 0.5 ** 2, 0.01)))
    n_oct = int(round(np.log2(min(base.shape)) - 2)) + 1        # OpenCV: cvRound(log2(min side of the DOUBLED base) - 2) - firstOctave, firstOctave = -1
    k = 2.0 ** (1.0 / LAYERS)
    inc = [np.sqrt((SIGMA * k ** i) ** 2 - (SIGMA * k ** (i - 1)) ** 2) for i in range(1, LAYERS + 3)]
    gauss, dog = [], []
    for o in range(n_oct):
        layers = [base]
        for s in inc:
            layers.append(_blur(layers[-1], s))
        gauss.append(np.stack(layers))
        dog.append(np.diff(gauss[-1], axis=0))
        base = layers[LAYERS][::2, ::2]
    return gauss, dog


def _fit(D, l, r, c):
    """Sub-pixel / sub-scale fit of a DoG extremum by Newton steps on the 3-D quadratic; None if it leaves the volume or is rejected."""
    n_l, h, w = D.shape
    for _ in range(MAX_STEPS):
        v = D[l - 1:l + 2, r - 1:r + 2, c - 1:c + 2] / 255.0
        g = 0.5 * np.array([v[1, 1, 2] - v[1, 1, 0], v[1, 2, 1] - v[1, 0, 1], v[2, 1, 1] - v[0, 1, 1]])
        c2 = 2 * v[1, 1, 1]
        dxx, dyy, dss = v[1, 1, 2] + v[1, 1, 0] - c2, v[1, 2, 1] + v[1, 0, 1] - c2, v[2, 1, 1] + v[0, 1, 1] - c2
        dxy = 0.25 * (v[1, 2, 2] - v[1, 2, 0] - v[1, 0, 2] + v[1, 0, 0])
        dxs = 0.25 * (v[2, 1, 2] - v[2, 1, 0] - v[0, 1, 2] + v[0, 1, 0])
        dys = 0.25 * (v[2, 2, 1] - v[2, 0, 1] - v[0, 2, 1] + v[0, 0, 1])
        H = np.array([[dxx, dxy, dxs], [dxy, dyy, dys], [dxs, dys, dss]])
        try:
            x = -np.linalg.solve(H, g)
        except np.linalg.LinAlgError:
            return None
        if np.all(np.abs(x) < 0.5):
            break
        if np.any(np.abs(x) > 1e9):
            return None
        c, r, l = c + int(round(x[0])), r + int(round(x[1])), l + int(round(x[2]))
        if not (1 <= l <= n_l - 2 and BORDER <= c < w - BORDER and BORDER <= r < h - BORDER):
            return None
    else:
        return None
    contrast = v[1, 1, 1] + 0.5 * g.dot(x)
    if abs(contrast) * LAYERS < CONTRAST:
        return None
    tr, det = dxx + dyy, dxx * dyy - dxy * dxy
    if det <= 0 or tr * tr * EDGE >= (EDGE + 1) ** 2 * det:
        return None
    return l, r, c, x, abs(contrast)


def _orientations(L, r, c, scl):
    """Dominant gradient orientations (degrees, counter-clockwise in image coordinates with y down = OpenCV's histogram bins)."""
    rad = int(round(3 * 1.5 * scl))
    sig = 1.5 * scl
    ys, xs = np.mgrid[-rad:rad + 1, -rad:rad + 1]
    y, x = r + ys, c + xs
    ok = (y > 0) & (y < L.shape[0] - 1) & (x > 0) & (x < L.shape[1] - 1)
    y, x, ys, xs = y[ok], x[ok], ys[ok], xs[ok]
    dx = L[y, x + 1] - L[y, x - 1]
    dy = L[y - 1, x] - L[y + 1, x]
    mag, ori = np.hypot(dx, dy), np.degrees(np.arctan2(dy, dx)) % 360
    wgt = np.exp(-(xs * xs + ys * ys) / (2 * sig * sig))
    hist = np.zeros(36)
    np.add.at(hist, np.round(ori * 36 / 360).astype(int) % 36, wgt * mag)
    p = np.concatenate([hist[-2:], hist, hist[:2]])
    sm = (p[:-4] + p[4:]) / 16 + (p[1:-3] + p[3:-1]) * 4 / 16 + p[2:-2] * 6 / 16
    out, top = [], sm.max()
    for b in range(36):
        lft, rgt = sm[b - 1], sm[(b + 1) % 36]
        if sm[b] > lft and sm[b] > rgt and sm[b] >= 0.8 * top:
            bb = (b + 0.5 * (lft - rgt) / (lft - 2 * sm[b] + rgt)) % 36
            out.append(bb * 10.0)
    return out


def _descriptor(L, r_f, c_f, ori_deg, scl):
    d, nb = 4, 8
    hw = 3.0 * scl
    rad = int(round(hw * np.sqrt(2) * (d + 1) * 0.5))
    rad = min(rad, int(np.hypot(*L.shape)))
    ct, st = np.cos(np.radians(ori_deg)) / hw, np.sin(np.radians(ori_deg)) / hw
    r0, c0 = int(round(r_f)), int(round(c_f))
    ys, xs = np.mgrid[-rad:rad + 1, -rad:rad + 1]
    crot, rrot = xs * ct - ys * st, xs * st + ys * ct
    rb, cb = rrot + d / 2 - 0.5, crot + d / 2 - 0.5
    y, x = r0 + ys, c0 + xs
    ok = (rb > -1) & (rb < d) & (cb > -1) & (cb < d) & (y > 0) & (y < L.shape[0] - 1) & (x > 0) & (x < L.shape[1] - 1)
    y, x, rb, cb, rrot, crot = y[ok], x[ok], rb[ok], cb[ok], rrot[ok], crot[ok]
    dx = L[y, x + 1] - L[y, x - 1]
    dy = L[y - 1, x] - L[y + 1, x]
    mag = np.hypot(dx, dy) * np.exp(-(rrot ** 2 + crot ** 2) / (0.5 * d * d))
    ob = ((np.degrees(np.arctan2(dy, dx)) - ori_deg) % 360) * nb / 360
    r_i, c_i, o_i = np.floor(rb).astype(int), np.floor(cb).astype(int), np.floor(ob).astype(int)
    fr, fc, fo = rb - r_i, cb - c_i, ob - o_i
    hist = np.zeros((d + 2, d + 2, nb))
    for ar, wr in ((0, 1 - fr), (1, fr)):
        for ac, wc in ((0, 1 - fc), (1, fc)):
            for ao, wo in ((0, 1 - fo), (1, fo)):
                np.add.at(hist, (r_i + 1 + ar, c_i + 1 + ac, (o_i + ao) % nb), mag * wr * wc * wo)
    v = hist[1:d + 1, 1:d + 1].reshape(-1)
    thr = 0.2 * np.linalg.norm(v)
    v = np.minimum(v, thr)
    v = v * (512.0 / max(np.linalg.norm(v), 1e-7))
    return np.clip(np.round(v), 0, 255)


def detect_and_compute(gray):
    """(kp (n, 4) float64 {x, y, size, angle in degrees clockwise}, desc (n, 128)) in the coordinates of `gray`."""
    gauss, dog = scale_space(gray)
    thr = np.floor(0.5 * CONTRAST / LAYERS * 255)
    kps, descs = [], []
    for o, D in enumerate(dog):
        if min(D.shape[1:]) <= 2 * BORDER + 2:
            continue
        mx = maximum_filter(D, size=3, mode="constant", cval=-np.inf)
        mn = minimum_filter(D, size=3, mode="constant", cval=np.inf)
        cand = (np.abs(D) > thr) & (((D > 0) & (D >= mx)) | ((D < 0) & (D <= mn)))
        cand[0] = cand[-1] = False
        cand[:, :BORDER] = cand[:, -BORDER:] = False
        cand[:, :, :BORDER] = cand[:, :, -BORDER:] = False
        for l, r, c in zip(*np.nonzero(cand)):
            fit = _fit(D, l, r, c)
            if fit is None:
                continue
            l, r, c, x, _ = fit
            scl = SIGMA * 2.0 ** ((l + x[2]) / LAYERS)            # in pixels of this octave
            size = scl * 2.0 ** o * 2 * 0.5                       # OpenCV: sigma 2^((l + xi) / s) 2^o * 2, halved for the doubled base
            px, py = (c + x[0]) * 2.0 ** o * 0.5, (r + x[1]) * 2.0 ** o * 0.5
            L = gauss[o][l]
            for a in _orientations(L, r, c, scl):
                ang = (360.0 - a) % 360.0
                kps.append((px, py, size, ang))
                descs.append(_descriptor(L, r + x[1], c + x[0], a, scl))
    return np.array(kps).reshape(-1, 4), np.array(descs).reshape(-1, 128)
