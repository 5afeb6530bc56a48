"""Randomised parity sweep of the geometry operators on the GPU box: HIP path (through the C-ABI) vs the CPU oracle.

  python scripts/fuzz_geometry.py [seconds] [seed]

Families (one case = one random draw of a family; the bars are those of tests/test_gpu_geometry.py / test_gpu_pipeline.py):
  ransac    findEssentialMat -> recoverPose -> solvePnPRansac on a random camera pair of pose.csv: random point count (5 .. 4000),
            pixel noise, outlier fraction (0 .. 0.8) and magnitude, thresholds, iteration caps, confidences, duplicated / collinear /
            coplanar / three-distinct-point sets, other intrinsics.  E, every
            mask, the inlier list and the iteration counts IDENTICAL; the refined pose within 1e-9.
  tri       triangulatePoints (4- and 6-row systems; the faithful Jacobi path and the guarded fast path): float32 outputs within
            1e-6 relative of the oracle, the guarded path bit-identical to the faithful one on >= 99.5 % of the points and within
            3e-7 on the rest; far points, near-parallel rays, identical observations mixed in.
  common    common_points (sfm.py:215-239) on coarse grids with duplicates, x-only / y-only hits, empty intersections: indices
            bit-exact.
  resid     reprojection error / normal-equation sums of one camera (fp64 sums within 1e-10 relative, inlier mask bit-exact).
  sweep     the dense and the indexed multi-camera Gauss-Newton sweeps (1 .. 40 cameras, fp64 blocks within 1e-9 relative).
  score     the RANSAC scoring kernels (PnP poses / essential matrices): inlier counts and masks bit-exact.
"""
import os
import sys
import time

os.environ.setdefault("OMP_WAIT_POLICY", "passive")

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import sfm_mvs_amd
from sfm_mvs_amd import _lib, ops, ransac
from oracle import oracle as O
from datagen import ba_problem, decompose_P, gustav_pair

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
f32 = np.float32


def cu(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return (t if dtype is None else t.to(dtype)).cuda()


def both(fo, fh):
    """Run the oracle's and the product's version of one call; an exception on one side only is a mismatch."""
    ro = rh = eo = eh = None
    try:
        ro = fo()
    except Exception as e:  # noqa: BLE001
        eo = e
    try:
        rh = fh()
    except Exception as e:  # noqa: BLE001
        eh = e
    return ro, rh, eo, eh


def case_ransac():
    k = int(rng.integers(0, 56))
    n = int(np.exp(rng.uniform(np.log(5), np.log(4000))))
    sigma = float(rng.choice([0.0, 0.05, 0.3, 1.0, 3.0]))
    K, P1, P2, X, x1, x2 = gustav_pair(k, n, sigma, seed=int(rng.integers(1 << 30)))
    nb = int(n * rng.choice([0.0, 0.0, 0.1, 0.3, 0.6, 0.8]))
    if nb:
        bad = rng.permutation(n)[:nb]
        mag = float(rng.choice([3.0, 20.0, 150.0]))
        x2 = x2.copy(); x2[bad] += rng.uniform(-mag, mag, (nb, 2)).astype(f32)
    quirk = int(rng.integers(0, 8))
    if quirk == 0 and n >= 10:       # duplicated correspondences
        d = rng.integers(0, n, n // 3); s = rng.integers(0, n, n // 3)
        x1 = x1.copy(); x2 = x2.copy(); X = X.copy()
        x1[d] = x1[s]; x2[d] = x2[s]; X[d] = X[s]
    elif quirk == 1 and n >= 10:     # a coplanar object
        X = X.copy(); X[:, 2] = 0.2 * X[:, 0] - 0.1 * X[:, 1] + 6.0
        R, t = decompose_P(K, P2)
        xp = (K @ (X @ R.T + t).T).T
        x2 = (xp[:, :2] / xp[:, 2:]).astype(f32)
        R1, t1 = decompose_P(K, P1)
        xp = (K @ (X @ R1.T + t1).T).T
        x1 = (xp[:, :2] / xp[:, 2:]).astype(f32)
    elif quirk == 2 and n >= 6:      # most image points of the second view on one line (degenerate five-point samples)
        m = rng.random(n) < 0.7
        x2 = x2.copy(); x2[m, 1] = (0.3 * x2[m, 0] + 100.0).astype(f32)
    elif quirk == 3:                 # three distinct correspondences only
        src = rng.integers(0, n, 3)
        pick = src[rng.integers(0, 3, n)]
        x1, x2, X = x1[pick].copy(), x2[pick].copy(), X[pick].copy()
    elif quirk == 4:                 # another camera: random focal length / principal point
        f = float(rng.uniform(300, 4000))
        K2 = np.array([[f, 0, rng.uniform(200, 800)], [0, f * rng.uniform(0.9, 1.1), rng.uniform(100, 600)], [0, 0, 1.0]])
        R1, t1 = decompose_P(K, P1); R2, t2 = decompose_P(K, P2)
        xa = (K2 @ (X @ R1.T + t1).T).T; xb = (K2 @ (X @ R2.T + t2).T).T
        x1 = (xa[:, :2] / xa[:, 2:] + rng.normal(0, sigma, (n, 2))).astype(f32)
        x2n = (xb[:, :2] / xb[:, 2:] + rng.normal(0, sigma, (n, 2))).astype(f32)
        if nb:
            x2n[bad] += rng.uniform(-mag, mag, (nb, 2)).astype(f32)
        x2 = x2n; K = K2
    thr = float(rng.choice([0.4, 0.4, 1.0, 3.0]))
    prob = float(rng.choice([0.999, 0.999, 0.99]))
    tag = f"ransac pair {k} n {n} sigma {sigma} bad {nb} quirk {quirk} thr {thr} prob {prob}"
    ro, rh, eo, eh = both(lambda: O.find_essential_mat(x1, x2, K, prob, thr, want_stats=True),
                          lambda: ransac.find_essential_mat(x1, x2, K, prob, thr, want_info=True))
    if (eo is None) != (eh is None):
        return tag + f": findEssentialMat raised on one side only ({eo!r} / {eh!r})"
    if eo is None:
        Eo, mo, so = ro; Eh, mh, ih = rh
        if (Eo is None) != (Eh is None):
            return tag + ": findEssentialMat found a model on one side only"
        if Eo is not None:
            if not (np.array_equal(Eo, Eh) and np.array_equal(mo, mh)):
                return tag + ": E / mask differ"
            if n > 5 and (ih[1], ih[2], ih[3]) != (so[2], so[0], so[1]):     # (exactly five points: no RANSAC loop ran, the counters mean nothing)
                return tag + f": counters differ {ih[1:4]} vs {(so[2], so[0], so[1])}"
            sel = mo.ravel() == 1
            if Eo.shape == (3, 3) and sel.sum() >= 1:
                go, Ro, to, m2o = O.recover_pose(Eo, x1[sel], x2[sel], K)
                gh, Rh, th, m2h = ransac.recover_pose(Eh, x1[sel], x2[sel], K)
                if not (go == gh and np.array_equal(m2o, m2h) and np.array_equal(Ro, Rh) and np.array_equal(to, th)):
                    return tag + ": recoverPose differs"
    Xf = X.astype(f32)
    rep = float(rng.choice([8.0, 8.0, 2.0, 20.0]))
    its = int(rng.choice([100, 100, 10, 500])); conf = float(rng.choice([0.99, 0.99, 0.9, 0.9999]))
    tag += f" rep {rep} its {its} conf {conf}"
    ro, rh, eo, eh = both(lambda: O.solve_pnp_ransac(Xf, x2, K, iterations=its, reproj_error=rep, confidence=conf, want_model=True),
                          lambda: ransac.solve_pnp_ransac(Xf, x2, K, iterations_count=its, reprojection_error=rep, confidence=conf, want_info=True))
    if (eo is None) != (eh is None):
        return tag + f": solvePnPRansac raised on one side only ({eo!r} / {eh!r})"
    if eo is None:
        oko, r_o, t_o, io, model, st = ro
        okh, r_h, t_h, inh, info = rh
        if bool(oko) != bool(okh):
            return tag + f": solvePnPRansac ok {oko} vs {okh}"
        if oko:
            if not np.array_equal(io, inh):
                return tag + ": PnP inlier lists differ"
            if info[2] != st:
                return tag + f": PnP refinement status {info[2]} vs {st}"
            nan_o, nan_h = np.isnan(np.r_[r_o.ravel(), t_o.ravel()]), np.isnan(np.r_[r_h.ravel(), t_h.ravel()])
            if nan_o.any() or nan_h.any():          # (five degenerate points: solvePnP(EPNP) hands back what the solver produced)
                return None if np.array_equal(nan_o, nan_h) else tag + ": PnP NaN pattern differs"
            close = np.abs(r_o - r_h).max() <= 1e-9 and np.abs(t_o - t_h).max() <= 1e-9 * max(1.0, np.abs(t_o).max())
            if not close:
                # A rank-deficient or ill-conditioned refinement (three distinct points, a handful of inliers, image points on a line)
                # amplifies the last bit of the sums along its flat directions: the two poses must then explain the inliers
                # equally well (RMS within 1e-7) and, with eight or more distinct points, still agree to 1e-6
                sel = io.ravel()
                def rms(r, t):
                    p = O.project_points_f64(np.asarray(r, float).ravel(), np.asarray(t, float).ravel(), K, Xf[sel].astype(np.float64))
                    return float(np.sqrt(((p - x2[sel]) ** 2).sum(1).mean()))
                distinct = len(np.unique(Xf[sel], axis=0))
                eo_, eh_ = rms(r_o, t_o), rms(r_h, t_h)
                if abs(eo_ - eh_) > 1e-7 * max(1.0, eo_) or (distinct >= 8 and np.abs(r_o - r_h).max() > 1e-6):
                    return tag + f": PnP pose differs by {np.abs(r_o - r_h).max():.3g} / {np.abs(t_o - t_h).max():.3g} (inliers {len(sel)}, distinct {distinct}, rms {eo_:.12g} vs {eh_:.12g})"
    return None


def case_tri():
    k = int(rng.integers(0, 56))
    n = int(np.exp(rng.uniform(0, np.log(20000))))
    sigma = float(rng.choice([0.0, 0.05, 0.3, 1.0, 3.0, 10.0]))
    K, P1, P2, X, x1, x2 = gustav_pair(k, n, sigma, seed=int(rng.integers(1 << 30)))
    quirk = int(rng.integers(0, 6))
    if quirk == 0:                   # identical observations in both views (points at infinity along the baseline's normal)
        m = rng.random(n) < 0.2; x2 = x2.copy(); x2[m] = x1[m]
    elif quirk == 1:                 # wild pixels (far outside the frame)
        m = rng.random(n) < 0.2; x2 = x2.copy(); x2[m] = rng.uniform(-5000, 5000, (int(m.sum()), 2)).astype(f32)
    elif quirk == 2:                 # second camera nearly the first (short baseline)
        P2 = P1 + 1e-4 * (P2 - P1)
        x2 = (x1 + 1e-4 * (x2 - x1)).astype(f32)
    rows = int(rng.choice([4, 6]))
    tag = f"tri pair {k} n {n} sigma {sigma} quirk {quirk} rows {rows}"
    want = O.triangulate(P1, P2, x1.T, x2.T, rows=rows, normalise_w=True)
    got = ops.triangulate(P1, P2, cu(x1).t(), cu(x2).t(), rows=rows, normalise_w=True).cpu().numpy()
    fin = np.isfinite(want).all(0)
    if not np.array_equal(np.isfinite(got).all(0), fin):
        return tag + ": finiteness differs (faithful path)"
    if quirk != 2 and not np.allclose(got[:, fin], want[:, fin], rtol=1e-6, atol=1e-7):
        return tag + f": faithful path off by {np.abs(got[:, fin] - want[:, fin]).max():.3g}"
    if rows == 4 and quirk not in (0, 2):
        fast = ops.triangulate(P1, P2, cu(x1).t(), cu(x2).t(), rows=4, normalise_w="fast").cpu().numpy()
        if not np.array_equal(np.isfinite(fast).all(0), fin):
            return tag + ": finiteness differs (guarded fast path)"
        # (3e-7 of the point's LARGEST coordinate: a coordinate near zero carries the vector's absolute error)
        scale = np.abs(want[:3, fin]).max(0)
        if not (np.abs(fast[:, fin] - want[:, fin]).max(0) <= 3e-7 * scale).all():
            return tag + f": guarded fast path off by {(np.abs(fast[:, fin] - want[:, fin]).max(0) / scale).max():.3g} of the point's largest coordinate"
        if fin.sum() >= 200 and (fast[:, fin] == want[:, fin]).all(0).mean() < 0.995:
            return tag + f": guarded fast path bit-identical on {(fast[:, fin] == want[:, fin]).all(0).mean():.4f} only"
    return None


def case_common():
    n1, n2 = int(np.exp(rng.uniform(0, np.log(6000)))), int(np.exp(rng.uniform(0, np.log(6000))))
    step = float(rng.choice([0.1, 1.0, 10.0]))
    p2 = (np.round(rng.uniform(0, 900, (n2, 2)) / step) * step).astype(f32)
    p1 = (np.round(rng.uniform(0, 900, (n1, 2)) / step) * step).astype(f32)
    share = float(rng.choice([0.0, 0.3, 1.0]))
    m = int(min(n1, n2) * share)
    if m:
        p1[rng.permutation(n1)[:m]] = p2[rng.permutation(n2)[:m]]
    if rng.random() < 0.3 and n2 > 4:
        p2[rng.integers(0, n2, n2 // 4)] = p2[rng.integers(0, n2, n2 // 4)]       # duplicates inside one set
    wi1, wi2, wt1, _ = O.common_points(p1, p2, p2)
    i1, i2, keep = ops.common_points(cu(p1), cu(p2))
    tag = f"common n1 {n1} n2 {n2} step {step} share {share}"
    if not (np.array_equal(i1.cpu().numpy(), wi1) and np.array_equal(i2.cpu().numpy(), wi2)):
        return tag + ": indices differ"
    if not np.array_equal(p2[keep.cpu().numpy()], wt1.reshape(-1, 2)):
        return tag + ": complement differs"
    return None


def case_resid():
    npt = int(np.exp(rng.uniform(0, np.log(30000))))
    sigma = float(rng.choice([0.0, 0.3, 3.0, 30.0]))
    K, cams, X, obs = ba_problem(2, npt, sigma, seed=int(rng.integers(1 << 30)), perturb=float(rng.choice([0.0, 0.01, 0.1])))
    thr2 = float(rng.choice([64.0, 4.0, 400.0]))
    want = O.project_residual(cams[:1], K, X, obs[0], thr2=thr2)
    out = ops.project_residual(cu(cams[:1]), K, cu(X), cu(obs[0]), thr2=thr2, want_inlier=True, want_jac=True, want_pt_jac=True)
    tag = f"resid npt {npt} sigma {sigma} thr2 {thr2}"
    if not np.array_equal(out["inlier"].cpu().numpy(), want["inlier"]):
        return tag + ": inlier mask differs"
    for key in ("JtJ_cam", "Jtr_cam", "JtJ_pt", "Jtr_pt"):
        g, w = out[key].cpu().numpy(), want[key]
        if not np.abs(g - w).max() <= 1e-10 * max(np.abs(w).max(), 1e-300):
            return tag + f": {key} off by {np.abs(g - w).max() / max(np.abs(w).max(), 1e-300):.3g} relative"
    if not abs(out["sumsq"].item() - want["sumsq"][0]) <= 1e-12 * max(want["sumsq"][0], 1e-300):
        return tag + ": sum of squares differs"
    return None


def case_sweep():
    """The dense Gauss-Newton sweep (every camera sees every point) and the indexed one (sparse visibility, arbitrary order)."""
    ncam = int(rng.integers(1, 41)); npt = int(np.exp(rng.uniform(0, np.log(6000))))
    sigma = float(rng.choice([0.0, 0.5, 5.0]))
    K, cams, X, obs = ba_problem(ncam, npt, sigma, seed=int(rng.integers(1 << 30)), perturb=float(rng.choice([0.0, 0.01, 0.05])))
    tag = f"sweep ncam {ncam} npt {npt} sigma {sigma}"
    if rng.random() < 0.5:
        ci = np.repeat(np.arange(ncam), npt).astype(np.int32); pi = np.tile(np.arange(npt), ncam).astype(np.int32)
        want = O.project_residual(cams, K, X, obs.reshape(-1, 2), ci, pi)
        out = ops.ba_dense_sweep(cu(cams), K, cu(X), cu(obs))
        tag += " dense"
    else:
        m = max(1, int(ncam * npt * rng.uniform(0.05, 1.0)))
        sel = rng.permutation(ncam * npt)[:m]
        ci, pi = (sel // npt).astype(np.int32), (sel % npt).astype(np.int32)
        o = obs.reshape(-1, 2)[sel]
        want = O.project_residual(cams, K, X, o, ci, pi)
        out = ops.project_residual(cu(cams), K, cu(X), cu(o), cu(ci), cu(pi), want_jac=True, want_pt_jac=True)
        tag += f" indexed {m}"
    # J^T r is a sum that CANCELS when the residuals are rounding noise (sigma = 0): its error is relative to the size of its
    # terms, sqrt(max diag(J^T J) * sum r^2) by Cauchy-Schwarz, not to the sum itself; the two sides add in different orders
    for key in ("JtJ_cam", "Jtr_cam", "JtJ_pt", "Jtr_pt"):
        g, w = out[key].cpu().numpy(), want[key]
        scale = np.abs(w).max()
        if key.startswith("Jtr"):
            JJ = want["JtJ" + key[3:]]
            # (a residual obs - proj of ~1e-5 px carries the projection's own rounding, ~1e-13 px absolute = 1e-8 of itself:
            # the floor of 1e-4 px per observation stands for that)
            scale = max(scale, float(np.sqrt(np.abs(JJ).max() * max(want["sumsq"][0], 1e-8 * len(ci)))))
        if not np.abs(g - w).max() <= 1e-9 * max(scale, 1e-300):
            return tag + f": {key} off by {np.abs(g - w).max() / max(scale, 1e-300):.3g} of its terms' size"
    if not abs(out["sumsq"].item() - want["sumsq"][0]) <= 1e-11 * max(want["sumsq"][0], 1e-300):
        return tag + f": sum of squares differs by {abs(out['sumsq'].item() - want['sumsq'][0]) / max(want['sumsq'][0], 1e-300):.3g} relative"
    return None


def case_score():
    """RANSAC scoring kernels: inlier counts and masks of several models, bit-exact."""
    if rng.random() < 0.5:
        ncam = int(rng.integers(1, 17)); npt = int(np.exp(rng.uniform(0, np.log(8000))))
        K, cams, X, obs = ba_problem(ncam, npt, float(rng.choice([0.5, 4.0, 20.0])), seed=int(rng.integers(1 << 30)), perturb=float(rng.choice([0.0, 0.002, 0.02])))
        thr2 = float(rng.choice([64.0, 4.0, 1.0]))
        j = int(rng.integers(0, ncam))
        wc, wm = O.score_pnp(cams, K, X, obs[j], thr2=thr2)
        gc, gm = ops.score_pnp(cu(cams), K, cu(X), cu(obs[j]), thr2, want_mask=True)
        tag = f"score pnp ncam {ncam} npt {npt} thr2 {thr2}"
    else:
        n = int(np.exp(rng.uniform(0, np.log(8000)))); m = int(rng.integers(1, 11))
        x1 = rng.uniform(-0.4, 0.4, (n, 2))
        x2 = x1 + [0.05, 0.0] + rng.normal(0, float(rng.choice([0.0, 2e-4, 2e-3])), (n, 2))
        E = np.array([[0, 0, 0], [0, 0, -1], [0, 1, 0.]])
        Es = np.stack([E + rng.normal(0, sg, (3, 3)) for sg in rng.choice([0, 1e-4, 1e-3, 1e-2, 0.1], m)])
        thr2 = np.float32((float(rng.choice([0.4, 1.0, 3.0])) / 1198.0) ** 2)
        wc, wm = O.score_essential(Es, x1, x2, thr2)
        gc, gm = ops.score_essential(cu(Es), cu(x1), cu(x2), thr2, want_mask=True)
        tag = f"score essential n {n} models {m} thr2 {thr2}"
    if not (np.array_equal(gc.cpu().numpy(), wc) and np.array_equal(gm.cpu().numpy(), wm)):
        return tag + ": counts / masks differ"
    return None


FAMILIES = [("ransac", case_ransac, 5), ("tri", case_tri, 3), ("common", case_common, 2), ("resid", case_resid, 2), ("sweep", case_sweep, 2),
            ("score", case_score, 2)]
weights = np.array([w for _, _, w in FAMILIES], float); weights /= weights.sum()
t0 = time.time(); counts = {n: 0 for n, _, _ in FAMILIES}; bad = 0
O.lib(); sfm_mvs_amd.lib()
while time.time() - t0 < budget:
    name, fn, _ = FAMILIES[int(rng.choice(len(FAMILIES), p=weights))]
    try:
        msg = fn()
    except Exception as e:  # noqa: BLE001
        msg = f"{name}: EXCEPTION {e!r}"[:300]
    counts[name] += 1
    if msg:
        bad += 1
        print("MISMATCH", msg, flush=True)
print(f"fuzz_geometry: seed {seed}, {sum(counts.values())} cases {counts}, {bad} mismatches, {time.time() - t0:.0f} s; build {_lib.build_id()}")
