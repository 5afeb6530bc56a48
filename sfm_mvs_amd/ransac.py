"""RANSAC entry points of the path — cv2.findEssentialMat (sfm.py:307), cv2.recoverPose (sfm.py:311),
cv2.solvePnPRansac (sfm.py:67) — with OpenCV's sequential semantics and device-side scoring.

Control flow follows RANSACPointSetRegistrator::run: RNG seeded with 2^64-1, `modelPoints` distinct
indices per iteration, every model of an iteration scored, the best replaced only on a STRICTLY larger
inlier count, `niters` re-estimated after each improvement.  The subsets do not depend on scoring, so
iterations are generated in chunks, all their hypotheses are scored in ONE kernel launch over all
correspondences (H x N, integer counts + masks), and the host then replays the sequential bookkeeping
— identical results, without one launch per hypothesis.

`backend` supplies the device kernels (default: HipBackend over libsfmhip.so).  Tests inject a CPU
backend built on the oracle to check masks/poses bit-for-bit; the product never imports it.
"""
import numpy as np
import torch

from . import hostgeom as hg
from . import ops
from ._lib import on_device


class HipBackend:
    """Scoring / sweep kernels on the current CUDA (HIP) device."""

    def __init__(self, device=None, dlt_rows=4):
        self.device = torch.device(device if device is not None else "cuda")
        self.dlt_rows = dlt_rows

    def _d(self, a, dtype):
        return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).to(self.device)

    def prepare_essential(self, x1n, x2n):
        return self._d(x1n, torch.float64), self._d(x2n, torch.float64)

    def score_essential(self, prep, Es, thr2):
        counts, mask = ops.score_essential(self._d(Es, torch.float64), prep[0], prep[1], thr2, want_mask=True)
        return counts.cpu().numpy(), mask.cpu().numpy()

    def recover_pose_score(self, prep, Ps, dist):
        import ctypes
        from . import _lib
        P = np.ascontiguousarray(Ps, np.float64).reshape(-1, 12)
        h, n = P.shape[0], prep[0].shape[0]
        counts = torch.empty(h, dtype=torch.int32, device=self.device)
        mask = torch.empty((h, n), dtype=torch.uint8, device=self.device)
        with on_device(self.device):
            _lib.check(_lib.lib().sfm_recover_pose_score(P.ctypes.data_as(ctypes.c_void_p), h, _lib.ptr(prep[0]),
                                                         _lib.ptr(prep[1]), n, float(dist), self.dlt_rows, _lib.ptr(counts),
                                                         _lib.ptr(mask), _lib.stream_ptr()), "sfm_recover_pose_score")
        return counts.cpu().numpy(), mask.cpu().numpy()

    def prepare_pnp(self, X, uv):
        return self._d(X, torch.float32), self._d(uv, torch.float32)

    def score_pnp(self, prep, poses, K, thr2):
        counts, mask = ops.score_pnp(self._d(poses, torch.float64), K, prep[0], prep[1], thr2, want_mask=True)
        return counts.cpu().numpy(), mask.cpu().numpy()

    def pose_sweep(self, prep, rvec, tvec, K, want_jac):
        cams = self._d(np.hstack([rvec, tvec])[None], torch.float64)
        out = ops.project_residual(cams, K, prep[0], prep[1], want_proj=False, want_jac=want_jac, want_res2=True)
        if want_jac:
            pack = torch.cat([out["JtJ_cam"].reshape(-1), out["Jtr_cam"].reshape(-1), out["res2"]]).cpu().numpy()
            return pack[:36].reshape(6, 6), pack[36:42], float(np.sqrt(pack[42]))
        return None, None, float(np.sqrt(out["res2"].item()))


_default_backend = None


def default_backend():
    global _default_backend
    if _default_backend is None:
        _default_backend = HipBackend()
    return _default_backend


def _sample_subsets(rng, count, model_points, iters):
    """getSubset: `model_points` distinct indices per iteration, duplicates redrawn in place."""
    out = np.empty((iters, model_points), np.int64)
    for it in range(iters):
        idx = []
        while len(idx) < model_points:
            v = rng.uniform(0, count)
            if v not in idx:
                idx.append(v)
        out[it] = idx
    return out


def _ransac(count, model_points, max_iters, confidence, make_models, score, chunk=4, max_chunk=64):
    """Sequential-semantics RANSAC with chunked hypothesis generation and batched scoring.  Chunks start small and
    double: with a high inlier ratio `niters` collapses after the first good model, and hypotheses generated beyond it
    would be wasted host work (the drawn subsets, and therefore the result, do not depend on the chunking)."""
    rng = hg.CvRNG()
    niters = max_iters
    best_count, best_model, best_mask = 0, None, None
    it = 0
    while it < niters:
        m = min(chunk, niters - it)
        chunk = min(2 * chunk, max_chunk)
        subsets = _sample_subsets(rng, count, model_points, m)
        models, owner = [], []
        for k in range(m):
            for mod in make_models(subsets[k]):
                models.append(mod)
                owner.append(k)
        if models:
            counts, masks = score(np.array(models))
        stop = False
        for j, k in enumerate(owner):
            if it + k >= niters:          # niters may have shrunk inside this chunk
                stop = True
                break
            good = int(counts[j])
            if good > max(best_count, model_points - 1):
                best_count, best_model, best_mask = good, models[j], masks[j].copy()
                niters = hg.ransac_update_num_iters(confidence, (count - good) / count, model_points, niters)
        if stop:
            break
        it += m
    return best_model, best_mask, best_count


# ------------------------------------------------------------------------------- findEssentialMat
def find_essential_mat(pts0, pts1, K, prob=0.999, threshold=1.0, max_iters=1000, backend=None):
    """cv2.findEssentialMat(points1, points2, K, method=RANSAC, prob, threshold) → (E, mask{0,1} (N,1) uint8)."""
    be = backend or default_backend()
    p0 = np.asarray(pts0, np.float64).reshape(-1, 2)
    p1 = np.asarray(pts1, np.float64).reshape(-1, 2)
    n = len(p0)
    if n < 5:
        return None, None
    fx, fy, cx, cy = K[0][0], K[1][1], K[0][2], K[1][2]
    x0 = np.stack([(p0[:, 0] - cx) / fx, (p0[:, 1] - cy) / fy], 1)
    x1 = np.stack([(p1[:, 0] - cx) / fx, (p1[:, 1] - cy) / fy], 1)
    thr = threshold / ((fx + fy) / 2)
    thr2 = np.float32(thr * thr)
    prep = be.prepare_essential(x0, x1)
    if n == 5:
        models = hg.five_point(x0, x1)
        if len(models) == 0:
            return None, None
        return models[0], np.ones((n, 1), np.uint8)
    model, mask, good = _ransac(n, 5, max_iters, prob, lambda idx: list(hg.five_point(x0[idx], x1[idx])),
                                lambda Es: be.score_essential(prep, Es.reshape(-1, 9), thr2))
    if model is None:
        return None, None
    return model.reshape(3, 3), mask.reshape(-1, 1).astype(np.uint8)


# ------------------------------------------------------------------------------------ recoverPose
def recover_pose(E, pts0, pts1, K, distance_thresh=50.0, backend=None):
    """cv2.recoverPose(E, points1, points2, K) → (good, R, t (3,1), mask{0,255} (N,1) uint8)."""
    be = backend or default_backend()
    p0 = np.asarray(pts0, np.float64).reshape(-1, 2)
    p1 = np.asarray(pts1, np.float64).reshape(-1, 2)
    fx, fy, cx, cy = K[0][0], K[1][1], K[0][2], K[1][2]
    x0 = np.stack([(p0[:, 0] - cx) / fx, (p0[:, 1] - cy) / fy], 1)
    x1 = np.stack([(p1[:, 0] - cx) / fx, (p1[:, 1] - cy) / fy], 1)
    R1, R2, t = hg.decompose_essential(E)
    cands = [(R1, t), (R2, t), (R1, -t), (R2, -t)]
    Ps = np.array([np.hstack([R, tt[:, None]]) for R, tt in cands])
    counts, masks = be.recover_pose_score(be.prepare_essential(x0, x1), Ps, distance_thresh)
    g = [int(c) for c in counts]
    # OpenCV's cascade of >= tests in candidate order
    if g[0] >= g[1] and g[0] >= g[2] and g[0] >= g[3]:
        k = 0
    elif g[1] >= g[0] and g[1] >= g[2] and g[1] >= g[3]:
        k = 1
    elif g[2] >= g[0] and g[2] >= g[1] and g[2] >= g[3]:
        k = 2
    else:
        k = 3
    R, tt = cands[k]
    return g[k], R.copy(), tt.reshape(3, 1).copy(), masks[k].reshape(-1, 1).astype(np.uint8)


# --------------------------------------------------------------------------------- solvePnPRansac
def solve_pnp_ransac(X, uv, K, iterations_count=100, reprojection_error=8.0, confidence=0.99, backend=None):
    """cv2.solvePnPRansac(objectPoints, imagePoints, K, dist) with all defaults (the reference's 5th
    positional argument lands in `rvec` and is ignored: SURVEY §3.6-1): RANSAC over EPnP on 5-point
    samples, then solvePnP(ITERATIVE) = DLT init + Levenberg-Marquardt on the inlier set.
    Returns (ok, rvec (3,1), tvec (3,1), inliers (k,1) int32 or None)."""
    be = backend or default_backend()
    Xf = np.ascontiguousarray(np.asarray(X, np.float32).reshape(-1, 3))
    uvf = np.ascontiguousarray(np.asarray(uv, np.float32).reshape(-1, 2))
    K = np.asarray(K, np.float64).reshape(3, 3)
    n = len(Xf)
    if n < 5:          # OpenCV asserts npoints >= 4 and switches to P3P for exactly 4; not on this path
        raise ValueError("solvePnPRansac: at least 5 correspondences are required on this path")
    Xd, uvd = Xf.astype(np.float64), uvf.astype(np.float64)
    prep = be.prepare_pnp(Xf, uvf)

    def make(idx):
        try:
            R, t = hg.epnp(K, Xd[idx], uvd[idx])
        except np.linalg.LinAlgError:
            return []
        if not (np.all(np.isfinite(R)) and np.all(np.isfinite(t))):
            return []
        return [np.hstack([hg.rodrigues_mat2vec(R), t])]

    thr2 = np.float32(reprojection_error * reprojection_error)
    if n == 5:
        models = make(np.arange(5))
        if not models:
            return False, None, None, None
        return True, models[0][:3].reshape(3, 1), models[0][3:].reshape(3, 1), np.arange(5, dtype=np.int32).reshape(-1, 1)
    model, mask, good = _ransac(n, 5, iterations_count, confidence, make, lambda P: be.score_pnp(prep, P, K, thr2))
    if model is None:
        return False, None, None, None
    inl = np.flatnonzero(mask)
    prep_in = be.prepare_pnp(Xf[inl], uvf[inl])
    init = hg.pnp_dlt_init(K, Xd[inl], uvd[inl]) if len(inl) >= 6 else None
    if init is None:     # planar / too few points: start from the RANSAC model (OpenCV uses a homography here)
        init = (model[:3], model[3:])
    rvec, tvec = hg.levmarq_pose(lambda r, t, j: be.pose_sweep(prep_in, r, t, K, j), init[0], init[1])
    return True, rvec.reshape(3, 1), tvec.reshape(3, 1), inl.astype(np.int32).reshape(-1, 1)
