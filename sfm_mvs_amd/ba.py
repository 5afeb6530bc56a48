"""Block Gauss-Newton bundle adjustment on top of the device sweep ("next" after the hot path).

The reference's BA (sfm.py:138-157, off by default) hands a dense finite-difference problem to SciPy: (5N+22)
residual evaluations per Jacobian, "close to half a minute per frame" (sfm.py:378).  Here ONE sweep
(sfm_project_residual / sfm_ba_dense_sweep) returns the analytic per-camera 6x6 and per-point 3x3 normal-equation
blocks, and the optimiser alternates damped block updates — cameras with points fixed ("resection"), then points
with cameras fixed ("intersection") — each accepted only if the fp64 cost decreases (Levenberg-Marquardt damping on
the block diagonals).  Every evaluation of the cost / blocks is a device sweep; the tiny 6x6 / 3x3 solves are
batched host / elementwise-device arithmetic.
"""
import numpy as np
import torch

from . import ops


def _solve3_sym(A, g):
    """Batched closed-form solve of symmetric 3x3 systems A x = g on the device (A [n,9], g [n,3])."""
    a, b, c, d, e, f = A[:, 0], A[:, 1], A[:, 2], A[:, 4], A[:, 5], A[:, 8]
    c00, c01, c02 = d * f - e * e, c * e - b * f, b * e - c * d
    det = a * c00 + b * c01 + c * c02
    c11, c12, c22 = a * f - c * c, b * c - a * e, a * d - b * b
    inv = 1.0 / torch.where(det.abs() > 1e-300, det, torch.ones_like(det))
    x0 = (c00 * g[:, 0] + c01 * g[:, 1] + c02 * g[:, 2]) * inv
    x1 = (c01 * g[:, 0] + c11 * g[:, 1] + c12 * g[:, 2]) * inv
    x2 = (c02 * g[:, 0] + c12 * g[:, 1] + c22 * g[:, 2]) * inv
    return torch.stack([x0, x1, x2], 1)


def bundle_adjust(cams, K, X, obs, cam_idx=None, pt_idx=None, iters=10, lam=1e-3, fix_first_camera=True, log=None):
    """cams [ncam,6] float64 (rvec, tvec), X [npt,3] float32, obs either [nobs,2] with cam_idx/pt_idx (sparse) or
    [ncam,npt,2] (dense visibility).  All CUDA tensors.  Returns (cams, X, history of fp64 costs)."""
    dense = obs.dim() == 3
    cams = cams.clone().to(torch.float64)
    X = X.clone().to(torch.float32)
    say = log or (lambda *a: None)

    def sweep(c, x, jac):
        if dense:
            out = ops.ba_dense_sweep(c, K, x, obs, want_cam=jac, want_pt=jac)
            return out, float(out["sumsq"].item())
        out = ops.project_residual(c, K, x, obs, cam_idx, pt_idx, want_proj=False, want_jac=jac, want_pt_jac=jac, want_res2=True)
        return out, float(out["res2"].item())

    out, cost = sweep(cams, X, True)
    hist = [cost]
    for it in range(iters):
        # resection: cameras, points fixed
        A = out["JtJ_cam"].cpu().numpy().reshape(-1, 6, 6).copy()
        g = out["Jtr_cam"].cpu().numpy()
        step_done = False
        for _ in range(6):
            Ad = A.copy()
            idx = np.arange(6)
            Ad[:, idx, idx] *= 1.0 + lam
            dc = np.linalg.solve(Ad, g[:, :, None])[:, :, 0]
            if fix_first_camera:
                dc[0] = 0
            cand = cams - torch.from_numpy(dc).to(cams.device)
            o2, c2 = sweep(cand, X, True)
            if c2 < cost:
                cams, out, cost, lam, step_done = cand, o2, c2, max(lam * 0.3, 1e-9), True
                break
            lam *= 10
        # intersection: points, cameras fixed
        Ap = out["JtJ_pt"].clone()
        Ap[:, 0] *= 1.0 + lam
        Ap[:, 4] *= 1.0 + lam
        Ap[:, 8] *= 1.0 + lam
        dx = _solve3_sym(Ap, out["Jtr_pt"])
        cand = (X.to(torch.float64) - dx).to(torch.float32)
        o2, c2 = sweep(cams, cand, True)
        if c2 < cost:
            X, out, cost, lam, step_done = cand, o2, c2, max(lam * 0.3, 1e-9), True
        else:
            lam *= 10
        hist.append(cost)
        say(f"ba iter {it}: cost {cost:.6g} lambda {lam:.2g}")
        if not step_done:
            break
    return cams, X, hist
