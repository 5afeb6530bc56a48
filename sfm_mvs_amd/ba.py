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


# ---------------------------------------------------------------------------------------------------------------
# Schur-complement Levenberg-Marquardt (dense visibility): the joint step instead of alternating block updates.
# ---------------------------------------------------------------------------------------------------------------
def _inv3_sym(A):
    """Batched inverse of symmetric 3x3 blocks on the device: A [n,9] → [n,9]."""
    a, b, c, d, e, f = A[:, 0], A[:, 1], A[:, 2], A[:, 4], A[:, 5], A[:, 8]
    c00, c01, c02 = d * f - e * e, c * e - b * f, b * e - c * d
    c11, c12, c22 = a * f - c * c, b * c - a * e, a * d - b * b
    det = a * c00 + b * c01 + c * c02
    inv = 1.0 / torch.where(det.abs() > 1e-300, det, torch.ones_like(det))
    return torch.stack([c00, c01, c02, c01, c11, c12, c02, c12, c22], 1) * inv[:, None]


def _bmv(A, x, n):
    """Batched (n x n blocks, row-major [m, n*n]) times [m, n] on the device (sfm_block_matvec)."""
    return ops.block_matvec(A, x, n)


def schur_step(cams, K, X, blocks, lam, fix_first_camera=True, cg_tol=1e-10, cg_iters=200, cam_idx=None, pt_idx=None, device_pcg=True):
    """One damped Gauss-Newton step of the dense problem through the reduced camera system.

        [ B+lam*diag(B)    W           ] [dc]   [g_c]        S dc = g_c - W Cd^-1 g_p,    S = Bd - W Cd^-1 W^T
        [ W^T              C+lam*diag(C)] [dp] = [g_p]       dp   = Cd^-1 (g_p - W^T dc)

    S is never formed: preconditioned conjugate gradients with S x = Bd x - W (Cd^-1 (W^T x)) (two device sweeps per
    iteration, ops.ba_schur_wt / ops.ba_schur_w), block-Jacobi preconditioner Bd^-1.  `blocks` = the dict returned by
    ops.ba_dense_sweep at (cams, X).  Returns (dc [ncam,6], dp [npt,3], number of CG iterations); the update is
    params - step (the sweep's gradient is J^T r).
    Dense visibility: the whole recurrence runs on the device (ops.ba_schur_solve: one C-ABI call, the CG vectors and
    scalars never leave HBM); the torch loop below is the sparse-window form and the cross-check of the device solver."""
    if device_pcg and cam_idx is None:
        dc, dp, it, status = ops.ba_schur_solve(cams, K, X, blocks, lam, fix_first_camera, cg_tol, cg_iters)
        if status & 1:
            raise ops.SfmHipError("schur_step: a camera block is singular (a camera without observations?)")
        return dc, dp, it
    dev = X.device
    B = blocks["JtJ_cam"].clone().view(-1, 6, 6)
    C = blocks["JtJ_pt"].clone()
    gc, gp = blocks["Jtr_cam"], blocks["Jtr_pt"]
    ncam = B.shape[0]
    di = torch.arange(6, device=dev)
    B[:, di, di] *= 1.0 + lam
    C[:, 0] *= 1.0 + lam
    C[:, 4] *= 1.0 + lam
    C[:, 8] *= 1.0 + lam
    Cinv = _inv3_sym(C)
    free = torch.ones((ncam, 1), dtype=torch.float64, device=dev)
    if fix_first_camera:
        free[0] = 0                                                        # gauge: camera 0 does not move
    B = B.contiguous()
    Minv = ops.block_inverse(B, 6)                                         # block-Jacobi preconditioner (sfm_block_inverse)

    def S(x):
        x = x * free
        u = ops.ba_schur_wt(cams, K, X, x, cam_idx, pt_idx)
        w = ops.ba_schur_w(cams, K, X, _bmv(Cinv, u, 3), cam_idx, pt_idx)
        return (_bmv(B, x, 6) - w) * free

    rhs = (gc - ops.ba_schur_w(cams, K, X, _bmv(Cinv, gp, 3), cam_idx, pt_idx)) * free
    x = torch.zeros_like(rhs)
    r = rhs.clone()
    z = _bmv(Minv, r, 6) * free
    p = z.clone()
    rz = (r * z).sum()                                   # CG scalars stay on the device: one host sync per 5 iterations
    stop2 = (cg_tol * cg_tol) * (rhs * rhs).sum()
    it = 0
    while it < cg_iters:
        if it % 5 == 0 and bool(((r * r).sum() <= stop2).item()):
            break
        Sp = S(p)
        alpha = rz / (p * Sp).sum()
        x += alpha * p
        r -= alpha * Sp
        z = _bmv(Minv, r, 6) * free
        rz_new = (r * z).sum()
        p = z + (rz_new / rz) * p
        rz = rz_new
        it += 1
    dp = _bmv(Cinv, gp - ops.ba_schur_wt(cams, K, X, x, cam_idx, pt_idx), 3)
    return x, dp, it


def bundle_adjust_schur(cams, K, X, obs, cam_idx=None, pt_idx=None, iters=10, lam=1e-3, fix_first_camera=True, log=None):
    """Levenberg-Marquardt with the joint Schur-complement step.  obs [ncam,npt,2] (dense visibility) or [nobs,2] with
    cam_idx / pt_idx (sparse: windows of an incremental reconstruction).  Returns (cams, X, history of fp64 costs)."""
    dense = obs.dim() == 3
    if not dense and (cam_idx is None or pt_idx is None):
        raise ops.SfmHipError("bundle_adjust_schur: sparse observations need cam_idx and pt_idx")
    cams = cams.clone().to(torch.float64)
    X = X.clone().to(torch.float32)
    say = log or (lambda *a: None)

    def sweep(c, x):
        if dense:
            out = ops.ba_dense_sweep(c, K, x, obs)
            return out, float(out["sumsq"].item())
        out = ops.project_residual(c, K, x, obs, cam_idx, pt_idx, want_proj=False, want_jac=True, want_pt_jac=True, want_res2=True)
        return out, float(out["res2"].item())

    blocks, cost = sweep(cams, X)
    hist = [cost]
    for it in range(iters):
        accepted = False
        for _ in range(8):
            dc, dp, ncg = schur_step(cams, K, X, blocks, lam, fix_first_camera, cam_idx=None if dense else cam_idx,
                                     pt_idx=None if dense else pt_idx)
            c_new = cams - dc
            x_new = (X.to(torch.float64) - dp).to(torch.float32)
            b_new, cost_new = sweep(c_new, x_new)
            if cost_new < cost:
                cams, X, blocks, accepted = c_new, x_new, b_new, True
                gain = (cost - cost_new) / cost
                cost, lam = cost_new, max(lam / 3.0, 1e-12)
                say(f"schur-lm iter {it}: cost {cost:.6g} lambda {lam:.2g} cg {ncg}")
                break
            lam *= 4.0
        hist.append(cost)
        if not accepted or gain < 1e-9:
            break
    return cams, X, hist
