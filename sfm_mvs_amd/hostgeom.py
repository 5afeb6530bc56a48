"""Host-side (NumPy, fp64) minimal solvers and small-matrix helpers used by the RANSAC entry points.

OpenCV keeps hypothesis GENERATION sequential on the host (one 5-point / EPnP solve per RANSAC
iteration on a handful of points); what scales with the data — scoring every correspondence against
every hypothesis, the cheirality vote, the LM normal equations — runs in the HIP kernels
(sfm_score_essential / sfm_score_pnp / sfm_recover_pose_score / sfm_project_residual).  This module is
the generation side: it restates the published algorithms behind

  cv2.findEssentialMat  (sfm.py:307)  Nistér five-point: null space → 10 cubic constraints → 10th-degree
                                      polynomial in z → up to 10 essential matrices
  cv2.recoverPose       (sfm.py:311)  decomposeEssentialMat: 4 (R, t) candidates
  cv2.solvePnPRansac    (sfm.py:67)   EPnP minimal solver (Lepetit et al.), DLT initialisation and the
                                      Levenberg-Marquardt schedule of the ITERATIVE refinement
  cv2.Rodrigues         (sfm.py:69,84,119)

plus OpenCV's RNG so that RANSAC draws the same subsets.  No device code here and nothing from oracle/.
"""
import numpy as np

DBL_EPSILON = float(np.finfo(np.float64).eps)
FLT_EPSILON = float(np.finfo(np.float32).eps)


# --------------------------------------------------------------------------------------------- RNG
class CvRNG:
    """cv::RNG: multiply-with-carry, seeded with (uint64)-1 by RANSACPointSetRegistrator::run."""
    A = 4164903690

    def __init__(self, state=0xFFFFFFFFFFFFFFFF):
        self.state = state & 0xFFFFFFFFFFFFFFFF

    def next(self):
        self.state = ((self.state & 0xFFFFFFFF) * self.A + (self.state >> 32)) & 0xFFFFFFFFFFFFFFFF
        return self.state & 0xFFFFFFFF

    def uniform(self, a, b):
        return a if a == b else a + self.next() % (b - a)


def ransac_update_num_iters(p, ep, model_points, max_iters):
    """RANSACUpdateNumIters (OpenCV ptsetreg.cpp)."""
    p = min(max(p, 0.0), 1.0)
    ep = min(max(ep, 0.0), 1.0)
    num = max(1.0 - p, np.finfo(np.float64).tiny)
    denom = 1.0 - (1.0 - ep) ** model_points
    if denom < np.finfo(np.float64).tiny:
        return 0
    num = np.log(num)
    denom = np.log(denom)
    return max_iters if (denom >= 0 or -num >= max_iters * (-denom)) else int(np.rint(num / denom))


# --------------------------------------------------------------------------------------- Rodrigues
def rodrigues_vec2mat(rvec):
    r = np.asarray(rvec, np.float64).reshape(3)
    theta = float(np.sqrt(r @ r))
    if theta < DBL_EPSILON:
        return np.eye(3)
    c, s = np.cos(theta), np.sin(theta)
    k = r / theta
    kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return c * np.eye(3) + (1 - c) * np.outer(k, k) + s * kx


def rodrigues_mat2vec(R):
    R = np.asarray(R, np.float64).reshape(3, 3)
    U, _, Vt = np.linalg.svd(R)
    R = U @ Vt
    r = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    s = np.sqrt((r @ r) * 0.25)
    c = min(max((R[0, 0] + R[1, 1] + R[2, 2] - 1) * 0.5, -1.0), 1.0)
    theta = np.arccos(c)
    if s < 1e-5:
        if c > 0:
            return np.zeros(3)
        rx = np.sqrt(max((R[0, 0] + 1) * 0.5, 0.0))
        ry = np.sqrt(max((R[1, 1] + 1) * 0.5, 0.0)) * (-1.0 if R[0, 1] < 0 else 1.0)
        rz = np.sqrt(max((R[2, 2] + 1) * 0.5, 0.0)) * (-1.0 if R[0, 2] < 0 else 1.0)
        if abs(rx) < abs(ry) and abs(rx) < abs(rz) and (R[1, 2] > 0) != (ry * rz > 0):
            rz = -rz
        v = np.array([rx, ry, rz])
        return v * (theta / np.sqrt(v @ v))
    return r * (theta / (2 * s))


# ------------------------------------------------------------------------------------- five-point
# Monomial order of the 10x20 constraint matrix (Nistér): the first ten are eliminated.
_MONO = [(3, 0, 0), (0, 3, 0), (2, 1, 0), (1, 2, 0), (2, 0, 1), (2, 0, 0), (0, 2, 1), (0, 2, 0), (1, 1, 1), (1, 1, 0),
         (1, 0, 2), (1, 0, 1), (1, 0, 0), (0, 1, 2), (0, 1, 1), (0, 1, 0), (0, 0, 3), (0, 0, 2), (0, 0, 1), (0, 0, 0)]


def _pmul(a, b):
    """Product of two polynomials in (x, y, z) stored as coefficient cubes c[i, j, k] of x^i y^j z^k."""
    out = np.zeros((a.shape[0] + b.shape[0] - 1,) * 3)
    for i, j, k in zip(*np.nonzero(a)):
        out[i:i + b.shape[0], j:j + b.shape[1], k:k + b.shape[2]] += a[i, j, k] * b
    return out


def _lin(cx, cy, cz, c1):
    p = np.zeros((2, 2, 2))
    p[1, 0, 0], p[0, 1, 0], p[0, 0, 1], p[0, 0, 0] = cx, cy, cz, c1
    return p


def _row(p):
    q = np.zeros((4, 4, 4))
    q[:p.shape[0], :p.shape[1], :p.shape[2]] = p
    return np.array([q[m] for m in _MONO])


def five_point(x1n, x2n):
    """Essential matrices consistent with 5 K-normalised correspondences (x2^T E x1 = 0).
    Returns an array (m, 3, 3), m <= 10, each of unit Frobenius norm, ordered by ascending root z."""
    x1n, x2n = np.asarray(x1n, np.float64), np.asarray(x2n, np.float64)
    Q = np.stack([x1n[:, 0] * x2n[:, 0], x1n[:, 1] * x2n[:, 0], x2n[:, 0], x1n[:, 0] * x2n[:, 1],
                  x1n[:, 1] * x2n[:, 1], x2n[:, 1], x1n[:, 0], x1n[:, 1], np.ones(len(x1n))], 1)
    _, _, Vt = np.linalg.svd(Q)
    B = Vt[5:9]                                  # null-space basis: E = x B0 + y B1 + z B2 + B3
    E = [[_lin(B[0, 3 * r + c], B[1, 3 * r + c], B[2, 3 * r + c], B[3, 3 * r + c]) for c in range(3)] for r in range(3)]
    # det(E) = 0
    det = (_pmul(E[0][0], _pmul(E[1][1], E[2][2]) - _pmul(E[1][2], E[2][1]))
           - _pmul(E[0][1], _pmul(E[1][0], E[2][2]) - _pmul(E[1][2], E[2][0]))
           + _pmul(E[0][2], _pmul(E[1][0], E[2][1]) - _pmul(E[1][1], E[2][0])))
    # 2 E E^T E - tr(E E^T) E = 0
    EEt = [[sum(_pmul(E[r][k], E[c][k]) for k in range(3)) for c in range(3)] for r in range(3)]
    tr = EEt[0][0] + EEt[1][1] + EEt[2][2]
    rows = [_row(det)]
    for r in range(3):
        for c in range(3):
            rows.append(_row(2 * sum(_pmul(EEt[r][k], E[k][c]) for k in range(3)) - _pmul(tr, E[r][c])))
    A = np.array(rows)
    try:
        A = np.linalg.solve(A[:, :10], A[:, 10:])
    except np.linalg.LinAlgError:
        return np.zeros((0, 3, 3))

    # rows 4..9 lead with x^2 z, x^2, y^2 z, y^2, xyz, xy; (row_a) - z (row_b) is  x p1(z) + y p2(z) + p3(z)
    def brow(a, b):
        p1 = np.array([-b[0], a[0] - b[1], a[1] - b[2], a[2]])
        p2 = np.array([-b[3], a[3] - b[4], a[4] - b[5], a[5]])
        p3 = np.array([-b[6], a[6] - b[7], a[7] - b[8], a[8] - b[9], a[9]])
        return p1, p2, p3

    Bz = [brow(A[4], A[5]), brow(A[6], A[7]), brow(A[8], A[9])]
    pm, pa, ps = np.polymul, np.polyadd, np.polysub
    detB = pa(ps(pm(Bz[0][0], ps(pm(Bz[1][1], Bz[2][2]), pm(Bz[1][2], Bz[2][1]))),
                 pm(Bz[0][1], ps(pm(Bz[1][0], Bz[2][2]), pm(Bz[1][2], Bz[2][0])))),
              pm(Bz[0][2], ps(pm(Bz[1][0], Bz[2][1]), pm(Bz[1][1], Bz[2][0]))))
    if not np.all(np.isfinite(detB)) or np.abs(detB).max() == 0:
        return np.zeros((0, 3, 3))
    roots = np.roots(detB)
    zs = np.sort(roots[np.abs(roots.imag) <= 1e-10].real)
    out = []
    for z in zs:
        M = np.array([[np.polyval(Bz[r][c], z) for c in range(3)] for r in range(3)])
        v = np.linalg.svd(M)[2][2]
        if abs(v[2]) < 1e-10:
            continue
        x, y = v[0] / v[2], v[1] / v[2]
        Em = (x * B[0] + y * B[1] + z * B[2] + B[3]).reshape(3, 3)
        out.append(Em / np.linalg.norm(Em))
    return np.array(out).reshape(-1, 3, 3)


def decompose_essential(E):
    """decomposeEssentialMat: R1 = U W Vt, R2 = U W^T Vt, t = U[:, 2] (det-fixed SVD)."""
    U, _, Vt = np.linalg.svd(np.asarray(E, np.float64).reshape(3, 3))
    if np.linalg.det(U) < 0:
        U = -U
    if np.linalg.det(Vt) < 0:
        Vt = -Vt
    W = np.array([[0, 1, 0], [-1, 0, 0], [0, 0, 1.0]])
    return U @ W @ Vt, U @ W.T @ Vt, U[:, 2].copy()


# ------------------------------------------------------------------------------------------- EPnP
def _epnp_betas(L, rho, cols):
    return np.linalg.lstsq(L[:, cols], rho, rcond=None)[0]


def epnp(K, Xw, uv):
    """EPnP (Lepetit, Moreno-Noguer, Fua 2009) as used for the 5-point minimal samples of solvePnPRansac.
    Xw (n,3), uv (n,2) pixels.  Returns R (3,3), t (3,).  Runs the library's host solver (sfm_host_epnp, C++: the
    NumPy restatement below took 1.2 ms per call and 60 % of a 57-camera run); samples larger than 64 points use NumPy."""
    Xw = np.ascontiguousarray(Xw, np.float64).reshape(-1, 3)
    uv = np.ascontiguousarray(uv, np.float64).reshape(-1, 2)
    if 4 <= len(Xw) <= 64:
        import ctypes
        from . import _lib
        Kc = np.ascontiguousarray(K, np.float64).reshape(9)
        R, t = np.empty(9), np.empty(3)
        vp = ctypes.c_void_p
        _lib.check(_lib.lib().sfm_host_epnp(Kc.ctypes.data_as(vp), Xw.ctypes.data_as(vp), uv.ctypes.data_as(vp), len(Xw),
                                            R.ctypes.data_as(vp), t.ctypes.data_as(vp)), "sfm_host_epnp")
        return R.reshape(3, 3), t
    return epnp_numpy(K, Xw, uv)


def epnp_numpy(K, Xw, uv):
    """The same algorithm in NumPy (reference for the C++ solver's tests; large samples)."""
    Xw = np.asarray(Xw, np.float64)
    uv = np.asarray(uv, np.float64)
    n = len(Xw)
    fu, fv, uc, vc = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    # control points: centroid + principal directions scaled by sqrt(lambda/n)
    cws = np.zeros((4, 3))
    cws[0] = Xw.mean(0)
    P0 = Xw - cws[0]
    U, dc, _ = np.linalg.svd(P0.T @ P0)
    for i in range(1, 4):
        cws[i] = cws[0] + np.sqrt(dc[i - 1] / n) * U[:, i - 1]
    # barycentric coordinates
    CC = (cws[1:] - cws[0]).T
    try:
        a123 = np.linalg.solve(CC, (Xw - cws[0]).T).T
    except np.linalg.LinAlgError:
        a123 = (np.linalg.pinv(CC) @ (Xw - cws[0]).T).T
    alphas = np.hstack([1 - a123.sum(1, keepdims=True), a123])
    M = np.zeros((2 * n, 12))
    for j in range(4):
        M[0::2, 3 * j] = alphas[:, j] * fu
        M[0::2, 3 * j + 2] = alphas[:, j] * (uc - uv[:, 0])
        M[1::2, 3 * j + 1] = alphas[:, j] * fv
        M[1::2, 3 * j + 2] = alphas[:, j] * (vc - uv[:, 1])
    Ut = np.linalg.svd(M.T @ M)[0].T            # rows: eigenvectors, descending eigenvalue
    v = [Ut[11], Ut[10], Ut[9], Ut[8]]
    pairs = [(0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3)]
    dv = np.array([[v[i][3 * a:3 * a + 3] - v[i][3 * b:3 * b + 3] for (a, b) in pairs] for i in range(4)])
    L = np.zeros((6, 10))
    for i in range(6):
        d0, d1, d2, d3 = dv[0, i], dv[1, i], dv[2, i], dv[3, i]
        L[i] = [d0 @ d0, 2 * d0 @ d1, d1 @ d1, 2 * d0 @ d2, 2 * d1 @ d2, d2 @ d2, 2 * d0 @ d3, 2 * d1 @ d3, 2 * d2 @ d3,
                d3 @ d3]
    rho = np.array([np.sum((cws[a] - cws[b]) ** 2) for (a, b) in pairs])

    def approx1():
        b4 = _epnp_betas(L, rho, [0, 1, 3, 6])
        if b4[0] < 0:
            b0 = np.sqrt(-b4[0])
            return np.array([b0, -b4[1] / b0, -b4[2] / b0, -b4[3] / b0])
        b0 = np.sqrt(b4[0])
        return np.array([b0, b4[1] / b0, b4[2] / b0, b4[3] / b0])

    def approx2():
        b3 = _epnp_betas(L, rho, [0, 1, 2])
        if b3[0] < 0:
            b0, b1 = np.sqrt(-b3[0]), (np.sqrt(-b3[2]) if b3[2] < 0 else 0.0)
        else:
            b0, b1 = np.sqrt(b3[0]), (np.sqrt(b3[2]) if b3[2] > 0 else 0.0)
        if b3[1] < 0:
            b0 = -b0
        return np.array([b0, b1, 0.0, 0.0])

    def approx3():
        b5 = _epnp_betas(L, rho, [0, 1, 2, 3, 4])
        if b5[0] < 0:
            b0, b1 = np.sqrt(-b5[0]), (np.sqrt(-b5[2]) if b5[2] < 0 else 0.0)
        else:
            b0, b1 = np.sqrt(b5[0]), (np.sqrt(b5[2]) if b5[2] > 0 else 0.0)
        if b5[1] < 0:
            b0 = -b0
        return np.array([b0, b1, (b5[3] / b0 if b0 != 0 else 0.0), 0.0])

    def gauss_newton(b):
        b = b.copy()
        for _ in range(5):
            A = np.stack([2 * L[:, 0] * b[0] + L[:, 1] * b[1] + L[:, 3] * b[2] + L[:, 6] * b[3],
                          L[:, 1] * b[0] + 2 * L[:, 2] * b[1] + L[:, 4] * b[2] + L[:, 7] * b[3],
                          L[:, 3] * b[0] + L[:, 4] * b[1] + 2 * L[:, 5] * b[2] + L[:, 8] * b[3],
                          L[:, 6] * b[0] + L[:, 7] * b[1] + L[:, 8] * b[2] + 2 * L[:, 9] * b[3]], 1)
            r = rho - (L[:, 0] * b[0] * b[0] + L[:, 1] * b[0] * b[1] + L[:, 2] * b[1] * b[1] + L[:, 3] * b[0] * b[2] +
                       L[:, 4] * b[1] * b[2] + L[:, 5] * b[2] * b[2] + L[:, 6] * b[0] * b[3] + L[:, 7] * b[1] * b[3] +
                       L[:, 8] * b[2] * b[3] + L[:, 9] * b[3] * b[3])
            b = b + np.linalg.lstsq(A, r, rcond=None)[0]
        return b

    def pose_from_betas(b):
        ccs = sum(b[i] * v[i].reshape(4, 3) for i in range(4))
        pcs = alphas @ ccs
        if pcs[0, 2] < 0:
            ccs, pcs = -ccs, -pcs
        pc0, pw0 = pcs.mean(0), Xw.mean(0)
        ABt = (pcs - pc0).T @ (Xw - pw0)
        Ua, _, Vta = np.linalg.svd(ABt)
        R = Ua @ Vta
        if np.linalg.det(R) < 0:
            R[2] = -R[2]
        t = pc0 - R @ pw0
        Xc = Xw @ R.T + t
        with np.errstate(divide="ignore", invalid="ignore"):
            ue = uc + fu * Xc[:, 0] / Xc[:, 2]
            ve = vc + fv * Xc[:, 1] / Xc[:, 2]
        err = np.sqrt((uv[:, 0] - ue) ** 2 + (uv[:, 1] - ve) ** 2).sum() / n
        return (err if np.isfinite(err) else np.inf), R, t

    with np.errstate(invalid="ignore", divide="ignore"):
        cands = [pose_from_betas(gauss_newton(f())) for f in (approx1, approx2, approx3)]
    best = 0
    if cands[1][0] < cands[0][0]:
        best = 1
    if cands[2][0] < cands[best][0]:
        best = 2
    return cands[best][1], cands[best][2]


# ------------------------------------------------------------------- ITERATIVE initialisation (DLT)
def pnp_dlt_init(K, Xw, uv):
    """Non-planar initialisation of solvePnP(ITERATIVE): 12-parameter DLT on K-normalised points,
    rotation re-orthonormalised, translation rescaled.  Returns rvec, tvec (or None if planar)."""
    Xw = np.asarray(Xw, np.float64)
    uv = np.asarray(uv, np.float64)
    mn = np.stack([(uv[:, 0] - K[0, 2]) / K[0, 0], (uv[:, 1] - K[1, 2]) / K[1, 1]], 1)
    Mc = Xw.mean(0)
    W = np.linalg.svd((Xw - Mc).T @ (Xw - Mc))[1]
    if W[2] / W[1] < 1e-3:
        return None
    n = len(Xw)
    L = np.zeros((2 * n, 12))
    L[0::2, 0:3], L[0::2, 3] = Xw, 1
    L[0::2, 8:11], L[0::2, 11] = -mn[:, :1] * Xw, -mn[:, 0]
    L[1::2, 4:7], L[1::2, 7] = Xw, 1
    L[1::2, 8:11], L[1::2, 11] = -mn[:, 1:] * Xw, -mn[:, 1]
    RRt = np.linalg.svd(L.T @ L)[2][11].reshape(3, 4)
    if np.linalg.det(RRt[:, :3]) < 0:
        RRt = -RRt
    RR, tt = RRt[:, :3], RRt[:, 3]
    sc = np.linalg.norm(RR)
    U, _, Vt = np.linalg.svd(RR)
    R = U @ Vt
    return rodrigues_mat2vec(R), tt * (np.linalg.norm(R) / sc)


def levmarq_pose(sweep, rvec, tvec, max_iter=20, eps=FLT_EPSILON):
    """CvLevMarq as driven by solvePnP(ITERATIVE): lambda = 10^k (k from -3), the DIAGONAL of J^T J is
    multiplied by (1 + lambda), a step is retried with larger lambda while the error grows.
    `sweep(rvec, tvec, want_jac)` returns (JtJ 6x6, Jtr 6, |err|) evaluated on the device."""
    param = np.hstack([rvec, tvec]).astype(np.float64)
    lam = -3
    JtJ, Jtr, err_norm = sweep(param[:3], param[3:], True)
    prev_err = err_norm
    iters = 0
    while True:
        prev_param = param.copy()
        while True:
            A = JtJ.copy()
            A[np.diag_indices(6)] *= 1.0 + 10.0 ** lam
            step = np.linalg.lstsq(A, Jtr, rcond=None)[0]
            param = prev_param - step
            _, _, err_norm = sweep(param[:3], param[3:], False)
            if err_norm > prev_err:
                lam += 1
                if lam <= 16:
                    continue
            break
        lam = max(lam - 1, -16)
        iters += 1
        rel = np.linalg.norm(param - prev_param) / max(np.linalg.norm(prev_param), np.finfo(np.float64).tiny)
        if iters >= max_iter or rel < eps:
            break
        prev_err = err_norm
        JtJ, Jtr, _ = sweep(param[:3], param[3:], True)
    return param[:3], param[3:]
