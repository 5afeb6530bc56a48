"""Independent NumPy implementations (LAPACK SVD, numpy.roots) of two minimal solvers — TEST INFRASTRUCTURE.

They share no arithmetic with oracle/solvers_oracle.c or the library's host_solvers.h (different eigen-solvers, different
root finder), so they pin the SOLUTION SETS of those restatements: the same essential matrices up to sign and order for
any five correspondences, and the same pose on exact data.  (On noisy minimal samples EPnP's result depends on the
null-space basis the SVD happens to return, so there the comparison is against the planted truth, not between solvers.)
"""
import numpy as np

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



def _epnp_betas(L, rho, cols):
    return np.linalg.lstsq(L[:, cols], rho, rcond=None)[0]



def epnp_numpy(K, Xw, uv, axis_signs=(1, 1, 1)):
    """The same algorithm in NumPy (reference for the C++ solver's tests; large samples).
    axis_signs: orientation of the three principal axes the control points sit on.  An SVD fixes them only up to sign; on
    exact data the pose does not depend on the choice, on NOISY data it does at the noise level (other control points, other
    algebraic error) — OpenCV's answer is the one its own SVD's signs give, so a comparison has to try the eight mirrorings."""
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
        cws[i] = cws[0] + axis_signs[i - 1] * np.sqrt(dc[i - 1] / n) * U[:, i - 1]
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


