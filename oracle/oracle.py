"""NumPy-facing ctypes wrapper of liboracle.so — TEST INFRASTRUCTURE ONLY.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  The product
package (sfm_mvs_amd/) must never import this module.  Parity status: see sfm_oracle.h
("parity unpinned" for the cv2-backed functions).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "liboracle.so")
_lib = None


def build():
    subprocess.check_call(["make", "-C", _HERE, "-s"])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            build()
        _lib = C.CDLL(_PATH)
        _lib.orc_l2sqr_f32.restype = C.c_float
        _lib.orc_ratio_filter.restype = C.c_int64
        _lib.orc_common_points.restype = C.c_int64
        _lib.orc_to_ply_filter.restype = C.c_int64
        _lib.orc_reprojection_error.restype = C.c_double
        _lib.orc_sift_detect_and_compute.restype = C.c_int64
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def knn2(q, t, nthreads=1):
    q, t = _f32(q), _f32(t)
    nq, dim = q.shape
    nt = t.shape[0]
    idx = np.empty((nq, 2), np.int32)
    dist = np.empty((nq, 2), np.float32)
    lib().orc_knn2_l2_f32(_p(q), C.c_int64(nq), C.c_int64(dim), _p(t), C.c_int64(nt), C.c_int64(dim), C.c_int(dim),
                          _p(idx), _p(dist), C.c_int(nthreads))
    return idx, dist


def l2sqr(a, b):
    a, b = _f32(a).ravel(), _f32(b).ravel()
    return float(lib().orc_l2sqr_f32(_p(a), _p(b), C.c_int(a.size)))


def ratio_filter(idx, dist, ratio=0.70):
    idx = np.ascontiguousarray(idx, np.int32)
    dist = _f32(dist)
    nq = idx.shape[0]
    oq = np.empty(nq, np.int32)
    ot = np.empty(nq, np.int32)
    mask = np.empty(nq, np.uint8)
    m = lib().orc_ratio_filter(_p(idx), _p(dist), C.c_int64(nq), C.c_double(ratio), _p(oq), _p(ot), _p(mask))
    return oq[:m].copy(), ot[:m].copy(), mask


def triangulate(P1, P2, pts1, pts2, rows=4, normalise_w=False):
    """pts1/pts2: (2,N) like cv2.triangulatePoints.  Returns (4,N) float32."""
    P1, P2 = _f64(P1).reshape(12), _f64(P2).reshape(12)
    x1, x2 = _f32(pts1), _f32(pts2)
    n = x1.shape[1]
    X4 = np.empty((4, n), np.float32)
    lib().orc_triangulate_dlt(_p(P1), _p(P2), _p(x1), _p(x2), C.c_int64(n), C.c_int64(1), C.c_int64(n),
                              C.c_int(rows), C.c_int(int(normalise_w)), _p(X4))
    return X4


def jacobi_stats(reset=True):
    """{rotations applied, pairs skipped, calls} of the Jacobi core since the last reset (single-threaded use)."""
    out = np.zeros(3, np.int64)
    lib().orc_jacobi_stats(_p(out), C.c_int(1 if reset else 0))
    return out


def jacobi_svd(A):
    A = _f64(A)
    m, n = A.shape
    w = np.empty(n)
    U = np.empty((m, n))
    Vt = np.empty((n, n))
    lib().orc_jacobi_svd(_p(A), C.c_int(m), C.c_int(n), _p(w), _p(U), _p(Vt))
    return U, w, Vt


def rodrigues_vec2mat(r, want_jac=False):
    r = _f64(r).reshape(3)
    R = np.empty(9)
    J = np.empty(27) if want_jac else None
    lib().orc_rodrigues_vec2mat(_p(r), _p(R), _p(J))
    return (R.reshape(3, 3), J.reshape(3, 9)) if want_jac else R.reshape(3, 3)


def rodrigues_mat2vec(R):
    R = _f64(R).reshape(9)
    r = np.empty(3)
    lib().orc_rodrigues_mat2vec(_p(R), _p(r))
    return r


def project_points(rvec, tvec, K, X):
    X = _f32(X).reshape(-1, 3)
    n = X.shape[0]
    p64 = np.empty((n, 2))
    p32 = np.empty((n, 2), np.float32)
    lib().orc_project_points(_p(_f64(rvec).reshape(3)), _p(_f64(tvec).reshape(3)), _p(_f64(K).reshape(9)), _p(X),
                             C.c_int64(n), C.c_int64(3), _p(p64), _p(p32))
    return p64, p32


def project_points_f64(rvec, tvec, K, X):
    X = _f64(X).reshape(-1, 3)
    out = np.empty((len(X), 2))
    lib().orc_project_points_f64(_p(_f64(rvec).reshape(3)), _p(_f64(tvec).reshape(3)), _p(_f64(K).reshape(9)), _p(X), C.c_int64(len(X)), _p(out))
    return out


def reprojection_error(Rt, K, X, obs):
    """X (N,3) float32, obs (N,2) float32 → (error, projected float32 (N,2))."""
    X = _f32(X).reshape(-1, 3)
    obs = _f32(obs).reshape(-1, 2)
    n = X.shape[0]
    p32 = np.empty((n, 2), np.float32)
    e = lib().orc_reprojection_error(_p(_f64(Rt).reshape(12)), _p(_f64(K).reshape(9)), _p(X), C.c_int64(n),
                                     C.c_int64(3), _p(obs), _p(p32))
    return float(e), p32


def project_residual(cams, K, X, obs, cam_idx=None, pt_idx=None, thr2=64.0, want_jac=True):
    cams = _f64(cams).reshape(-1, 6)
    X = _f32(X).reshape(-1, 3)
    obs = _f32(obs).reshape(-1, 2)
    ncam, npt, nobs = cams.shape[0], X.shape[0], obs.shape[0]
    ci = None if cam_idx is None else np.ascontiguousarray(cam_idx, np.int32)
    pi = None if pt_idx is None else np.ascontiguousarray(pt_idx, np.int32)
    out = dict(proj=np.empty((nobs, 2), np.float32), sumsq=np.zeros(1), inlier=np.empty(nobs, np.uint8), res2=np.zeros(1))
    if want_jac:
        out.update(JtJ_cam=np.zeros((ncam, 36)), Jtr_cam=np.zeros((ncam, 6)), JtJ_pt=np.zeros((npt, 9)),
                   Jtr_pt=np.zeros((npt, 3)))
    lib().orc_project_residual(_p(cams), C.c_int64(ncam), _p(_f64(K).reshape(9)), _p(X), C.c_int64(npt), C.c_int64(3),
                               _p(obs), _p(ci), _p(pi), C.c_int64(nobs), _p(out["proj"]), _p(out["sumsq"]),
                               _p(out["inlier"]), C.c_float(thr2), _p(out.get("JtJ_cam")), _p(out.get("Jtr_cam")),
                               _p(out.get("JtJ_pt")), _p(out.get("Jtr_pt")), _p(out["res2"] if want_jac else None), C.c_int(1))
    return out


def common_points(pts1, pts2, pts3):
    """Mirror of sfm.py:215-239: returns (indx1, indx2, temp_array1, temp_array2)."""
    pts1, pts2, pts3 = _f32(pts1).reshape(-1, 2), _f32(pts2).reshape(-1, 2), _f32(pts3).reshape(-1, 2)
    n1, n2 = pts1.shape[0], pts2.shape[0]
    i1 = np.empty(max(n1, 1), np.int64)
    i2 = np.empty(max(n1, 1), np.int64)
    keep = np.empty(max(n2, 1), np.uint8)
    m = lib().orc_common_points(_p(pts1), C.c_int64(n1), _p(pts2), C.c_int64(n2), _p(i1), _p(i2), _p(keep))
    k = keep[:n2].astype(bool)
    return i1[:m].copy(), i2[:m].copy(), pts2[k], pts3[k]


def to_ply_filter(points):
    pts = _f64(points).reshape(-1, 3)
    n = pts.shape[0]
    scaled = np.empty((n, 3))
    keep = np.empty(n, np.uint8)
    lib().orc_to_ply_filter(_p(pts), C.c_int64(n), _p(scaled), _p(keep))
    return scaled, keep.astype(bool)


def score_essential(E, x1n, x2n, thr2):
    E = _f64(E).reshape(-1, 9)
    x1n, x2n = _f64(x1n).reshape(-1, 2), _f64(x2n).reshape(-1, 2)
    h, n = E.shape[0], x1n.shape[0]
    counts = np.empty(h, np.int32)
    mask = np.empty((h, n), np.uint8)
    lib().orc_score_essential(_p(E), C.c_int(h), _p(x1n), _p(x2n), C.c_int64(n), C.c_float(thr2), _p(counts), _p(mask))
    return counts, mask


def score_pnp(poses, K, X, obs, thr2=64.0):
    poses = _f64(poses).reshape(-1, 6)
    X, obs = _f32(X).reshape(-1, 3), _f32(obs).reshape(-1, 2)
    h, n = poses.shape[0], X.shape[0]
    counts = np.empty(h, np.int32)
    mask = np.empty((h, n), np.uint8)
    lib().orc_score_pnp(_p(poses), C.c_int(h), _p(_f64(K).reshape(9)), _p(X), _p(obs), C.c_int64(n), C.c_float(thr2),
                        _p(counts), _p(mask))
    return counts, mask


def recover_pose_score(Ps, x1n, x2n, dist=50.0, rows=4):
    Ps = _f64(Ps).reshape(-1, 12)
    x1n, x2n = _f64(x1n).reshape(-1, 2), _f64(x2n).reshape(-1, 2)
    h, n = Ps.shape[0], x1n.shape[0]
    counts = np.empty(h, np.int32)
    mask = np.empty((h, n), np.uint8)
    lib().orc_recover_pose_score(_p(Ps), C.c_int(h), _p(x1n), _p(x2n), C.c_int64(n), C.c_double(dist), C.c_int(rows),
                                 _p(counts), _p(mask))
    return counts, mask


# ---- RANSAC entry points (solvers_oracle.c): sequential restatements of the cv2 calls of sfm.py:67,307,311 ----
class CvRNG:
    """cv::RNG stream (seed 2^64-1 inside RANSAC)."""

    def __init__(self, state=0xFFFFFFFFFFFFFFFF):
        self.state = C.c_uint64(state)
        lib().orc_rng_next.restype = C.c_uint32

    def next(self):
        return int(lib().orc_rng_next(C.byref(self.state)))

    def uniform(self, a, b):
        return int(lib().orc_rng_uniform(C.byref(self.state), C.c_int(a), C.c_int(b)))


def ransac_update_num_iters(p, ep, model_points, max_iters):
    return int(lib().orc_ransac_update_num_iters(C.c_double(p), C.c_double(ep), C.c_int(model_points), C.c_int(max_iters)))


def svd(A, full_uv=False):
    """cv::SVD::compute: returns w, U, Vt."""
    A = _f64(A)
    m, n = A.shape
    k = min(m, n)
    w = np.empty(k)
    U = np.empty((m, m if full_uv else k))
    Vt = np.empty((n if full_uv else k, n))
    lib().orc_svd(_p(A), C.c_int(m), C.c_int(n), C.c_int(1 if full_uv else 0), _p(w), _p(U), _p(Vt))
    return w, U, Vt


def solve_poly(coeffs, max_iters=300):
    """cv::solvePoly: coeffs[k] multiplies x^k; complex roots in OpenCV's order."""
    c = _f64(coeffs).reshape(-1)
    deg = len(c) - 1
    re, im = np.empty(deg), np.empty(deg)
    n = lib().orc_solve_poly(_p(c), C.c_int(deg), _p(re), _p(im), C.c_int(max_iters))
    return re[:n] + 1j * im[:n]


def five_point(x1n, x2n):
    x1n, x2n = _f64(x1n).reshape(5, 2), _f64(x2n).reshape(5, 2)
    E = np.empty((10, 9))
    k = lib().orc_five_point(_p(x1n), _p(x2n), _p(E))
    return E[:k].reshape(k, 3, 3).copy()


def k_normalise(pts, K):
    pts = _f32(pts).reshape(-1, 2)
    out = np.empty((len(pts), 2))
    lib().orc_k_normalise(_p(pts), C.c_int64(len(pts)), _p(_f64(K).reshape(9)), _p(out))
    return out


def find_essential_mat(pts0, pts1, K, prob=0.999, threshold=1.0, max_iters=1000, want_stats=False):
    """cv2.findEssentialMat(pts0, pts1, K, RANSAC, prob, threshold) -> (E (3k,3) or None, mask (N,1) uint8 {0,1})."""
    p0, p1 = _f32(pts0).reshape(-1, 2), _f32(pts1).reshape(-1, 2)
    n = len(p0)
    E = np.empty((10, 9))
    mask = np.zeros(max(n, 1), np.uint8)
    stats = np.zeros(3, np.int32)
    k = lib().orc_find_essential_mat(_p(p0), _p(p1), C.c_int64(n), _p(_f64(K).reshape(9)), C.c_double(prob),
                                     C.c_double(threshold), C.c_int(max_iters), _p(E), _p(mask), _p(stats))
    out = (None, None) if k <= 0 else (E[:k].reshape(3 * k, 3).copy(), mask[:n].reshape(-1, 1))
    return out + (stats,) if want_stats else out


def decompose_essential(E):
    R1, R2, t = np.empty((3, 3)), np.empty((3, 3)), np.empty(3)
    lib().orc_decompose_essential(_p(_f64(E).reshape(9)), _p(R1), _p(R2), _p(t))
    return R1, R2, t


def recover_pose(E, pts0, pts1, K, dist=50.0, rows=4):
    """cv2.recoverPose(E, pts0, pts1, K) -> (good, R, t (3,1), mask (N,1) uint8 {0,255})."""
    p0, p1 = _f32(pts0).reshape(-1, 2), _f32(pts1).reshape(-1, 2)
    n = len(p0)
    R, t = np.empty((3, 3)), np.empty(3)
    mask = np.zeros(max(n, 1), np.uint8)
    good = lib().orc_recover_pose(_p(_f64(E).reshape(-1)[:9].copy()), _p(p0), _p(p1), C.c_int64(n), _p(_f64(K).reshape(9)),
                                  C.c_double(dist), C.c_int(rows), _p(R), _p(t), _p(mask))
    return int(good), R, t.reshape(3, 1), mask[:n].reshape(-1, 1)


def epnp(K, Xw, uv):
    Xw, uv = _f64(Xw).reshape(-1, 3), _f64(uv).reshape(-1, 2)
    R, t = np.empty((3, 3)), np.empty(3)
    rc = lib().orc_epnp(_p(_f64(K).reshape(9)), _p(Xw), _p(uv), C.c_int(len(Xw)), _p(R), _p(t))
    if rc != 0:
        raise ValueError("orc_epnp: 4 <= n <= 64 points")
    return R, t


def p3p(K, Xw, uv):
    """solvePnP(P3P) on exactly four correspondences -> (ok, R, t)."""
    Xw, uv = _f64(Xw).reshape(4, 3), _f64(uv).reshape(4, 2)
    R, t = np.zeros((3, 3)), np.zeros(3)
    ok = lib().orc_p3p(_p(_f64(K).reshape(9)), _p(Xw), _p(uv), _p(R), _p(t))
    return bool(ok), R, t


def pnp_dlt_init(K, X, uv):
    """Non-planar initialisation of solvePnP(ITERATIVE) -> (status, rvec, tvec); status 1 planar, 2 too few points."""
    X, uv = _f64(X).reshape(-1, 3), _f64(uv).reshape(-1, 2)
    r, t = np.zeros(3), np.zeros(3)
    st = lib().orc_pnp_dlt_init(_p(X), _p(uv), C.c_int64(len(X)), _p(_f64(K).reshape(9)), _p(r), _p(t))
    return int(st), r, t


def lm_sums(J, err, mode):
    """The 28 sums of one Levenberg-Marquardt sweep (upper triangle of J^T J, J^T e, |e|^2): mode 0 the fixed tree the HIP library
    shares, mode 1 a plain long-double sum in index order (independent check)."""
    J, err = _f64(J).reshape(-1, 6), _f64(err).reshape(-1)
    out = np.zeros(28)
    lib().orc_lm_sums(_p(J), _p(err), C.c_int64(len(err) // 2), C.c_int(mode), _p(out))
    return out


def set_lm_sum_mode(mode):
    lib().orc_set_lm_sum_mode(C.c_int(mode))


def levmarq_pose(K, X, uv, rvec, tvec):
    X, uv = _f64(X).reshape(-1, 3), _f64(uv).reshape(-1, 2)
    r, t = _f64(rvec).reshape(3).copy(), _f64(tvec).reshape(3).copy()
    it = C.c_int(0)
    lib().orc_levmarq_pose(_p(X), _p(uv), C.c_int64(len(X)), _p(_f64(K).reshape(9)), _p(r), _p(t), C.byref(it))
    return r, t, it.value


def solve_pnp_ransac(X, uv, K, iterations=100, reproj_error=8.0, confidence=0.99, want_model=False):
    """cv2.solvePnPRansac(X, uv, K, zeros(5,1)) with the defaults -> (ok, rvec (3,1), tvec (3,1), inliers (k,1) int32)."""
    X, uv = _f32(X).reshape(-1, 3), _f32(uv).reshape(-1, 2)
    n = len(X)
    r, t, model = np.zeros(3), np.zeros(3), np.zeros(6)
    inl = np.empty(max(n, 1), np.int32)
    ninl, st = C.c_int64(0), C.c_int(0)
    ok = lib().orc_solve_pnp_ransac(_p(X), _p(uv), C.c_int64(n), _p(_f64(K).reshape(9)), C.c_int(iterations),
                                    C.c_float(reproj_error), C.c_double(confidence), _p(r), _p(t), _p(inl), C.byref(ninl),
                                    _p(model), C.byref(st))
    if ok < 0:
        raise ValueError("orc_solve_pnp_ransac: at least 4 correspondences (OpenCV asserts npoints >= 4)")
    if ok == 0:
        return (False, None, None, None) + ((None, st.value) if want_model else ())
    out = (True, r.reshape(3, 1), t.reshape(3, 1), inl[:ninl.value].reshape(-1, 1).copy())
    return out + ((model, st.value) if want_model else ())


def bgr2gray(bgr):
    bgr = np.ascontiguousarray(bgr, np.uint8)
    h, w, _ = bgr.shape
    out = np.empty((h, w), np.uint8)
    lib().orc_bgr2gray_u8(_p(bgr), C.c_int64(w), C.c_int64(h), C.c_int64(3 * w), _p(out))
    return out


def pyrdown(img):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape[:2]
    ch = 1 if img.ndim == 2 else img.shape[2]
    out = np.empty(((h + 1) // 2, (w + 1) // 2) + img.shape[2:], np.uint8)
    lib().orc_pyrdown_u8(_p(img), C.c_int64(w), C.c_int64(h), C.c_int(ch), _p(out))
    return out


def sift_gauss_kernel(sigma):
    k = np.zeros(128, np.float32)
    n = lib().orc_sift_gauss_kernel(C.c_double(sigma), _p(k))
    return k[:n].copy()


def sift(gray, n_octave_layers=3, contrast=0.04, edge=10.0, sigma=1.6, cap=200000, want_pyramid=False):
    """detectAndCompute: (kp (n,8) f32 {x,y,size,angle,response,octave bits,class_id bits,0}, desc (n,128) f32[, pyr])."""
    gray = np.ascontiguousarray(gray, np.uint8)
    h, w = gray.shape
    kp = np.zeros((cap, 8), np.float32)
    desc = np.zeros((cap, 128), np.float32)
    pyr = np.zeros((n_octave_layers + 3, 2 * h, 2 * w), np.float32) if want_pyramid else None
    n = lib().orc_sift_detect_and_compute(_p(gray), C.c_int64(w), C.c_int64(h), C.c_int64(w), C.c_int(n_octave_layers),
                                          C.c_double(contrast), C.c_double(edge), C.c_double(sigma), C.c_int64(cap),
                                          _p(kp), _p(desc), _p(pyr))
    n = min(int(n), cap)
    return (kp[:n].copy(), desc[:n].copy(), pyr) if want_pyramid else (kp[:n].copy(), desc[:n].copy())
