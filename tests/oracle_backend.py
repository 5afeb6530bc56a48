"""CPU twins of the product's operator surface built on the ORACLE ONLY (oracle/liboracle.so) — tests and
tests/golden/make_golden.py.  Nothing here imports sfm_mvs_amd: the RANSAC entry points, the minimal solvers, Rodrigues,
projection, triangulation and matching all come from the oracle's own restatements."""
import numpy as np


class OracleCv2:
    """cv2-named facade over the CPU oracle (the CPU twin of sfm_mvs_amd.cv2compat)."""
    RANSAC, NORM_L2, SOLVEPNP_ITERATIVE = 8, 4, 0

    def __init__(self, oracle, rows=4):
        self.O, self.rows = oracle, rows

    def triangulatePoints(self, P1, P2, a, b):
        return self.O.triangulate(P1, P2, np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32), rows=self.rows)

    def Rodrigues(self, src):
        src = np.asarray(src, np.float64)
        if src.size == 9:
            return self.O.rodrigues_mat2vec(src.reshape(3, 3)).reshape(3, 1), None
        return self.O.rodrigues_vec2mat(src.reshape(3)), None

    def convertPointsFromHomogeneous(self, src):
        src = np.asarray(src)
        w = src[:, -1:]
        scale = np.where(w != 0, 1.0 / np.where(w != 0, w, 1), 1.0).astype(src.dtype)
        return (src[:, :-1] * scale).reshape(src.shape[0], 1, src.shape[1] - 1)

    def projectPoints(self, objectPoints, rvec, tvec, cameraMatrix, distCoeffs=None):
        X = np.asarray(objectPoints)
        if X.dtype == np.float64:                      # cv2 computes in the object points' type
            return self.O.project_points_f64(np.ravel(rvec), np.ravel(tvec), cameraMatrix, X.reshape(-1, 3)).reshape(-1, 1, 2), None
        _, p32 = self.O.project_points(np.ravel(rvec), np.ravel(tvec), cameraMatrix, np.float32(X).reshape(-1, 3))
        return p32.reshape(-1, 1, 2), None

    def norm(self, a, b=None, normType=4):
        d = np.asarray(a) if b is None else np.asarray(a) - np.asarray(b)      # difference in the inputs' dtype
        return float(np.sqrt(np.sum(np.float64(d) ** 2)))                        # squares accumulated in double

    def findEssentialMat(self, p1, p2, K, method=8, prob=0.999, threshold=1.0, mask=None):
        return self.O.find_essential_mat(p1, p2, np.asarray(K, np.float64), prob, threshold)

    def recoverPose(self, E, p1, p2, K):
        return self.O.recover_pose(E, p1, p2, np.asarray(K, np.float64), rows=self.rows)

    def solvePnPRansac(self, X, p, K, d, *a, **k):
        return self.O.solve_pnp_ransac(X, p, K)

    # the detector side (sfm.py:40, 243-252)
    COLOR_BGR2GRAY = 6

    def cvtColor(self, img, code):
        assert code == self.COLOR_BGR2GRAY
        return self.O.bgr2gray(img)

    def pyrDown(self, img):
        return self.O.pyrdown(img)

    @property
    def xfeatures2d(self):
        return self

    def SIFT_create(self):
        return _OracleSift(self.O)


class _KeyPoint:
    __slots__ = ("pt", "size", "angle", "response", "octave", "class_id")

    def __init__(self, row):
        self.pt, self.size, self.angle, self.response = (float(row[0]), float(row[1])), float(row[2]), float(row[3]), float(row[4])
        self.octave, self.class_id = int(np.float32(row[5]).view(np.int32)), int(np.float32(row[6]).view(np.int32))


class _OracleSift:
    def __init__(self, oracle):
        self.O = oracle

    def detectAndCompute(self, gray, mask):
        kp, des = self.O.sift(np.asarray(gray))
        return [_KeyPoint(r) for r in kp], des


class _Backend:
    """Duck-typed stand-in for sfm_mvs_amd.pipeline.Backend (attributes cv, match, reproj)."""

    def __init__(self, cv, match, reproj):
        self.cv, self.match, self.reproj = cv, match, reproj


def oracle_pipeline_backend(oracle):
    """A driver backend whose every numeric operator is the CPU oracle."""

    def match(feat0, feat1):
        idx, dist = oracle.knn2(feat0[1], feat1[1], nthreads=8)
        q, t, _ = oracle.ratio_filter(idx, dist, 0.70)
        return np.float32(feat0[0])[q], np.float32(feat1[0])[t]

    def reproj(r, t, K, Xf, obs):
        out = oracle.project_residual(np.hstack([np.ravel(r), np.ravel(t)])[None], K, Xf, obs, want_jac=False)
        return float(out["sumsq"][0]), out["proj"]

    return _Backend(cv=OracleCv2(oracle), match=match, reproj=reproj)
