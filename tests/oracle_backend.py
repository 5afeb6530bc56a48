"""CPU kernel backend for sfm_mvs_amd.ransac built on the oracle — injected by tests only."""
import numpy as np


class OracleBackend:
    def __init__(self, oracle, dlt_rows=4):
        self.O = oracle
        self.dlt_rows = dlt_rows

    def prepare_essential(self, x1n, x2n):
        return np.ascontiguousarray(x1n, np.float64), np.ascontiguousarray(x2n, np.float64)

    def score_essential(self, prep, Es, thr2):
        return self.O.score_essential(Es, prep[0], prep[1], thr2)

    def recover_pose_score(self, prep, Ps, dist):
        return self.O.recover_pose_score(Ps, prep[0], prep[1], dist, self.dlt_rows)

    def prepare_pnp(self, X, uv):
        return np.ascontiguousarray(X, np.float32), np.ascontiguousarray(uv, np.float32)

    def score_pnp(self, prep, poses, K, thr2):
        return self.O.score_pnp(poses, K, prep[0], prep[1], thr2)

    def pose_sweep(self, prep, rvec, tvec, K, want_jac):
        out = self.O.project_residual(np.hstack([rvec, tvec])[None], K, prep[0], prep[1], want_jac=True)
        err = float(np.sqrt(out["res2"][0]))
        if want_jac:
            return out["JtJ_cam"][0].reshape(6, 6), out["Jtr_cam"][0], err
        return None, None, err


class OracleCv2:
    """cv2-named facade over the CPU oracle (the CPU twin of sfm_mvs_amd.cv2compat) — tests only."""
    RANSAC, NORM_L2, SOLVEPNP_ITERATIVE = 8, 4, 0

    def __init__(self, oracle, rows=4):
        from sfm_mvs_amd import hostgeom, ransac
        self.O, self.hg, self.ransac, self.rows = oracle, hostgeom, ransac, rows
        self.be = OracleBackend(oracle, rows)

    def triangulatePoints(self, P1, P2, a, b):
        return self.O.triangulate(P1, P2, np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32), rows=self.rows)

    def Rodrigues(self, src):
        src = np.asarray(src, np.float64)
        if src.size == 9:
            return self.hg.rodrigues_mat2vec(src.reshape(3, 3)).reshape(3, 1), None
        return self.hg.rodrigues_vec2mat(src.reshape(3)), None

    def convertPointsFromHomogeneous(self, src):
        from sfm_mvs_amd import cv2compat
        return cv2compat.convertPointsFromHomogeneous(src)

    def projectPoints(self, objectPoints, rvec, tvec, cameraMatrix, distCoeffs=None):
        X = np.asarray(objectPoints)
        p64, p32 = self.O.project_points(np.ravel(rvec), np.ravel(tvec), cameraMatrix, np.float32(X).reshape(-1, 3))
        return (p32 if X.dtype == np.float32 else p64).reshape(-1, 1, 2), None

    def norm(self, a, b=None, normType=4):
        d = np.asarray(a) if b is None else np.asarray(a) - np.asarray(b)      # difference in the inputs' dtype
        return float(np.sqrt(np.sum(np.float64(d) ** 2)))                        # squares accumulated in double

    def findEssentialMat(self, p1, p2, K, method=8, prob=0.999, threshold=1.0, mask=None):
        return self.ransac.find_essential_mat(p1, p2, np.asarray(K, np.float64), prob, threshold, backend=self.be)

    def recoverPose(self, E, p1, p2, K):
        return self.ransac.recover_pose(E, p1, p2, np.asarray(K, np.float64), backend=self.be)

    def solvePnPRansac(self, X, p, K, d, *a, **k):
        return self.ransac.solve_pnp_ransac(X, p, K, backend=self.be)


def oracle_pipeline_backend(oracle):
    """sfm_mvs_amd.pipeline.Backend whose every numeric operator is the CPU oracle."""
    from sfm_mvs_amd.pipeline import Backend

    def match(feat0, feat1):
        idx, dist = oracle.knn2(feat0[1], feat1[1], nthreads=8)
        q, t, _ = oracle.ratio_filter(idx, dist, 0.70)
        return np.float32(feat0[0])[q], np.float32(feat1[0])[t]

    def reproj(r, t, K, Xf, obs):
        out = oracle.project_residual(np.hstack([np.ravel(r), np.ravel(t)])[None], K, Xf, obs, want_jac=False)
        return float(out["sumsq"][0]), out["proj"]

    return Backend(cv=OracleCv2(oracle), match=match, reproj=reproj)
