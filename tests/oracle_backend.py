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
