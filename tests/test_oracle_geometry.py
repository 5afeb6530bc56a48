"""Known-answer tests that pin the oracle's geometry restatement (triangulation, Rodrigues,
projection, reprojection metric, Gauss-Newton blocks) — SURVEY.md §8c fixtures."""
import numpy as np
import pytest
from scipy.spatial.transform import Rotation

from datagen import ba_problem, decompose_P, gustav_pair, load_pose_csv


def test_pose_csv_invariants():
    """The reference artefact itself: P0 = K[I|0], rotations orthonormal, unit first baseline."""
    K, P = load_pose_csv()
    assert P.shape == (57, 3, 4)
    assert np.array_equal(P[0], K @ np.hstack([np.eye(3), np.zeros((3, 1))]))
    for k in range(57):
        R, t = decompose_P(K, P[k])
        assert np.abs(R @ R.T - np.eye(3)).max() < 1e-12 and abs(np.linalg.det(R) - 1) < 1e-12
    assert abs(np.linalg.norm(decompose_P(K, P[1])[1]) - 1) < 1e-9


@pytest.mark.parametrize("rows", [4, 6])
@pytest.mark.parametrize("k", [0, 1, 17, 40, 55])
def test_triangulation_recovers_planted_points_without_noise(oracle, rows, k):
    K, P1, P2, X, x1, x2 = gustav_pair(k, 500, 0.0, seed=k)
    X4 = oracle.triangulate(P1, P2, x1.T, x2.T, rows=rows, normalise_w=True)
    err = np.linalg.norm(X4[:3].T - X, axis=1) / np.linalg.norm(X, axis=1)
    assert np.median(err) < 2e-4 and np.all(X4[3] == 1.0)     # float32 pixel rounding only


def test_triangulation_matches_numpy_svd(oracle):
    K, P1, P2, X, x1, x2 = gustav_pair(3, 200, 0.5, seed=9)
    X4 = oracle.triangulate(P1, P2, x1.T, x2.T, rows=4)
    for i in range(0, 200, 7):
        A = np.array([x1[i, 0] * P1[2] - P1[0], x1[i, 1] * P1[2] - P1[1], x2[i, 0] * P2[2] - P2[0], x2[i, 1] * P2[2] - P2[1]],
                     dtype=np.float64)
        v = np.linalg.svd(A)[2][3]
        v = v * np.sign(v[3]) * np.sign(X4[3, i])
        assert np.allclose(X4[:, i], v, rtol=2e-6, atol=1e-9)


def test_jacobi_svd_against_lapack(oracle):
    rng = np.random.default_rng(5)
    for m, n in [(4, 4), (6, 4), (3, 3), (12, 12)]:
        A = rng.standard_normal((m, n))
        U, w, Vt = oracle.jacobi_svd(A)
        assert np.allclose(U @ np.diag(w) @ Vt, A, atol=1e-13)
        assert np.allclose(w, np.linalg.svd(A)[1], rtol=1e-13) and np.all(np.diff(w) <= 0)


def test_rodrigues_roundtrip_and_closed_form(oracle):
    rng = np.random.default_rng(6)
    for _ in range(50):
        r = rng.standard_normal(3)
        r *= rng.uniform(0.01, 3.1) / np.linalg.norm(r)
        R = oracle.rodrigues_vec2mat(r)
        assert np.allclose(R, Rotation.from_rotvec(r).as_matrix(), atol=1e-14)
        assert np.allclose(oracle.rodrigues_mat2vec(R), r, atol=1e-12)
    assert np.array_equal(oracle.rodrigues_vec2mat(np.zeros(3)), np.eye(3))
    rpi = np.array([np.pi, 0, 0])
    assert np.allclose(np.abs(oracle.rodrigues_mat2vec(oracle.rodrigues_vec2mat(rpi))), rpi, atol=1e-7)


def test_rodrigues_jacobian_vs_central_differences(oracle):
    r = np.array([0.3, -0.5, 0.8])
    _, J = oracle.rodrigues_vec2mat(r, want_jac=True)
    Jn = np.zeros((3, 9))
    for i in range(3):
        e = np.zeros(3)
        e[i] = 1e-6
        Jn[i] = ((oracle.rodrigues_vec2mat(r + e) - oracle.rodrigues_vec2mat(r - e)) / 2e-6).ravel()
    assert np.abs(J - Jn).max() < 1e-9


def test_reprojection_metric_definition(oracle):
    """sfm.py:93-97: ||f32(p) - f32(pts)||_F / N measured in the second camera."""
    K, P1, P2, X, x1, x2 = gustav_pair(5, 400, 0.3, seed=11)
    R, t = decompose_P(K, P2)
    Rt = np.hstack([R, t[:, None]])
    Xf = X.astype(np.float32)
    err, p = oracle.reprojection_error(Rt, K, Xf, x2)
    Xc = Xf.astype(np.float64) @ R.T + t
    ref = np.stack([K[0, 0] * Xc[:, 0] / Xc[:, 2] + K[0, 2], K[1, 1] * Xc[:, 1] / Xc[:, 2] + K[1, 2]], 1)
    assert np.abs(p - ref).max() < 1e-3
    assert err == pytest.approx(np.linalg.norm(p.astype(np.float64) - x2) / 400, rel=1e-12)
    assert 0.01 < err < 0.05        # sigma 0.3 px, N 400 → ~0.3*sqrt(2)/sqrt(400)


def test_sweep_blocks_match_dense_numpy_normal_equations(oracle):
    K, cams, X, obs = ba_problem(3, 40, 0.5, seed=3)
    ci = np.repeat(np.arange(3), 40).astype(np.int32)
    pi = np.tile(np.arange(40), 3).astype(np.int32)
    out = oracle.project_residual(cams, K, X, obs.reshape(-1, 2), ci, pi)

    def resid(cv, Xv):
        r = np.empty((3, 40, 2))
        for c in range(3):
            p64, _ = oracle.project_points(cv[c, :3], cv[c, 3:], K, Xv)
            r[c] = p64 - obs[c]
        return r.ravel()

    # central differences in float64 on the camera parameters (X is float32 storage: perturb via cams only)
    r0 = resid(cams, X)
    J = np.zeros((r0.size, 18))
    for j in range(18):
        d = np.zeros(18)
        d[j] = 1e-6
        J[:, j] = (resid(cams + d.reshape(3, 6), X) - resid(cams - d.reshape(3, 6), X)) / 2e-6
    for c in range(3):
        Jc = J[c * 80:(c + 1) * 80, 6 * c:6 * c + 6]
        A, B = out["JtJ_cam"][c].reshape(6, 6), Jc.T @ Jc
        assert np.abs(A - B).max() < 1e-8 * np.abs(B).max()
        g = Jc.T @ r0[c * 80:(c + 1) * 80]
        assert np.abs(out["Jtr_cam"][c] - g).max() < 1e-8 * np.abs(g).max()
    assert np.all(np.linalg.eigvalsh(out["JtJ_pt"][0].reshape(3, 3)) > 0)
    pf = out["proj"].astype(np.float32)
    assert out["sumsq"][0] == pytest.approx(((pf - obs.reshape(-1, 2)).astype(np.float64) ** 2).sum(), rel=1e-12)


def test_scoring_thresholds(oracle):
    K, cams, X, obs = ba_problem(2, 100, 1.0, seed=4, perturb=0.0)
    counts, mask = oracle.score_pnp(cams, K, X, obs[0], thr2=64.0)
    assert counts[0] == mask[0].sum() == 100 and counts[1] < 20     # camera 1's pose does not explain camera 0's pixels
    # essential matrix of a pure x-translation scores exact correspondences as inliers
    E = np.array([[0, 0, 0], [0, 0, -1], [0, 1, 0.]])
    x1 = np.random.default_rng(0).uniform(-0.3, 0.3, (50, 2))
    x2 = x1 + [0.1, 0.0]
    counts, _ = oracle.score_essential(np.stack([E, np.eye(3)]), x1, x2, 1e-8)
    assert counts.tolist()[0] == 50 and counts[1] < 50
