"""Host-side solvers and RANSAC control flow (sfm.py:67,307,311), CPU only: the device kernels are
replaced by the oracle through dependency injection (tests/oracle_backend.py)."""
import os

import numpy as np
import pytest

from datagen import GOLDEN, decompose_P, gustav_pair, gustav_scene
from oracle_backend import OracleBackend
from sfm_mvs_amd import hostgeom as hg
from sfm_mvs_amd import ransac


def test_cv_rng_stream_is_the_mwc_generator():
    r = hg.CvRNG()
    s, out = 0xFFFFFFFFFFFFFFFF, []
    for _ in range(5):
        s = ((s & 0xFFFFFFFF) * 4164903690 + (s >> 32)) & 0xFFFFFFFFFFFFFFFF
        out.append(s & 0xFFFFFFFF)
    assert [r.next() for _ in range(5)] == out
    r = hg.CvRNG()
    assert [r.uniform(0, 100) for _ in range(6)] == [5, 4, 40, 73, 31, 12]


def test_update_num_iters_matches_the_closed_form():
    assert hg.ransac_update_num_iters(0.99, 0.5, 5, 1000) == int(np.rint(np.log(0.01) / np.log(1 - 0.5 ** 5)))
    assert hg.ransac_update_num_iters(0.99, 1.0, 5, 100) == 100
    assert hg.ransac_update_num_iters(0.999, 0.0, 5, 1000) == 0


def _norm(K, x):
    return np.stack([(x[:, 0] - K[0, 2]) / K[0, 0], (x[:, 1] - K[1, 2]) / K[1, 1]], 1).astype(np.float64)


@pytest.mark.parametrize("k", [0, 7, 33])
def test_five_point_contains_the_true_essential_matrix(k):
    K, P1, P2, X, x1, x2 = gustav_pair(k, 40, 0.0, seed=k)
    R1, t1 = decompose_P(K, P1)
    R2, t2 = decompose_P(K, P2)
    R, t = R2 @ R1.T, t2 - R2 @ R1.T @ t1
    tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    Et = tx @ R
    Et /= np.linalg.norm(Et)
    Es = hg.five_point(_norm(K, x1[:5]), _norm(K, x2[:5]))
    assert 1 <= len(Es) <= 10
    assert min(min(np.abs(E - Et).max(), np.abs(E + Et).max()) for E in Es) < 5e-3      # float32 pixels, minimal sample
    for E in Es:      # every returned model satisfies the epipolar and cubic constraints
        r = np.einsum("ni,ij,nj->n", np.c_[_norm(K, x2[:5]), np.ones(5)], E, np.c_[_norm(K, x1[:5]), np.ones(5)])
        assert np.abs(r).max() < 1e-10 and abs(np.linalg.det(E)) < 1e-10
    Ra, Rb, tt = hg.decompose_essential(Et)
    assert min(np.abs(Ra - R).max(), np.abs(Rb - R).max()) < 1e-9


def test_epnp_and_dlt_recover_a_planted_pose():
    K, P1, P2, X, x1, x2 = gustav_pair(3, 60, 0.0, seed=5)
    R, t = decompose_P(K, P2)
    Re, te = hg.epnp(K, X[:5], x2[:5].astype(np.float64))
    assert np.abs(Re - R).max() < 1e-5 and np.abs(te - t).max() < 1e-4
    rv, tv = hg.pnp_dlt_init(K, X, x2.astype(np.float64))
    assert np.abs(hg.rodrigues_vec2mat(rv) - R).max() < 1e-6 and np.abs(tv - t).max() < 1e-5
    assert np.allclose(hg.rodrigues_mat2vec(hg.rodrigues_vec2mat([0.2, -0.4, 0.9])), [0.2, -0.4, 0.9], atol=1e-14)


def test_solve_pnp_ransac_rejects_planted_outliers(oracle):
    K, P1, P2, X, x1, x2 = gustav_pair(10, 300, 0.3, seed=2)
    R, t = decompose_P(K, P2)
    rng = np.random.default_rng(0)
    bad = rng.permutation(300)[:60]
    x2 = x2.copy()
    x2[bad] += rng.uniform(30, 200, (60, 2)).astype(np.float32)
    ok, rvec, tvec, inl = ransac.solve_pnp_ransac(X.astype(np.float32), x2, K, backend=OracleBackend(oracle))
    assert ok and inl.dtype == np.int32 and inl.shape[1] == 1
    assert not set(inl[:, 0]) & set(bad) and len(inl) >= 230
    assert np.abs(hg.rodrigues_vec2mat(rvec.ravel()) - R).max() < 2e-3 and np.abs(tvec.ravel() - t).max() < 2e-2


def test_essential_ransac_and_recover_pose(oracle):
    K, P1, P2, X, x1, x2 = gustav_pair(0, 400, 0.2, seed=3)
    rng = np.random.default_rng(1)
    bad = rng.permutation(400)[:80]
    x2 = x2.copy()
    x2[bad] += rng.uniform(20, 100, (80, 2)).astype(np.float32)
    be = OracleBackend(oracle)
    E, mask = ransac.find_essential_mat(x1, x2, K, 0.999, 0.4, backend=be)
    assert mask.shape == (400, 1) and mask.dtype == np.uint8 and set(np.unique(mask)) <= {0, 1}
    assert mask[bad].sum() <= 2 and mask.sum() > 150
    sel = mask.ravel() == 1
    good, R, t, m2 = ransac.recover_pose(E, x1[sel], x2[sel], K, backend=be)
    assert set(np.unique(m2)) <= {0, 255} and good == (m2 > 0).sum() > 150
    Rt, tt = decompose_P(K, P2)           # P1 is the identity camera for pair 0
    # E is the best MINIMAL-sample model (OpenCV does not refine it): loose pose tolerance under 0.2 px noise
    assert np.abs(R - Rt).max() < 5e-2 and np.abs(t.ravel() - tt / np.linalg.norm(tt)).max() < 0.15


def test_common_points_mirror_matches_reference_vectors():
    from sfm_mvs_amd.pipeline import common_points
    z = np.load(os.path.join(GOLDEN, "common_points.npz"))
    for name in "abcd":
        i1, i2, t1, t2 = common_points(z[f"{name}_pts1"], z[f"{name}_pts2"], z[f"{name}_pts3"])
        assert np.array_equal(i1, z[f"{name}_indx1"]) and np.array_equal(i2, z[f"{name}_indx2"])
        assert np.array_equal(t1, z[f"{name}_temp1"].reshape(-1, 2)) and np.array_equal(t2, z[f"{name}_temp2"].reshape(-1, 2))


def test_to_ply_is_byte_identical_to_the_reference_output(tmp_path):
    from sfm_mvs_amd.pipeline import to_ply
    z = np.load(os.path.join(GOLDEN, "to_ply.npz"))
    os.makedirs(tmp_path / "Point_Cloud")
    to_ply(str(tmp_path), z["points"], z["colors"], False)
    assert open(tmp_path / "Point_Cloud" / "sparse.ply").read() == str(z["ply_text"])


def test_scene_generator_is_consistent():
    K, P, feats, ids = gustav_scene(4, seed=1)
    assert len(feats) == 4 and all(f[0].dtype == np.float32 and f[1].shape[1] == 128 for f in feats)
    common = np.intersect1d(ids[1][ids[1] >= 0], ids[2][ids[2] >= 0])
    assert len(common) > 100


def test_cpp_epnp_agrees_with_the_numpy_restatement():
    """sfm_host_epnp (C++) vs hostgeom.epnp_numpy.  On exact correspondences both return the same pose; on noisy minimal
    samples EPnP's three beta approximations + 5 Gauss-Newton steps can settle on different, equally good solutions
    (eigenvector bases of near-equal eigenvalues differ between the two eigen-solvers), so there the two are compared
    against the ground truth instead of against each other."""
    K, P1, P2, X, x1, x2 = gustav_pair(3, 400, 0.0, seed=11)
    R, t = decompose_P(K, P2)
    rng = np.random.default_rng(0)
    for trial in range(100):
        n = 5 if trial < 70 else int(rng.integers(6, 40))
        sel = rng.choice(len(X), n, replace=False)
        Rc, tc = hg.epnp(K, X[sel], x2[sel].astype(np.float64))
        Rn, tn = hg.epnp_numpy(K, X[sel], x2[sel].astype(np.float64))
        assert abs(np.linalg.det(Rc) - 1) < 1e-9 and np.allclose(Rc @ Rc.T, np.eye(3), atol=1e-9)
        assert np.abs(Rc - Rn).max() < 1e-6 and np.abs(tc - tn).max() < 1e-5 * max(1.0, np.abs(tn).max()), trial
    K, P1, P2, X, x1, x2 = gustav_pair(3, 400, 0.3, seed=11)
    ec, en = [], []
    for trial in range(200):
        n = 5 if trial < 150 else int(rng.integers(6, 40))
        sel = rng.choice(len(X), n, replace=False)
        Rc, tc = hg.epnp(K, X[sel], x2[sel].astype(np.float64))
        Rn, tn = hg.epnp_numpy(K, X[sel], x2[sel].astype(np.float64))
        assert np.isfinite(Rc).all() and abs(np.linalg.det(Rc) - 1) < 1e-9
        ec.append(np.abs(Rc - R).max())
        en.append(np.abs(Rn - R).max())
    assert 0.7 < np.median(ec) / np.median(en) < 1.4 and np.percentile(ec, 90) < 1.5 * np.percentile(en, 90)
