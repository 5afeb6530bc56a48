"""HIP path vs oracle for triangulation (sfm.py:53-54), reprojection error (sfm.py:79-100), the
Gauss-Newton sweep (sfm.py:104-157) and RANSAC scoring (sfm.py:67,307), through the C-ABI.
Tolerances: north_star asks 1e-4 relative; these kernels mirror the oracle's operation order so the
tests hold them to 1e-6 (float32 outputs) / 1e-9 (fp64 sums); integer masks bit-exact."""
import numpy as np
import pytest
import torch

from datagen import ba_problem, decompose_P, gustav_pair

pytestmark = pytest.mark.gpu


def cu(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return (t if dtype is None else t.to(dtype)).cuda()


@pytest.mark.parametrize("rows", [4, 6])
@pytest.mark.parametrize("k,n,sigma", [(0, 1, 0.3), (0, 700, 0.0), (1, 1500, 0.3), (30, 257, 1.0), (55, 64, 0.3)])
def test_triangulate_matches_oracle(hip, oracle, rows, k, n, sigma):
    K, P1, P2, X, x1, x2 = gustav_pair(k, n, sigma, seed=100 + k)
    want = oracle.triangulate(P1, P2, x1.T, x2.T, rows=rows, normalise_w=True)
    # the reference passes transposed VIEWS of (N,2) arrays (sfm.py:47-48): do the same
    got = hip.triangulate(P1, P2, cu(x1).t(), cu(x2).t(), rows=rows, normalise_w=True).cpu().numpy()
    assert got.shape == (4, n) and np.all(got[3] == 1.0)
    assert np.allclose(got, want, rtol=1e-6, atol=1e-7)
    assert (got == want).mean() > 0.99          # same operation order: almost always bit-identical
    raw = hip.triangulate(P1, P2, cu(np.ascontiguousarray(x1.T)), cu(np.ascontiguousarray(x2.T)), rows=rows).cpu().numpy()
    want_raw = oracle.triangulate(P1, P2, x1.T, x2.T, rows=rows, normalise_w=False)
    assert np.allclose(np.abs(raw), np.abs(want_raw), rtol=1e-6, atol=1e-9)
    assert np.allclose(np.linalg.norm(raw.astype(np.float64), axis=0), 1.0, atol=1e-6)


@pytest.mark.parametrize("k,n,sigma", [(1, 4000, 0.0), (1, 4000, 0.3), (30, 2500, 1.0), (50, 3000, 3.0), (10, 1000, 0.05)])
def test_fast_triangulation_equals_the_faithful_path(hip, oracle, k, n, sigma):
    """normalise_w="fast" (inverse iteration on A^T A) against the OpenCV-faithful oracle: north_star asks for 1e-4
    relative; it is bit-identical on almost every point and within one float32 ulp on the rest."""
    K, P1, P2, X, x1, x2 = gustav_pair(k, n, sigma, seed=100 + k)
    want = oracle.triangulate(P1, P2, x1.T, x2.T, rows=4, normalise_w=True)
    got = hip.triangulate(P1, P2, cu(x1).t(), cu(x2).t(), rows=4, normalise_w="fast").cpu().numpy()
    assert got.shape == (4, n) and np.all(got[3] == 1.0)
    assert np.allclose(got, want, rtol=3e-7, atol=0)
    assert (got == want).all(0).mean() > 0.995


def test_fast_triangulation_degenerate_inputs_fall_back(hip, oracle):
    """Identical cameras / coincident rays (rank-deficient systems) and a zero start-vector overlap must not produce
    NaNs or hang: such lanes run the Jacobi sweeps and give the faithful path's answer."""
    K, P1, P2, X, x1, x2 = gustav_pair(1, 256, 0.3, seed=5)
    same = hip.triangulate(P1, P1, cu(x1).t(), cu(x1).t(), rows=4, normalise_w="fast").cpu().numpy()
    ref = hip.triangulate(P1, P1, cu(x1).t(), cu(x1).t(), rows=4, normalise_w=True).cpu().numpy()
    assert np.array_equal(np.isfinite(same), np.isfinite(ref))
    ok = np.isfinite(ref).all(0)
    # a rank-2 system has a 2-D null space: any vector of it is "the" answer, so compare reprojections, not coordinates
    p = P1 @ same[:, ok].astype(np.float64)
    q = P1 @ ref[:, ok].astype(np.float64)
    assert np.allclose(p[:2] / p[2], q[:2] / q[2], atol=1e-2)


def test_triangulate_large_roundtrip_property(hip):
    """1e6 correspondences (north-star synthetic): triangulate → reproject reproduces the pixels."""
    K, P1, P2, X, x1, x2 = gustav_pair(1, 4000, 0.0, seed=2)
    reps = 250
    x1b, x2b = np.tile(x1, (reps, 1)), np.tile(x2, (reps, 1))
    X4 = hip.triangulate(P1, P2, cu(x1b).t(), cu(x2b).t(), normalise_w=True)
    torch.cuda.synchronize()
    X4 = X4.cpu().numpy().astype(np.float64)
    assert np.array_equal(X4[:, :4000], X4[:, -4000:])          # same input → same output anywhere in the grid
    p = P2 @ X4
    assert np.abs(p[:2] / p[2] - x2b.T).max() < 0.05


def test_reprojection_error_matches_oracle(hip, oracle):
    for k, n, sigma in [(1, 900, 0.3), (20, 333, 1.0), (40, 5, 0.0)]:
        K, P1, P2, X, x1, x2 = gustav_pair(k, n, sigma, seed=k)
        R, t = decompose_P(K, P2)
        Rt = np.hstack([R, t[:, None]])
        Xf = X.astype(np.float32)
        want, wp = oracle.reprojection_error(Rt, K, Xf, x2)
        rvec = oracle.rodrigues_mat2vec(R)
        out = hip.project_residual(cu(np.hstack([rvec, t])[None]), K, cu(Xf), cu(x2))
        got = float(np.sqrt(out["sumsq"].item()) / n)
        assert got == pytest.approx(want, rel=1e-9)
        assert np.allclose(out["proj"].cpu().numpy(), wp, rtol=1e-6, atol=1e-4)


def test_single_camera_normal_equations_and_inliers(hip, oracle):
    K, cams, X, obs = ba_problem(2, 3000, 3.0, seed=12)
    want = oracle.project_residual(cams[:1], K, X, obs[0], thr2=64.0)
    out = hip.project_residual(cu(cams[:1]), K, cu(X), cu(obs[0]), thr2=64.0, want_inlier=True, want_jac=True,
                               want_pt_jac=True)
    assert np.array_equal(out["inlier"].cpu().numpy(), want["inlier"])          # integer mask: bit-exact
    assert 0 < want["inlier"].sum() < 3000
    for key in ("JtJ_cam", "Jtr_cam", "JtJ_pt", "Jtr_pt"):
        g, w = out[key].cpu().numpy(), want[key]
        assert np.abs(g - w).max() <= 1e-10 * np.abs(w).max(), key
    assert out["sumsq"].item() == pytest.approx(want["sumsq"][0], rel=1e-12)
    # deterministic: the single-camera path uses no atomics
    out2 = hip.project_residual(cu(cams[:1]), K, cu(X), cu(obs[0]), want_jac=True)
    assert torch.equal(out2["JtJ_cam"], out["JtJ_cam"]) and torch.equal(out2["Jtr_cam"], out["Jtr_cam"])


def test_indexed_multi_camera_sweep(hip, oracle):
    K, cams, X, obs = ba_problem(7, 500, 1.0, seed=13)
    rng = np.random.default_rng(0)
    sel = rng.permutation(7 * 500)[:2100]                         # sparse visibility, arbitrary order
    ci, pi = (sel // 500).astype(np.int32), (sel % 500).astype(np.int32)
    o = obs.reshape(-1, 2)[sel]
    want = oracle.project_residual(cams, K, X, o, ci, pi)
    out = hip.project_residual(cu(cams), K, cu(X), cu(o), cu(ci), cu(pi), want_jac=True, want_pt_jac=True)
    for key in ("JtJ_cam", "Jtr_cam", "JtJ_pt", "Jtr_pt"):
        g, w = out[key].cpu().numpy(), want[key]
        assert np.abs(g - w).max() <= 1e-9 * np.abs(w).max(), key
    assert out["sumsq"].item() == pytest.approx(want["sumsq"][0], rel=1e-12)
    assert np.allclose(out["proj"].cpu().numpy(), want["proj"], rtol=1e-6, atol=1e-4)


@pytest.mark.parametrize("ncam,npt", [(3, 100), (17, 1000), (40, 5000)])
def test_dense_sweep_matches_oracle(hip, oracle, ncam, npt):
    K, cams, X, obs = ba_problem(ncam, npt, 0.5, seed=ncam)
    ci = np.repeat(np.arange(ncam), npt).astype(np.int32)
    pi = np.tile(np.arange(npt), ncam).astype(np.int32)
    want = oracle.project_residual(cams, K, X, obs.reshape(-1, 2), ci, pi)
    out = hip.ba_dense_sweep(cu(cams), K, cu(X), cu(obs))
    for key in ("JtJ_cam", "Jtr_cam", "JtJ_pt", "Jtr_pt"):
        g, w = out[key].cpu().numpy(), want[key]
        assert np.abs(g - w).max() <= 1e-10 * np.abs(w).max(), key
    assert out["sumsq"].item() == pytest.approx(want["sumsq"][0], rel=1e-12)
    out2 = hip.ba_dense_sweep(cu(cams), K, cu(X), cu(obs))
    for key in out:
        assert torch.equal(out[key], out2[key]), key                # fixed-order reductions: reproducible


def test_dense_sweep_linearity_property(hip):
    """Size-independent property at a larger size: the blocks of a camera subset equal the subset's own sweep."""
    K, cams, X, obs = ba_problem(24, 20000, 0.5, seed=5)
    full = hip.ba_dense_sweep(cu(cams), K, cu(X), cu(obs))
    half = hip.ba_dense_sweep(cu(cams[:12]), K, cu(X), cu(obs[:12]))
    rest = hip.ba_dense_sweep(cu(cams[12:]), K, cu(X), cu(obs[12:]))
    assert torch.allclose(full["JtJ_cam"][:12], half["JtJ_cam"], rtol=1e-12, atol=0)
    assert torch.allclose(full["JtJ_pt"], half["JtJ_pt"] + rest["JtJ_pt"], rtol=1e-10, atol=1e-6)
    assert full["sumsq"].item() == pytest.approx(half["sumsq"].item() + rest["sumsq"].item(), rel=1e-12)


def test_ransac_scoring_masks_bit_exact(hip, oracle):
    K, cams, X, obs = ba_problem(9, 777, 4.0, seed=21, perturb=0.002)
    wc, wm = oracle.score_pnp(cams, K, X, obs[4], thr2=64.0)
    gc, gm = hip.score_pnp(cu(cams), K, cu(X), cu(obs[4]), 64.0, want_mask=True)
    assert np.array_equal(gc.cpu().numpy(), wc) and np.array_equal(gm.cpu().numpy(), wm)
    assert 0 < wc[4] <= 777
    rng = np.random.default_rng(3)
    x1 = rng.uniform(-0.4, 0.4, (1000, 2))
    x2 = x1 + [0.05, 0.0] + rng.normal(0, 2e-4, (1000, 2))
    E = np.array([[0, 0, 0], [0, 0, -1], [0, 1, 0.]])
    Es = np.stack([E + rng.normal(0, s, (3, 3)) for s in (0, 1e-4, 1e-3, 1e-2, 0.1)])
    thr2 = np.float32((0.4 / 1198.0) ** 2)
    wc, wm = oracle.score_essential(Es, x1, x2, thr2)
    gc, gm = hip.score_essential(cu(Es), cu(x1), cu(x2), thr2, want_mask=True)
    assert np.array_equal(gc.cpu().numpy(), wc) and np.array_equal(gm.cpu().numpy(), wm)
    assert wc[0] > wc[-1]


def test_empty_and_degenerate_inputs(hip, oracle):
    """Edge cases the reference can hit: no correspondences, a single one, points behind the camera."""
    K, P1, P2, X, x1, x2 = gustav_pair(1, 4, 0.0, seed=1)
    e = torch.empty((2, 0), dtype=torch.float32, device="cuda")
    assert hip.triangulate(P1, P2, e, e).shape == (4, 0)
    one = hip.triangulate(P1, P2, cu(x1[:1]).t(), cu(x2[:1]).t(), normalise_w=True).cpu().numpy()
    assert np.allclose(one, oracle.triangulate(P1, P2, x1[:1].T, x2[:1].T, normalise_w=True), rtol=1e-6)
    R, t = decompose_P(K, P2)
    cams = cu(np.hstack([oracle.rodrigues_mat2vec(R), t])[None])
    out = hip.project_residual(cams, K, torch.empty((0, 3), dtype=torch.float32, device="cuda"),
                               torch.empty((0, 2), dtype=torch.float32, device="cuda"))
    assert out["sumsq"].item() == 0.0 and out["proj"].shape == (0, 2)
    # a point exactly in the camera plane (Z' = 0): OpenCV's `z ? 1/z : 1` branch, same on both sides
    Xw = np.float32(R.T @ (np.array([0.3, -0.2, 0.0]) - t))[None]
    pz = hip.project_residual(cams, K, cu(Xw), cu(np.zeros((1, 2), np.float32)))["proj"].cpu().numpy()
    _, wz = oracle.project_points(oracle.rodrigues_mat2vec(R), t, K, Xw)
    assert np.all(np.isfinite(pz)) and np.allclose(pz, wz, rtol=1e-3, atol=0.5)
    counts = hip.score_pnp(cams, K, torch.empty((0, 3), dtype=torch.float32, device="cuda"),
                           torch.empty((0, 2), dtype=torch.float32, device="cuda"))
    assert counts.cpu().tolist() == [0]


def test_block_bundle_adjustment_reduces_cost_to_the_noise_floor(hip):
    """an optimiser on top of the sweep.  Perturbed cameras/points of a dense problem are pulled back: the
    fp64 cost must fall monotonically towards its noise floor, in the dense and the indexed form alike."""
    from sfm_mvs_amd import ba
    K, cams, X, obs = ba_problem(6, 800, 0.3, seed=8, perturb=0.01)
    c0, x0 = cu(cams), cu(X)
    cams1, X1, hist = ba.bundle_adjust(c0, K, x0, cu(obs), iters=12)
    assert all(b <= a * (1 + 1e-12) for a, b in zip(hist, hist[1:])) and hist[-1] < 0.1 * hist[0]
    noise_floor = 2 * 6 * 800 * 0.3 ** 2          # E[sum of squared residuals] at the true parameters
    assert hist[-1] < noise_floor
    ci = np.repeat(np.arange(6), 800).astype(np.int32)
    pi = np.tile(np.arange(800), 6).astype(np.int32)
    _, _, hist2 = ba.bundle_adjust(c0, K, x0, cu(obs.reshape(-1, 2)), cu(ci), cu(pi), iters=12)
    assert hist2[-1] == pytest.approx(hist[-1], rel=1e-6)


# ---------------------------------------------------------------- Schur-complement products and solver
def _numpy_pair_jacobians(cams, K, X):
    """Central-difference Jacobians of the float64 projection: Jc [ncam,npt,2,6], Jp [ncam,npt,2,3]."""
    from scipy.spatial.transform import Rotation

    def proj(c, Xs):
        R = Rotation.from_rotvec(c[:3]).as_matrix()
        Xc = Xs @ R.T + c[3:]
        return np.stack([K[0, 0] * Xc[:, 0] / Xc[:, 2] + K[0, 2], K[1, 1] * Xc[:, 1] / Xc[:, 2] + K[1, 2]], 1)

    ncam, npt = len(cams), len(X)
    X = X.astype(np.float64)
    Jc, Jp = np.empty((ncam, npt, 2, 6)), np.empty((ncam, npt, 2, 3))
    for i, c in enumerate(cams):
        for a in range(6):
            h = 1e-6 * max(1.0, abs(c[a]))
            d = np.zeros(6); d[a] = h
            Jc[i, :, :, a] = (proj(c + d, X) - proj(c - d, X)) / (2 * h)
        for a in range(3):
            h = 1e-6
            d = np.zeros(3); d[a] = h
            Jp[i, :, :, a] = (proj(c, X + d) - proj(c, X - d)) / (2 * h)
    return Jc, Jp


def test_schur_products_match_finite_difference_jacobians(hip):
    K, cams, X, obs = ba_problem(9, 700, 0.5, seed=31)
    Jc, Jp = _numpy_pair_jacobians(cams, K, X)
    rng = np.random.default_rng(1)
    x, v = rng.standard_normal((9, 6)), rng.standard_normal((700, 3))
    want_u = np.einsum("ijkb,ijk->jb", Jp, np.einsum("ijka,ia->ijk", Jc, x))        # W^T x
    want_w = np.einsum("ijka,ijk->ia", Jc, np.einsum("ijkb,jb->ijk", Jp, v))        # W v
    u = hip.ba_schur_wt(cu(cams), K, cu(X), cu(x))
    w = hip.ba_schur_w(cu(cams), K, cu(X), cu(v))
    assert np.abs(u.cpu().numpy() - want_u).max() <= 1e-6 * np.abs(want_u).max()
    assert np.abs(w.cpu().numpy() - want_w).max() <= 1e-6 * np.abs(want_w).max()
    # adjointness  <v, W^T x> = <x, W v>  to rounding, and reproducibility of the fixed-order reductions
    assert float((u.cpu() * torch.from_numpy(v)).sum()) == pytest.approx(float((w.cpu() * torch.from_numpy(x)).sum()), rel=1e-11)
    assert torch.equal(u, hip.ba_schur_wt(cu(cams), K, cu(X), cu(x))) and torch.equal(w, hip.ba_schur_w(cu(cams), K, cu(X), cu(v)))


def test_schur_step_equals_dense_normal_equation_solve(hip):
    """The PCG solution of the reduced camera system + back-substitution = the solve of the full damped normal
    equations assembled (on the host, from the sweep's own blocks and finite-difference W) for a small problem."""
    from sfm_mvs_amd import ba
    ncam, npt, lam = 5, 80, 1e-2
    K, cams, X, obs = ba_problem(ncam, npt, 0.5, seed=32)
    blocks = hip.ba_dense_sweep(cu(cams), K, cu(X), cu(obs))
    dc, dp, ncg = ba.schur_step(cu(cams), K, cu(X), blocks, lam, fix_first_camera=False, cg_tol=1e-13)
    Jc, Jp = _numpy_pair_jacobians(cams, K, X)
    n = 6 * ncam + 3 * npt
    H = np.zeros((n, n))
    B = blocks["JtJ_cam"].cpu().numpy().reshape(ncam, 6, 6)
    C = blocks["JtJ_pt"].cpu().numpy().reshape(npt, 3, 3)
    for i in range(ncam):
        H[6 * i:6 * i + 6, 6 * i:6 * i + 6] = B[i] + lam * np.diag(np.diag(B[i]))
    for j in range(npt):
        o = 6 * ncam + 3 * j
        H[o:o + 3, o:o + 3] = C[j] + lam * np.diag(np.diag(C[j]))
    W = np.einsum("ijka,ijkb->ijab", Jc, Jp)
    for i in range(ncam):
        for j in range(npt):
            o = 6 * ncam + 3 * j
            H[6 * i:6 * i + 6, o:o + 3] = W[i, j]
            H[o:o + 3, 6 * i:6 * i + 6] = W[i, j].T
    g = np.hstack([blocks["Jtr_cam"].cpu().numpy().ravel(), blocks["Jtr_pt"].cpu().numpy().ravel()])
    sol = np.linalg.solve(H, g)
    assert ncg <= 6 * ncam + 2
    assert np.abs(dc.cpu().numpy().ravel() - sol[:6 * ncam]).max() <= 2e-5 * np.abs(sol[:6 * ncam]).max()
    assert np.abs(dp.cpu().numpy().ravel() - sol[6 * ncam:]).max() <= 2e-5 * np.abs(sol[6 * ncam:]).max()


def test_device_pcg_equals_the_host_driven_recurrence(hip):
    """sfm_ba_schur_solve (the whole PCG recurrence on the device) against the same recurrence driven from torch ops: same
    iteration count, steps equal to CG rounding, identical Levenberg-Marquardt cost histories; a singular camera block is
    reported instead of silently producing inf / NaN."""
    from sfm_mvs_amd import ba
    ncam, npt, lam = 9, 2500, 3e-3
    K, cams, X, obs = ba_problem(ncam, npt, 0.5, seed=35, perturb=0.01)
    blocks = hip.ba_dense_sweep(cu(cams), K, cu(X), cu(obs))
    d1 = ba.schur_step(cu(cams), K, cu(X), blocks, lam, device_pcg=True)
    d2 = ba.schur_step(cu(cams), K, cu(X), blocks, lam, device_pcg=False)
    assert d1[2] == d2[2] and d1[2] > 0
    for a, b in zip(d1[:2], d2[:2]):
        assert float((a - b).abs().max()) <= 1e-8 * float(b.abs().max())
    assert torch.equal(d1[0], ba.schur_step(cu(cams), K, cu(X), blocks, lam)[0])          # fixed-order reductions: deterministic
    real = ba.schur_step
    h_dev = ba.bundle_adjust_schur(cu(cams), K, cu(X), cu(obs), iters=6)[2]
    try:
        ba.schur_step = lambda *a, **k: real(*a, **dict(k, device_pcg=False))
        h_host = ba.bundle_adjust_schur(cu(cams), K, cu(X), cu(obs), iters=6)[2]
    finally:
        ba.schur_step = real
    assert len(h_dev) == len(h_host) and np.allclose(h_dev, h_host, rtol=1e-7)
    dead = {k: v.clone() for k, v in blocks.items()}
    dead["JtJ_cam"][3] = 0
    dc, dp, it, status = hip.ba_schur_solve(cu(cams), K, cu(X), dead, 0.0)
    assert status & 1
    with pytest.raises(hip.SfmHipError):
        ba.schur_step(cu(cams), K, cu(X), dead, 0.0)


def test_schur_lm_converges_to_the_noise_floor(hip):
    """Joint Schur-complement LM vs the alternating block updates on the same problem and iteration budget."""
    from sfm_mvs_amd import ba
    ncam, npt, sigma = 12, 3000, 0.5
    K, cams, X, obs = ba_problem(ncam, npt, sigma, seed=33, perturb=0.01)
    c1, x1, h1 = ba.bundle_adjust_schur(cu(cams), K, cu(X), cu(obs), iters=8)
    c2, x2, h2 = ba.bundle_adjust(cu(cams), K, cu(X), cu(obs), iters=8)
    floor = 2 * ncam * npt * sigma ** 2                               # E[sum of squared noise]
    assert all(b <= a for a, b in zip(h1, h1[1:]))
    assert h1[-1] < 1.05 * floor and h1[-1] < 0.3 * h1[0]
    assert h1[-1] <= h2[-1] * 1.0001                                   # the joint step is at least as good


def test_sparse_schur_products_and_solver(hip):
    """Indexed visibility (each observation names its camera and point): products against the finite-difference
    Jacobians restricted to the visible pairs; the LM solver reaches the noise floor of the visible observations."""
    from sfm_mvs_amd import ba
    ncam, npt, sigma = 8, 900, 0.5
    K, cams, X, obs = ba_problem(ncam, npt, sigma, seed=41, perturb=0.01)
    rng = np.random.default_rng(2)
    vis = rng.random((ncam, npt)) < 0.45
    vis[:3] = True                                                    # every point is seen at least three times
    ci, pi = np.nonzero(vis)
    ci, pi = ci.astype(np.int32), pi.astype(np.int32)
    o = obs[ci, pi]
    Jc, Jp = _numpy_pair_jacobians(cams, K, X)
    x, v = rng.standard_normal((ncam, 6)), rng.standard_normal((npt, 3))
    tu = np.einsum("oka,oa->ok", Jc[ci, pi], x[ci])
    want_u = np.zeros((npt, 3)); np.add.at(want_u, pi, np.einsum("okb,ok->ob", Jp[ci, pi], tu))
    sv = np.einsum("okb,ob->ok", Jp[ci, pi], v[pi])
    want_w = np.zeros((ncam, 6)); np.add.at(want_w, ci, np.einsum("oka,ok->oa", Jc[ci, pi], sv))
    u = hip.ba_schur_wt(cu(cams), K, cu(X), cu(x), cu(ci), cu(pi)).cpu().numpy()
    w = hip.ba_schur_w(cu(cams), K, cu(X), cu(v), cu(ci), cu(pi)).cpu().numpy()
    assert np.abs(u - want_u).max() <= 1e-6 * np.abs(want_u).max()
    assert np.abs(w - want_w).max() <= 1e-6 * np.abs(want_w).max()
    c1, x1, h1 = ba.bundle_adjust_schur(cu(cams), K, cu(X), cu(o), cu(ci), cu(pi), iters=8)
    floor = 2 * len(ci) * sigma ** 2
    assert all(b <= a for a, b in zip(h1, h1[1:])) and h1[-1] < 1.05 * floor and h1[-1] < 0.3 * h1[0]


def test_triangulation_one_million_distinct_points_vs_oracle(hip, oracle):
    """The north-star synthetic at 1e6 DISTINCT correspondences (pose.csv cameras 1, 2, points uniform in the sparse.ply
    bounding box, sigma 0.3 px): every lane runs its own Jacobi sweep count, and the result is held to the oracle point
    by point — the faithful path bit for bit on >= 99 % of the points (<= 1e-6 otherwise), the fast path (inverse
    iteration) within one float32 ulp of the faithful one."""
    from datagen import load_pose_csv
    K, P = load_pose_csv()
    n = 1_000_000
    rng = np.random.default_rng(2)
    X = np.stack([rng.uniform(-6.3, 3.6, n), rng.uniform(-2.6, 5.0, n), rng.uniform(3.2, 13.0, n)], 1)
    Xh = np.c_[X, np.ones(n)].T
    xs = []
    for Pm in (P[1], P[2]):
        x = Pm @ Xh
        xs.append(((x[:2] / x[2]).T + rng.normal(0, 0.3, (n, 2))).astype(np.float32))
    want = oracle.triangulate(P[1], P[2], np.ascontiguousarray(xs[0].T), np.ascontiguousarray(xs[1].T), normalise_w=True)
    a, b = cu(xs[0]).t(), cu(xs[1]).t()
    got = hip.triangulate(P[1], P[2], a, b, normalise_w=True).cpu().numpy()
    same = (got == want).all(0)
    assert same.mean() >= 0.99, same.mean()
    scale = np.abs(want).max(0)
    assert (np.abs(got - want).max(0) / scale).max() <= 1e-6
    fast = hip.triangulate(P[1], P[2], a, b, normalise_w="fast").cpu().numpy()
    assert (fast == got).all(0).mean() >= 0.995
    assert (np.abs(fast - got).max(0) / scale).max() <= 3e-7
    # the guarded fast path (what the driver uses): the fast result where its float32 casts cannot differ, the Jacobi sweeps
    # (second, compacted pass) elsewhere — EVERY point equal to the faithful path, bit for bit
    guarded = hip.triangulate(P[1], P[2], a, b, normalise_w="guarded").cpu().numpy()
    assert np.array_equal(guarded.view(np.uint32), got.view(np.uint32))
    # ... also where the geometry is nearly degenerate (baseline 1e-4 of the depth: the inverse iteration converges slowly or
    # not at all and most points take the fallback) and on strided (N, 2) inputs
    P2n = P[1].copy()
    P2n[:, 3] += P[1][:, :3] @ np.array([1e-3, 0.0, 0.0])
    xb = (P2n @ Xh)
    xb = ((xb[:2] / xb[2]).T + rng.normal(0, 0.3, (n, 2))).astype(np.float32)
    m = 300_000          # (>= 2^18: the first pass keeps a compact reject list; here nearly every point is rejected, the per-workgroup
                         #  queues overflow and the second pass falls back to scanning the marks)
    a2, b2 = cu(xs[0][:m]).t(), cu(xb[:m]).t()
    assert np.array_equal(hip.triangulate(P[1], P2n, a2, b2, normalise_w="guarded").cpu().numpy().view(np.uint32),
                          hip.triangulate(P[1], P2n, a2, b2, normalise_w=True).cpu().numpy().view(np.uint32))


def test_block_kernels_and_norm(hip):
    """sfm_block_inverse / sfm_block_matvec (the Schur solver's batched 3x3 and 6x6 blocks) and sfm_norm_l2 (cv2.norm)."""
    rng = np.random.default_rng(8)
    for k, n in ((3, 5000), (6, 777), (6, 1), (3, 0)):
        J = rng.normal(size=(n, 2 * k, k))
        A = np.einsum("nij,nik->njk", J, J) + 1e-3 * np.eye(k)          # SPD like the normal-equation blocks
        inv = hip.block_inverse(cu64(A.reshape(n, k * k)), k).cpu().numpy().reshape(n, k, k)
        if n:
            assert np.abs(inv @ A - np.eye(k)).max() < 1e-8
            assert np.abs(inv - np.linalg.inv(A)).max() <= 1e-9 * np.abs(np.linalg.inv(A)).max()
        x = rng.normal(size=(n, k))
        y = hip.block_matvec(cu64(A.reshape(n, k * k)), cu64(x), k).cpu().numpy()
        assert np.allclose(y, np.einsum("nij,nj->ni", A, x), rtol=1e-13, atol=1e-13)
    general = rng.normal(size=(100, 6, 6))                                # needs the pivoting
    general[:, 0, 0] = 0
    inv = hip.block_inverse(cu64(general.reshape(100, 36)), 6).cpu().numpy().reshape(100, 6, 6)
    assert np.abs(inv @ general - np.eye(6)).max() < 1e-8
    # singular / non-finite blocks are REPORTED through the device status word (sfm_block_inverse_checked): the wrapper raises
    bad = general.copy()
    bad[7] = 0.0                                                          # a camera without observations
    bad[11, :, 2] = bad[11, :, 1]                                         # rank 5
    bad[13, 2, 2] = np.nan
    with pytest.raises(hip.SfmHipError, match="singular or non-finite"):
        hip.block_inverse(cu64(bad.reshape(100, 36)), 6)
    inv_b, cnt = hip.block_inverse(cu64(bad.reshape(100, 36)), 6, check_singular=False)
    assert int(cnt.item()) >= 2                                           # (the rank-5 block may escape by rounding; zeros and NaNs never do)
    ok = np.ones(100, bool); ok[[7, 11, 13]] = False
    assert np.abs(inv_b.cpu().numpy().reshape(100, 6, 6)[ok] @ general[ok] - np.eye(6)).max() < 1e-8
    for dt in (np.float32, np.float64):
        a, b = rng.normal(0, 30, 100_003).astype(dt), rng.normal(0, 30, 100_003).astype(dt)
        want = np.sqrt(np.sum(np.float64(a - b) ** 2))
        got = hip.norm_l2(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()).item()
        assert got == pytest.approx(want, rel=1e-13)
        assert hip.norm_l2(torch.from_numpy(a).cuda()).item() == pytest.approx(np.sqrt(np.sum(np.float64(a) ** 2)), rel=1e-13)
    assert hip.norm_l2(torch.zeros(0, device="cuda")).item() == 0.0


def cu64(a):
    return torch.from_numpy(np.ascontiguousarray(a, np.float64)).cuda()


def test_matches_to_points_in_one_pass_equals_the_separate_operators(hip, oracle):
    """sfm_triangulate_matches_batch (sfm.py:262-268 then :53-54 for up to 8 pairs per call) against the chain it fuses —
    the oracle's Lowe loop, a gather, the faithful triangulation: survivor counts, column order (ascending queryIdx), every
    point bit for bit (the guarded fast path incl. its in-kernel Jacobi redo), zeroed tails.  Pairs of different sizes in
    one call, a query without a second neighbour, distances exactly ON the ratio boundary (strict <), a pair with no
    survivor, a pair whose every query survives, and a nearly degenerate camera pair (most points take the redo path)."""
    from datagen import load_pose_csv
    K, P = load_pose_csv()
    rng = np.random.default_rng(31)
    cap = 5000
    sizes = [5000, 4097, 1024, 1, 0, 3000, 2500, 777]
    P2n = P[3].copy()
    P2n[:, 3] += P[3][:, :3] @ np.array([1e-3, 0.0, 0.0])                # baseline 1e-3: ill-conditioned null vectors
    cams = [(P[1], P[2]), (P[5], P[6]), (P[10], P[11]), (P[0], P[1]), (P[0], P[1]), (P[3], P2n), (P[20], P[21]), (P[30], P[31])]
    blocks, kp0s, kp1s, wants = [], [], [], []
    for b, nq in enumerate(sizes):
        nt = 3000 + 100 * b
        idx = np.stack([rng.integers(0, nt, nq), rng.integers(0, nt, nq)], 1).astype(np.int32)
        d1 = rng.uniform(50, 400, nq).astype(np.float32)
        d0 = (d1 * rng.uniform(0.3, 1.0, nq)).astype(np.float32)
        if nq > 100:
            idx[7, 1] = -1                                               # no second neighbour: never a survivor
            d0[11] = np.float32(0.7 * np.float64(d1[11]))                # as close to the boundary as float32 gets
            d1[13], d0[13] = np.float32(10.0), np.float32(7.0)           # 7.0 < 0.7 * 10.0 in double? (0.7 is below 7/10: False)
        if b == 6:
            d0[:] = 0.0                                                  # every query passes
        if b == 7:
            d0[:] = d1                                                   # none does
        dist = np.stack([d0, d1], 1)
        blk = np.zeros((2, cap, 2), np.int32)
        blk[0, :nq], blk[1, :nq] = idx, dist.view(np.int32)
        blk[:, nq:] = rng.integers(-5, 5, (2, cap - nq, 2))             # rows past nq must be ignored
        kp0 = rng.uniform(0, 900, (max(nq, 1), 2)).astype(np.float32)
        kp1 = rng.uniform(0, 900, (nt, 2)).astype(np.float32)
        if b == 5:                                                      # consistent rays for the degenerate pair
            n = max(nq, 1)
            X = np.stack([rng.uniform(-6, 3, n), rng.uniform(-2, 5, n), rng.uniform(4, 13, n), np.ones(n)], 0)
            a, c = cams[b][0] @ X, cams[b][1] @ X
            kp0 = (a[:2] / a[2]).T.astype(np.float32)
            kp1 = np.tile((c[:2] / c[2]).T.astype(np.float32), (2, 1))[:nt] if nt <= 2 * n else rng.uniform(0, 900, (nt, 2)).astype(np.float32)
            idx[:, 0] = np.arange(nq) % len(kp1)
            blk[0, :nq] = idx
        wq, wt, _ = oracle.ratio_filter(idx, dist, 0.70)
        want = oracle.triangulate(cams[b][0], cams[b][1], kp0[wq].T.copy(), kp1[wt].T.copy(), normalise_w=True) if len(wq) else np.zeros((4, 0), np.float32)
        blocks.append(cu(blk)); kp0s.append(cu(kp0)); kp1s.append(cu(kp1)); wants.append((wq, wt, want))
    out = torch.full((8, 4 * cap + 4), 7.0, dtype=torch.float32, device="cuda")      # stale slot content must disappear
    pts = [out[b, :4 * cap].view(4, cap) for b in range(8)]
    cnt = [out[b, 4 * cap:4 * cap + 1].view(torch.int32) for b in range(8)]
    hip.triangulate_matches_batch(blocks, sizes, kp0s, kp1s, [c[0] for c in cams], [c[1] for c in cams], pts, cnt, ratio=0.70)
    torch.cuda.synchronize()
    assert len(wants[6][0]) == sizes[6] - 1 and len(wants[7][0]) == 0 and len(wants[4][0]) == 0      # (query 7 has no second neighbour)
    for b in range(8):
        wq, wt, want = wants[b]
        m = len(wq)
        assert int(cnt[b].item()) == m, (b, int(cnt[b].item()), m)
        got = pts[b].cpu().numpy()
        assert float(np.abs(got[:, m:]).sum()) == 0.0
        if m:
            faithful = hip.triangulate(cams[b][0], cams[b][1], kp0s[b][cu(wq).long()].t(), kp1s[b][cu(wt).long()].t(), normalise_w=True).cpu().numpy()
            assert np.array_equal(got[:, :m].view(np.uint32), faithful.view(np.uint32)), b      # = the faithful kernel, bit for bit
            if b != 5:                                                  # ... and the oracle (same operation order: almost always bit-identical)
                assert np.allclose(got[:, :m], want, rtol=1e-6, atol=1e-7) and (got[:, :m] == want).mean() > 0.99
    # a single pair through the same entry point, no count outputs
    one = torch.zeros((4, cap), dtype=torch.float32, device="cuda")
    hip.triangulate_matches_batch(blocks[1:2], sizes[1:2], kp0s[1:2], kp1s[1:2], [cams[1][0]], [cams[1][1]], [one], None)
    assert torch.equal(one, pts[1])
    with pytest.raises(hip.SfmHipError):
        hip.triangulate_matches_batch(blocks[:1], [cap + 1], kp0s[:1], kp1s[:1], [cams[0][0]], [cams[0][1]], [one], None)
