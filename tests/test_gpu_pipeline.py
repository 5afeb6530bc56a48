"""End-to-end entry points on the MI355X: RANSAC with device scoring vs the same control flow over the
CPU oracle (integer masks bit-exact), the cv2-named facade, and the sfm.py driver on a synthetic
Gustav-geometry sequence checked against the reference's own pose.csv."""
import numpy as np
import pytest

from datagen import decompose_P, gustav_pair, gustav_scene, planted_pair
from oracle_backend import oracle_pipeline_backend

pytestmark = pytest.mark.gpu


def _corrupt(x2, n_bad, lo, hi, seed):
    rng = np.random.default_rng(seed)
    bad = rng.permutation(len(x2))[:n_bad]
    x2 = x2.copy()
    x2[bad] += rng.uniform(lo, hi, (n_bad, 2)).astype(np.float32)
    return x2, bad


@pytest.mark.parametrize("pair,n,sigma,n_bad,seed", [(0, 900, 0.3, 200, 11), (0, 400, 0.05, 0, 2), (17, 1500, 0.5, 600, 3),
                                                      (40, 250, 0.2, 25, 4), (5, 64, 1.0, 10, 5), (30, 3000, 0.3, 300, 6)])
def test_ransac_entry_points_vs_sequential_oracle(hip, oracle, pair, n, sigma, n_bad, seed):
    """sfm_find_essential_mat / sfm_recover_pose / sfm_solve_pnp_ransac (chunked hypotheses, device scoring, device LM
    sweeps) against the oracle's one-model-at-a-time restatements: E and every integer output identical, the refined
    pose within 1e-9.  The high-inlier cases collapse `niters` inside the first chunk (the iteration-boundary rule)."""
    from sfm_mvs_amd import ransac
    K, P1, P2, X, x1, x2 = gustav_pair(pair, n, sigma, seed=seed)
    x2, bad = _corrupt(x2, n_bad, 10, 150, seed)
    Eo, mo, so = oracle.find_essential_mat(x1, x2, K, 0.999, 0.4, want_stats=True)
    Eh, mh, ih = ransac.find_essential_mat(x1, x2, K, 0.999, 0.4, want_info=True)
    assert np.array_equal(Eo, Eh) and np.array_equal(mo, mh) and mh.dtype == np.uint8 and mh.shape == (n, 1)
    assert (ih[1], ih[2], ih[3]) == (so[2], so[0], so[1])                       # inliers, iterations run, models scored
    sel = mo.ravel() == 1
    go, Ro, to, m2o = oracle.recover_pose(Eo, x1[sel], x2[sel], K)
    gh, Rh, th, m2h = ransac.recover_pose(Eh, x1[sel], x2[sel], K)
    assert go == gh and np.array_equal(m2o, m2h) and np.array_equal(Ro, Rh) and np.array_equal(to, th)
    assert set(np.unique(m2h)) <= {0, 255} and th.shape == (3, 1)

    Xf = X.astype(np.float32)
    oko, ro, tvo, io, model, st = oracle.solve_pnp_ransac(Xf, x2, K, want_model=True)
    okh, rh, tvh, inh, info = ransac.solve_pnp_ransac(Xf, x2, K, want_info=True)
    assert oko and okh and np.array_equal(io, inh) and inh.dtype == np.int32 and not set(io[:, 0]) & set(bad)
    assert info[1] == len(io) and info[2] == st
    assert np.abs(ro - rh).max() <= 1e-9 and np.abs(tvo - tvh).max() <= 1e-9 * max(1.0, np.abs(tvo).max())
    R, t = decompose_P(K, P2)
    assert np.abs(oracle.rodrigues_vec2mat(rh.ravel()) - R).max() < 5e-3 and np.abs(tvh.ravel() - t).max() < 5e-2


@pytest.mark.parametrize("pair,n,sigma,n_bad,seed", [(0, 900, 0.3, 200, 11), (5, 64, 1.0, 10, 5), (30, 3000, 0.3, 300, 6), (12, 9000, 0.4, 500, 7),
                                                      (21, 1024, 0.2, 0, 8), (21, 1025, 0.2, 0, 9), (8, 256, 0.2, 0, 10), (8, 257, 0.2, 0, 11), (3, 8192, 0.3, 0, 12),
                                                      (3, 8300, 0.3, 0, 13)])
def test_sweep_server_equals_launch_per_sweep(hip, pair, n, sigma, n_bad, seed):
    """solvePnPRansac through the resident PnP server (ONE launch per call; copy-in, hypothesis scoring, the inlier list and the
    Levenberg-Marquardt sweeps are requests through the pinned mailbox: csrc/ransac.hip pnp_server_kernel) and through a copy / launch +
    stream synchronisation per step: the same scores (the inlier test is a threshold on float32 errors), the same sums in
    the same tree, so rvec, tvec, the iteration count and the inlier list are bit-identical — within the server's range (<= 8 192 inliers: one workgroup,
    several, more than one virtual workgroup of the tree) and beyond it (both runs take the launch path).  Calls in a row reuse the
    mailbox: the sequence numbers carry over."""
    from sfm_mvs_amd import ransac, _lib
    K, P1, P2, X, x1, x2 = gustav_pair(pair, n, sigma, seed=seed)
    x2, _ = _corrupt(x2, n_bad, 10, 150, seed)
    Xf = X.astype(np.float32)
    L = _lib.lib()
    polls0, syncs0 = L.sfm_host_poll_count(), L.sfm_host_sync_count()
    got = [ransac.solve_pnp_ransac(Xf, x2, K, want_info=True) for _ in range(3)]
    polls1, syncs1 = L.sfm_host_poll_count(), L.sfm_host_sync_count()
    assert L.sfm_debug_pnp_sweep_server(0) == 1
    try:
        want = ransac.solve_pnp_ransac(Xf, x2, K, want_info=True)
        syncs2 = L.sfm_host_sync_count()
    finally:
        assert L.sfm_debug_pnp_sweep_server(1) == 0
    assert want[0] and want[4][3] >= 1                                                    # LM iterations ran
    for g in got:
        assert g[0] and np.array_equal(g[1], want[1]) and np.array_equal(g[2], want[2]) and np.array_equal(g[3], want[3])
        assert list(g[4]) == list(want[4])
    inl = int(want[4][1])
    if n <= 8192:                                     # served: polls instead of stream synchronisations — none at all
        assert polls1 > polls0 and syncs1 == syncs0 and syncs2 - syncs1 >= 3
    else:
        assert polls1 == polls0


def test_pnp_server_on_small_and_degenerate_sets_never_waits_for_a_timeout(hip):
    """Found by scripts/fuzz_geometry.py: the inlier list used to travel as its own, unacknowledged request; when the DLT
    initialisation that follows returns at once (fewer than six inliers, coplanar sets) the first sweep request overwrote it in the
    one-slot mailbox before a workgroup had polled it, the server waited a second for a request that never came, left, was restarted
    — three times, then the call failed ("the PnP server keeps leaving").  The list now travels WITH the first sweep request.  Many
    small / outlier-heavy / degenerate calls: each equals the launch-per-step path bit for bit and none takes as long as a time-out."""
    import time
    from sfm_mvs_amd import ransac, _lib
    L = _lib.lib()
    rng = np.random.default_rng(12)
    worst = 0.0
    for case in range(120):
        n = int(rng.integers(6, 40))
        K, P1, P2, X, x1, x2 = gustav_pair(int(rng.integers(0, 55)), n, float(rng.choice([0.0, 0.3, 3.0])), seed=1000 + case)
        x2, _ = _corrupt(x2, int(rng.integers(0, n)), 10, 150, case)
        Xf = X.astype(np.float32)
        kind = case % 4
        if kind == 1:
            Xf[:, 2] = 0.2 * Xf[:, 0] - 0.1 * Xf[:, 1] + 6.0          # coplanar object points
        elif kind == 2:
            pick = rng.integers(0, 3, n)
            Xf, x2 = Xf[pick], x2[pick]                                  # three distinct correspondences
        its, conf, rep = int(rng.choice([10, 100, 500])), float(rng.choice([0.9, 0.99, 0.9999])), float(rng.choice([2.0, 8.0, 20.0]))
        t0 = time.perf_counter()
        got = ransac.solve_pnp_ransac(Xf, x2, K, iterations_count=its, reprojection_error=rep, confidence=conf, want_info=True)
        worst = max(worst, time.perf_counter() - t0)
        assert L.sfm_debug_pnp_sweep_server(0) == 1
        try:
            want = ransac.solve_pnp_ransac(Xf, x2, K, iterations_count=its, reprojection_error=rep, confidence=conf, want_info=True)
        finally:
            L.sfm_debug_pnp_sweep_server(1)
        assert bool(got[0]) == bool(want[0]) and list(got[4]) == list(want[4]), (case, n, got[4], want[4])
        if got[0]:
            assert np.array_equal(got[3], want[3]), case
            for a, b in ((got[1], want[1]), (got[2], want[2])):
                assert np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(a[~np.isnan(a)], b[~np.isnan(b)]), case
    assert worst < 0.25, f"a served call took {worst:.2f} s: a request was lost and the server timed out"


def test_ransac_entry_point_edge_cases(hip, oracle):
    from sfm_mvs_amd import ransac
    from sfm_mvs_amd._lib import SfmHipError
    import torch
    K, P1, P2, X, x1, x2 = gustav_pair(3, 40, 0.2, seed=8)
    assert ransac.find_essential_mat(x1[:4], x2[:4], K, 0.999, 0.4) == (None, None)           # count < modelPoints
    E5o, m5o = oracle.find_essential_mat(x1[:5], x2[:5], K, 0.999, 0.4)                       # count == modelPoints
    E5h, m5h = ransac.find_essential_mat(x1[:5], x2[:5], K, 0.999, 0.4)
    assert np.array_equal(E5o, E5h) and np.array_equal(m5o, m5h) and E5h.shape[0] % 3 == 0
    Xf = X.astype(np.float32)
    ok5o, r5o, t5o, i5o = oracle.solve_pnp_ransac(Xf[:5], x2[:5], K)
    ok5h, r5h, t5h, i5h = ransac.solve_pnp_ransac(Xf[:5], x2[:5], K)
    assert ok5o and ok5h and np.array_equal(r5o, r5h) and np.array_equal(t5o, t5h) and np.array_equal(i5o, i5h)
    ok4o, r4o, t4o, i4o = oracle.solve_pnp_ransac(Xf[:4], x2[:4], K)                         # npoints == 4: solvePnP(P3P)
    ok4h, r4h, t4h, i4h = ransac.solve_pnp_ransac(Xf[:4], x2[:4], K)
    assert ok4o and ok4h and np.array_equal(r4o, r4h) and np.array_equal(t4o, t4h) and np.array_equal(i4o, i4h)
    with pytest.raises(SfmHipError):
        ransac.solve_pnp_ransac(Xf[:3], x2[:3], K)                                            # OpenCV asserts npoints >= 4
    # five correspondences of which three are distinct (found by scripts/fuzz_geometry.py): model_points == npoints is a plain
    # solvePnP(EPNP) — it reports success with whatever the solver produced, it does not filter like the RANSAC loop
    pick = np.array([0, 1, 2, 0, 1])
    okdo, rdo, tdo, ido = oracle.solve_pnp_ransac(Xf[pick], x2[pick], K)
    okdh, rdh, tdh, idh = ransac.solve_pnp_ransac(Xf[pick], x2[pick], K)
    assert bool(okdo) == bool(okdh) and np.array_equal(ido, idh)
    assert np.array_equal(np.isnan(rdo), np.isnan(rdh)) and np.array_equal(np.isnan(tdo), np.isnan(tdh))
    # pure outliers: no model survives (good must exceed modelPoints - 1)
    rng = np.random.default_rng(0)
    junk = rng.uniform(0, 900, (60, 2)).astype(np.float32)
    oko, *_ = oracle.solve_pnp_ransac(Xf[:30] * 0 + rng.normal(0, 1, (30, 3)).astype(np.float32), junk[:30], K)
    okh, *_ = ransac.solve_pnp_ransac(Xf[:30] * 0 + np.random.default_rng(0).normal(0, 1, (30, 3)).astype(np.float32), junk[:30], K)
    assert isinstance(okh, bool)
    # a planar object: the refinement starts from the RANSAC model on both sides (status 1)
    flat = Xf.copy()
    flat[:, 2] = 0.2 * flat[:, 0] - 0.1 * flat[:, 1] + 6.0
    Rm, tm = decompose_P(K, P2)
    xp = (K @ (flat.astype(np.float64) @ Rm.T + tm).T).T
    xp = (xp[:, :2] / xp[:, 2:]).astype(np.float32)
    oko, ro, to, io, model, st = oracle.solve_pnp_ransac(flat, xp, K, want_model=True)
    okh, rh, th, ih, info = ransac.solve_pnp_ransac(flat, xp, K, want_info=True)
    assert oko and okh and st == 1 and info[2] == 1 and np.array_equal(io, ih) and np.abs(ro - rh).max() <= 1e-8
    # device tensors in, device mask / inlier list out
    x1d, x2d = torch.from_numpy(x1).cuda(), torch.from_numpy(x2).cuda()
    Ed, md = ransac.find_essential_mat(x1d, x2d, K, 0.999, 0.4, return_device_mask=True)
    Eo, mo = oracle.find_essential_mat(x1, x2, K, 0.999, 0.4)
    assert md.is_cuda and np.array_equal(md.cpu().numpy(), mo) and np.array_equal(Ed, Eo)


def test_cv2compat_shapes_and_values(hip, oracle):
    from sfm_mvs_amd import cv2compat as cv2
    rng = np.random.default_rng(1)
    q, t, _ = planted_pair(rng, 300, 400, 0.3)
    matches = cv2.BFMatcher().knnMatch(q, t, k=2)
    wi, wd = oracle.knn2(q, t)
    assert len(matches) == 300 and all(len(m) == 2 for m in matches)
    assert [m[0].trainIdx for m in matches] == wi[:, 0].tolist() and [m[1].distance for m in matches] == wd[:, 1].tolist()
    good = [m for m, n in matches if m.distance < 0.70 * n.distance]            # the loop of sfm.py:262-265
    assert [g.queryIdx for g in good] == oracle.ratio_filter(wi, wd, 0.70)[0].tolist()
    assert len(cv2.BFMatcher().knnMatch(q[:3], t[:1], k=2)[0]) == 1              # fewer than k trains

    K, P1, P2, X, x1, x2 = gustav_pair(2, 333, 0.3, seed=4)
    cloud = cv2.triangulatePoints(P1, P2, x1.T, x2.T)                            # transposed views, sfm.py:47-53
    assert cloud.shape == (4, 333) and cloud.dtype == np.float32
    assert np.allclose(cloud, oracle.triangulate(P1, P2, x1.T, x2.T), rtol=1e-6, atol=1e-9)
    R, tv = decompose_P(K, P2)
    r, _ = cv2.Rodrigues(R)
    assert r.shape == (3, 1) and np.allclose(cv2.Rodrigues(r)[0], R, atol=1e-12)
    Xh = cv2.convertPointsFromHomogeneous((cloud / cloud[3]).T)
    assert Xh.shape == (333, 1, 3)
    p, _ = cv2.projectPoints(Xh, r, tv, K, distCoeffs=None)
    assert p.shape == (333, 1, 2) and p.dtype == np.float32
    _, wp = oracle.project_points(r.ravel(), tv, K, Xh[:, 0, :])
    assert np.allclose(p[:, 0, :], wp, rtol=1e-6, atol=1e-4)
    n = cv2.norm(p[:, 0, :], x2, cv2.NORM_L2)
    assert n == pytest.approx(np.sqrt(((p[:, 0, :] - x2).astype(np.float64) ** 2).sum()), rel=1e-12)
    ret, rvec, tvec, inl = cv2.solvePnPRansac(Xh[:, 0, :], x2, K, np.zeros((5, 1), np.float32), cv2.SOLVEPNP_ITERATIVE)
    assert ret and rvec.shape == (3, 1) and tvec.shape == (3, 1) and inl.shape[1] == 1 and inl.dtype == np.int32


def test_reference_helpers_mirror(hip, oracle):
    """Triangulation / ReprojectionError / PnP with the reference's argument shapes (sfm.py:320-325)."""
    from sfm_mvs_amd import pipeline as pl
    K, P1, P2, X, x1, x2 = gustav_pair(1, 500, 0.3, seed=9)
    a, b, cloud = pl.Triangulation(P1, P2, x1, x2, K, repeat=False)
    assert a.shape == (2, 500) and cloud.shape == (4, 500) and np.all(cloud[3] == 1)
    R, t = decompose_P(K, P2)
    Rt = np.hstack([R, t[:, None]])
    err, X3, p = pl.ReprojectionError(cloud, b, Rt, K, homogenity=1)
    assert X3.shape == (500, 1, 3) and p.shape == (500, 2)
    want, _ = oracle.reprojection_error(Rt, K, np.ascontiguousarray(oracle.triangulate(P1, P2, x1.T, x2.T, normalise_w=True)[:3].T), x2)
    assert err == pytest.approx(want, rel=1e-4)                                  # north_star bar: 1e-4 relative
    Rot, trans, pf, Xf, p0f = pl.PnP(X3, b, K, np.zeros((5, 1), np.float32), a, initial=1)
    assert Rot.shape == (3, 3) and trans.shape == (3, 1) and pf.shape[1] == 2 and Xf.shape[1] == 3 and len(pf) == len(p0f)
    assert np.abs(Rot - R).max() < 1e-3


def test_driver_reproduces_pose_csv_on_gustav_geometry(hip):
    """BASELINE config 3 in miniature: the sfm.py driver over 8 synthetic frames rendered from the
    reference's own cameras and cloud must recover those cameras (pose.csv) and near-zero reprojection error."""
    from sfm_mvs_amd import pipeline as pl
    K, P, feats, ids = gustav_scene(8, seed=3)
    out = pl.run_sfm(feats, K)
    got = out["posearr"][9:].reshape(-1, 3, 4)
    assert np.array_equal(out["posearr"][:9], K.ravel()) and got.shape == (8, 3, 4)
    assert np.array_equal(got[0], P[0])
    for k in range(1, 8):
        Rg, tg = decompose_P(K, got[k])
        Rw, tw = decompose_P(K, P[k])
        assert np.abs(Rg - Rw).max() < 2e-3, k
        assert np.linalg.norm(tg - tw) < 2e-2 * max(1.0, np.linalg.norm(tw)), k
    assert out["first_error"] < 0.05 and max(out["errors"]) < 0.05
    assert len(out["Xtot"]) > 100 and np.all(out["Xtot"][0] == 0)              # quirk 8: leading zero row


def test_full_57_camera_sequence_matches_pose_csv(hip):
    """BASELINE config 3 at full length on Gustav geometry: all 57 cameras of the reference's pose.csv are
    re-registered by the incremental driver (match -> E/RANSAC -> triangulate -> PnP per frame)."""
    from sfm_mvs_amd import pipeline as pl
    K, P, feats, ids = gustav_scene(57, seed=3)
    out = pl.run_sfm(feats, K)
    got = out["posearr"][9:].reshape(-1, 3, 4)
    assert got.shape == (57, 3, 4) and out["posearr"].shape == (9 + 57 * 12,)      # = pose.csv's 693 values
    for k in range(57):
        Rg, tg = decompose_P(K, got[k])
        Rw, tw = decompose_P(K, P[k])
        assert np.abs(Rg - Rw).max() < 1e-3 and np.linalg.norm(tg - tw) < 5e-3 * max(1.0, np.linalg.norm(tw)), k
    assert max(out["errors"]) < 0.01                      # the reference's metric ||dp||_F / N per frame


def test_every_frame_of_the_57_camera_chain_teacher_forced(hip, oracle):
    """Row D at north_star's bar, frame by frame over the WHOLE Gustav-geometry sequence: for every frame k the HIP frame step
    (match -> associate -> solvePnPRansac -> triangulate -> ReprojectionError, sfm.py:341-409) is fed the state the CPU-oracle
    chain reached before that frame and must reproduce the oracle's outputs: identical integer decisions (match lists,
    association, RANSAC inlier sets -> array shapes), projection matrix, new cloud and reprojection error within 1e-4 relative
    (measured: 1e-9 .. 1e-7).  Same inputs -> same outputs for all 55 registrations; the drift of the free-running chain is
    what test_free_running_chain_drift quantifies separately."""
    from sfm_mvs_amd import pipeline as pl
    K, P, feats, ids = gustav_scene(57, seed=3)
    ora = pl.make_engine(feats, K, be=oracle_pipeline_backend(oracle))
    dev = pl.make_engine(feats, K)
    s_o, first_o = pl.bootstrap_pair(ora)
    s_h, first_h = pl.bootstrap_pair(dev)
    assert dev.errors([first_h])[0] == pytest.approx(ora.errors([first_o])[0], rel=1e-6)
    assert np.allclose(s_h.P2, s_o.P2, rtol=1e-9, atol=1e-9)
    for a, b in ((s_h.pts0, s_o.pts0), (s_h.pts1, s_o.pts1)):
        assert np.array_equal(dev.host(a), b)                              # identical E / pose / PnP inlier sets
    assert np.allclose(dev.host(s_h.cloud0), s_o.cloud0, rtol=1e-5, atol=1e-6)
    worst = dict(P=0.0, error=0.0, cloud=0.0)
    for i in range(len(feats) - 2):
        s_next, want = pl.register_next(ora, s_o, i)
        _, got = pl.register_next(dev, s_o.on(dev), i)                     # the HIP step on the ORACLE's state
        assert got["pnp_inliers"] == want["pnp_inliers"], i
        assert np.array_equal(dev.host(got["lookup"]), want["lookup"]), i  # same association / complement
        gc, wc = dev.host(got["cloud"]), want["cloud"]
        assert gc.shape == wc.shape, i
        ge, we = dev.errors([got["error"]])[0], ora.errors([want["error"]])[0]
        worst["P"] = max(worst["P"], float(np.abs(got["P"] - want["P"]).max() / np.abs(want["P"]).max()))
        worst["error"] = max(worst["error"], abs(ge - we) / we)
        worst["cloud"] = max(worst["cloud"], float(np.abs(gc - wc).max() / np.abs(wc).max()))
        assert np.allclose(got["P"], want["P"], rtol=1e-4, atol=1e-4 * np.abs(want["P"]).max()), i
        assert ge == pytest.approx(we, rel=1e-4), i
        assert np.allclose(gc, wc, rtol=1e-4, atol=1e-4 * np.abs(wc).max()), i
        s_o = s_next
    print("teacher-forced worst relative differences over 55 frames:", worst)
    assert worst["P"] < 1e-5 and worst["error"] < 1e-4 and worst["cloud"] < 1e-4


def test_free_running_chain_drift(hip, oracle):
    """Row D at north_star's closing bar: the two drivers running FREE over the WHOLE 57-camera Gustav-geometry sequence
    (sfm.py:341-409) — once on the HIP back-end, once with every numeric operator, RANSAC entry points and their minimal
    solvers included, replaced by the CPU oracle's sequential restatements.  Every camera is registered against float32
    points triangulated from the earlier ones, so any last-bit difference is amplified ~2.5x per frame (rounds 1-3 held
    1e-4 over 6-8 frames only).  Since round 4 the one floating-point reduction whose order differed between the two sides —
    the 28 sums of solvePnP's Levenberg-Marquardt sweep — follows ONE fixed tree in csrc/ransac.hip and
    oracle/solvers_oracle.c, and the chain is reproduced to the bit: identical shapes (integer decisions), poses, cloud and
    per-frame errors within 1e-4 relative over all 55 registrations (measured differences are printed; expected 0)."""
    from sfm_mvs_amd import pipeline as pl
    for n, seed, noise in ((57, 3, 0.0), (57, 5, 0.0), (11, 7, 0.2)):      # (with pixel noise the synthetic sequence runs out of new points at frame 12: the reference would divide by len(p) = 0 there too)
        K, P, feats, ids = gustav_scene(n, seed=seed, pix_noise=noise)
        got = pl.run_sfm(feats, K)
        want = pl.run_sfm(feats, K, be=oracle_pipeline_backend(oracle))
        assert got["posearr"].shape == want["posearr"].shape == (9 + 12 * n,) and got["Xtot"].shape == want["Xtot"].shape
        assert len(got["errors"]) == len(want["errors"]) == n - 2
        dP = np.abs(got["posearr"] - want["posearr"]).reshape(-1)[9:].reshape(n, 12).max(1) / np.abs(want["posearr"][9:]).reshape(n, 12).max(1)
        dE = np.array([abs(a - b) / b for a, b in zip(got["errors"], want["errors"])])
        dX = np.abs(got["Xtot"] - want["Xtot"]).max() / np.abs(want["Xtot"]).max()
        print(f"free-running chain, {n} cameras (seed {seed}, pixel noise {noise}): max rel diff P {dP.max():.2e} (frame {int(dP.argmax())}), "
              f"error {dE.max():.2e} (frame {int(dE.argmax())}), cloud {dX:.2e}; bit-identical poses: {bool(np.array_equal(got['posearr'], want['posearr']))}, "
              f"cloud: {bool(np.array_equal(got['Xtot'], want['Xtot']))}")
        assert got["first_error"] == pytest.approx(want["first_error"], rel=1e-6)
        assert dP.max() <= 1e-4 and dE.max() <= 1e-4 and dX <= 1e-4, (dP.max(), dE.max(), dX)


def test_driver_with_bundle_adjustment_enabled(hip, oracle):
    """The `if bundle_adjustment:` branch of the reference's loop (sfm.py:378-388, off as shipped): every registration is
    followed by SciPy's least_squares over the new cloud, its observations and the camera, and the refined values replace
    the frame's.  Teacher-forced per frame against the CPU twin (same driver, oracle operators): same integer decisions, P,
    cloud, lookup points and minimised error within 1e-4; the error really is minimised; and the free-running driver
    keeps the PRE-adjustment camera in the pose array (sfm.py:375 appends Pnew before the `if bundle_adjustment:` branch) while
    the refined one drives the next frame."""
    from sfm_mvs_amd import pipeline as pl
    K, P, feats, ids = gustav_scene(5, seed=3, pix_noise=0.5)      # (new clouds of 1306, 2 and 20 points: the first is left as it is —
    # its gradient is below gtol — the other two are moved)
    ora = pl.make_engine(feats, K, be=oracle_pipeline_backend(oracle))
    dev = pl.make_engine(feats, K)
    s_o, _ = pl.bootstrap_pair(ora)
    for i in range(len(feats) - 2):
        s_next, want = pl.register_next(ora, s_o, i, bundle_adjustment=True, gtol_thresh=0.5)
        _, got = pl.register_next(dev, s_o.on(dev), i, bundle_adjustment=True, gtol_thresh=0.5)
        assert got["pnp_inliers"] == want["pnp_inliers"] and got["cloud"].shape == want["cloud"].shape and len(want["cloud"]) >= 2
        print(i, len(want["cloud"]), float(np.abs(got["P"] - want["P"]).max() / np.abs(want["P"]).max()), float(np.abs(got["cloud"] - want["cloud"]).max()))
        assert np.allclose(got["P"], want["P"], rtol=1e-4, atol=1e-4 * np.abs(want["P"]).max()), i
        assert np.allclose(got["cloud"], want["cloud"], rtol=1e-4, atol=1e-4 * np.abs(want["cloud"]).max()), i
        assert np.allclose(got["lookup"], want["lookup"], rtol=1e-4, atol=1e-3), i
        ge, we = dev.errors([got["error"]])[0], ora.errors([want["error"]])[0]
        assert ge == pytest.approx(we, rel=1e-4, abs=1e-9), i
        assert ge <= dev.errors([got["ba_error_before"]])[0] * (1 + 1e-9), i      # "Minimized error" (sfm.py:385)
        s_o = s_next
    out = pl.run_sfm(feats, K, bundle_adjustment=True)
    plain = pl.run_sfm(feats, K)
    assert out["posearr"].shape == plain["posearr"].shape == (9 + 12 * 5,) and out["Xtot"].shape == plain["Xtot"].shape
    # posearr holds what PnP found.  Frames 2 and 3 are registered from clouds no adjustment has touched yet (frame 2's own
    # adjustment leaves it as it is: gradient below gtol), so their entries equal the plain run's; frame 3's adjustment moves its
    # camera, and frame 4 — triangulated from that refined P — is registered differently
    assert np.array_equal(out["posearr"][:9 + 48], plain["posearr"][:9 + 48])
    assert not np.array_equal(out["posearr"][9 + 48:], plain["posearr"][9 + 48:])
    assert len(out["errors"]) == 3 and out["errors"][1] < plain["errors"][1] and out["errors"][2] < plain["errors"][2]


def test_bundle_adjustment_mirror(hip, oracle):
    """sfm.py:104-157 (off by default in the reference): the residual vector is fp64 end to end — equal to a float64
    NumPy twin to 1e-12 and to the CPU oracle twin — so SciPy's finite-difference Jacobian sees X, Rt and K (a float32
    round trip would zero those columns), and TRF really lowers the cost.  Small N: every Jacobian costs 5N+22 residual
    evaluations, as in the reference."""
    from sfm_mvs_amd import pipeline as pl
    K, P1, P2, X, x1, x2 = gustav_pair(2, 12, 0.5, seed=4)
    R, t = decompose_P(K, P2)
    Rt = np.hstack([R, t[:, None]])
    Xn = X + np.random.default_rng(0).normal(0, 0.01, X.shape)
    x0 = np.hstack([Rt.ravel(), K.ravel(), x2.T.astype(np.float64).ravel(), Xn.ravel()])
    r_hip = pl.OptimReprojectionError(x0)
    r_cpu = pl.OptimReprojectionError(x0, be=oracle_pipeline_backend(oracle))
    Xc = Xn @ R.T + t
    twin = np.stack([K[0, 0] * Xc[:, 0] / Xc[:, 2] + K[0, 2], K[1, 1] * Xc[:, 1] / Xc[:, 2] + K[1, 2]], 1)
    want = ((x2.astype(np.float64) - twin) ** 2).ravel() / 12
    assert r_hip.shape == (24,) and np.allclose(r_hip, want, rtol=1e-10, atol=1e-12) and np.allclose(r_cpu, want, rtol=1e-10, atol=1e-12)
    dx = np.zeros_like(x0)
    dx[21 + 24 + 5] = 1e-8                                                       # a finite-difference step on one X coordinate
    assert np.abs(pl.OptimReprojectionError(x0 + dx) - r_hip).max() > 0          # ... is visible in the residual
    Xo, po, Rto = pl.BundleAdjustment(Xn, x2.T, Rt, K, 0.5)
    assert Xo.shape == (12, 3) and po.shape == (12, 2) and Rto.shape == (3, 4)
    x1v = np.hstack([Rto.ravel(), K.ravel(), po.T.ravel(), Xo.ravel()])
    assert pl.OptimReprojectionError(x1v).sum() < 0.9 * r_hip.sum()              # a real decrease
    assert np.abs(Xo - Xn).max() > 1e-6 or np.abs(Rto - Rt).max() > 1e-9        # the 3-D points / pose moved, not only the observations


def test_common_points_kernel_vs_reference_golden_vectors(hip, oracle):
    """sfm_common_points against the vectors produced by executing the reference's own common_points (x-OR-y quirk,
    duplicates, empty intersections) and against the oracle on a frame-sized random case."""
    import os
    from datagen import GOLDEN
    z = np.load(os.path.join(GOLDEN, "common_points.npz"))
    for name in "abcd":
        i1, i2, keep = hip.common_points(cu32(z[f"{name}_pts1"]), cu32(z[f"{name}_pts2"]))
        assert np.array_equal(i1.cpu().numpy(), z[f"{name}_indx1"]) and np.array_equal(i2.cpu().numpy(), z[f"{name}_indx2"]), name
        k = keep.cpu().numpy()
        assert np.array_equal(z[f"{name}_pts2"][k], z[f"{name}_temp1"].reshape(-1, 2))
        assert np.array_equal(z[f"{name}_pts3"][k], z[f"{name}_temp2"].reshape(-1, 2))
    rng = np.random.default_rng(3)
    p2 = np.round(rng.uniform(0, 900, (3100, 2)), 1).astype(np.float32)       # coarse grid: many x-only / y-only hits
    p1 = np.round(rng.uniform(0, 900, (2900, 2)), 1).astype(np.float32)
    p1[::3] = p2[rng.permutation(3100)[:967]]
    wi1, wi2, wt1, _ = oracle.common_points(p1, p2, p2)
    i1, i2, keep = hip.common_points(cu32(p1), cu32(p2))
    assert np.array_equal(i1.cpu().numpy(), wi1) and np.array_equal(i2.cpu().numpy(), wi2)
    assert np.array_equal(p2[keep.cpu().numpy()], wt1)


def cu32(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()


def test_hip_helpers_against_the_references_own_helpers(hip, oracle):
    """Same replay as tests/test_oracle_golden.py, on the HIP back-end, against the golden outputs of the reference's
    helper source: index sets identical, floating-point outputs within north_star's 1e-4."""
    import os
    from datagen import GOLDEN
    from test_oracle_golden import _replay_helpers
    from sfm_mvs_amd import pipeline as pl
    g = np.load(os.path.join(GOLDEN, "helpers.npz"))
    for tag in ("a", "b"):
        got = _replay_helpers(pl, None, g, tag)
        for k, v in got.items():
            want = g[f"{tag}_{k}"]
            v = np.asarray(v)
            assert v.shape == want.shape, (tag, k, v.shape, want.shape)       # same inlier COUNT for the gathered arrays
            if k in ("pts1", "pts2", "p_in", "p0_in", "q_in", "q0_in"):         # pure gathers of inputs: the inlier sets
                assert np.array_equal(v, want), (tag, k)
            else:
                scale = max(np.abs(want).max(), 1e-30)
                assert np.abs(v - want).max() <= 1e-4 * scale, (tag, k, np.abs(v - want).max(), scale)


def test_sharded_path_on_one_gpu(hip, oracle):
    """sfm_mvs_amd.sharded with its default HIP engines on a single rank (the code bench.py --gpus N and the gloo tests
    drive): images of different sizes (padded blocks), batches of 2 with a partial last one, then the point exchange."""
    import torch
    from datagen import load_pose_csv
    from sfm_mvs_amd import sharded
    rng = np.random.default_rng(5)
    sizes = [700, 640, 700, 512, 700, 333]
    des = [planted_pair(rng, n, 10, 0.0)[0] for n in sizes]
    for k in range(5):
        m = min(sizes[k], sizes[k + 1]) // 2
        des[k + 1][:m] = des[k][rng.permutation(sizes[k])[:m]]
    kps = [rng.uniform(0, 900, (n, 2)).astype(np.float32) for n in sizes]
    pairs = sharded.sequential_pairs(6)
    dd = [torch.from_numpy(d).cuda() for d in des]
    store, nq = sharded.match_pairs_sharded(dd, pairs, batch=2)
    torch.cuda.synchronize()
    assert tuple(store.shape) == (5, 2, 700, 2) and nq == sizes[:5]
    for p, (i, j) in enumerate(pairs):
        wi, wd = oracle.knn2(des[i], des[j])
        assert np.array_equal(store[p, 0, :nq[p]].cpu().numpy(), wi)
        assert np.array_equal(store[p, 1, :nq[p]].cpu().numpy().view(np.float32), wd)
    K, P = load_pose_csv()
    pts, counts = sharded.triangulate_pairs_sharded(store, nq, pairs, [torch.from_numpy(k).cuda() for k in kps], list(P[:6]), batch=2)
    for p, (i, j) in enumerate(pairs):
        q, t = sharded.ratio_survivors(store[p], nq[p])
        wq, wt, _ = oracle.ratio_filter(*oracle.knn2(des[i], des[j]), 0.70)
        assert np.array_equal(q.cpu().numpy(), wq) and np.array_equal(t.cpu().numpy(), wt) and int(counts[p]) == len(wq) > 50
        want = oracle.triangulate(P[i], P[j], kps[i][wq].T.copy(), kps[j][wt].T.copy(), normalise_w=True)
        got = pts[p, :, :len(wq)].cpu().numpy()
        assert np.allclose(got, want, rtol=1e-6, atol=1e-7) and float(pts[p, :, len(wq):].abs().sum()) == 0.0
    # the train-split form on one rank with the HIP kernel as its engine: a no-op merge that must equal the plain call
    gi, gd = sharded.knn2_train_split(dd[0], dd[1], 0)
    wi, wd = oracle.knn2(des[0], des[1])
    assert np.array_equal(gi.cpu().numpy(), wi) and np.array_equal(gd.cpu().numpy(), wd)


def test_batched_match_engine_padded_blocks_and_exhaustive_verification(hip, oracle):
    """(1) sharded.HipMatchEngine issues pairs of one shape up to 8 per launch set: a sequence whose shapes repeat with fewer
    queries than the block has rows (padded blocks, copied behind the launch set) and with exactly as many (written in place),
    batches of 3.  (2) isfm.py:56-94 on a 5-image ring scene: all 10 pairs matched, then findEssentialMat + recoverPose per
    pair through sharded.verify_pairs_sharded — the printed inlier counts equal the oracle twin's."""
    import torch
    from datagen import ring_scene
    from sfm_mvs_amd import sharded
    rng = np.random.default_rng(9)
    sizes = [512, 512, 512, 700, 700, 700, 300]
    des = [planted_pair(rng, n, 10, 0.0)[0] for n in sizes]
    for k in range(len(sizes) - 1):
        m = min(sizes[k], sizes[k + 1]) // 2
        des[k + 1][:m] = des[k][rng.permutation(sizes[k])[:m]]
    pairs = sharded.sequential_pairs(len(sizes))
    store, nq = sharded.match_pairs_sharded([torch.from_numpy(d).cuda() for d in des], pairs, batch=3)
    torch.cuda.synchronize()
    for p, (i, j) in enumerate(pairs):
        wi, wd = oracle.knn2(des[i], des[j])
        assert np.array_equal(store[p, 0, :nq[p]].cpu().numpy(), wi), p
        assert np.array_equal(store[p, 1, :nq[p]].cpu().numpy().view(np.float32), wd), p
    K, P, image = ring_scene(5, 1500, 1000, seed=4)
    imgs = [image(k) for k in range(5)]
    pairs = sharded.all_pairs(5)
    store, nq = sharded.match_pairs_sharded([torch.from_numpy(im[1]).cuda() for im in imgs], pairs, batch=4)
    kps = [torch.from_numpy(im[0]).cuda() for im in imgs]
    got = sharded.verify_pairs_sharded(store, nq, pairs, kps, K)
    for p, (j, i) in enumerate(pairs):
        wi, wd = oracle.knn2(imgs[j][1], imgs[i][1])
        assert np.array_equal(store[p, 0].cpu().numpy(), wi) and np.array_equal(store[p, 1].cpu().numpy().view(np.float32), wd)
        q, t, _ = oracle.ratio_filter(wi, wd, 0.70)
        assert len(q) > 900
        a, b = imgs[j][0][q], imgs[i][0][t]
        E, m = oracle.find_essential_mat(a, b, K, 0.999, 0.4)
        keep = m.ravel() == 1
        want = int((oracle.recover_pose(E, a[keep], b[keep], K)[3].ravel() > 0).sum())
        assert int(got[p]) == want and want > 100, (p, int(got[p]), want)


def test_train_split_merge_kernel_equals_a_single_scan(hip, oracle):
    """sfm_knn_merge_top2 (SURVEY 8e's train-split fallback): partial 2-NN results of S shards of the train set, merged by
    (distance, global trainIdx), equal the single scan — including exact ties across shards (duplicated train rows: the
    lower index must win), shards that contribute one or no neighbour, and queries with fewer than two neighbours at all."""
    import torch
    rng = np.random.default_rng(17)
    q = rng.integers(0, 60, (700, 128)).astype(np.float32)
    t = rng.integers(0, 60, (901, 128)).astype(np.float32)
    t[400:420] = t[100:120]                              # exact duplicates in different shards
    t[650] = q[3]                                        # a zero distance ...
    t[20] = q[3]                                         # ... twice
    bounds = [0, 1, 300, 300, 600, 901]                  # shards of 1, 299, 0, 300 and 301 rows
    S = len(bounds) - 1
    cand = torch.empty((S, 2, 700, 2), dtype=torch.int32, device="cuda")
    dq = torch.from_numpy(q).cuda()
    for s in range(S):
        lo, hi = bounds[s], bounds[s + 1]
        if hi > lo:
            idx, d = hip.knn2(dq, torch.from_numpy(t[lo:hi]).cuda())
            cand[s, 0] = torch.where(idx >= 0, idx + lo, idx)
            cand[s, 1] = d.contiguous().view(torch.int32)
        else:
            cand[s, 0].fill_(-1)
            cand[s, 1].zero_()
    gi, gd = hip.knn_merge_top2(cand)
    wi, wd = oracle.knn2(q, t)
    assert np.array_equal(gi.cpu().numpy(), wi) and np.array_equal(gd.cpu().numpy().view(np.uint32), wd.view(np.uint32))
    # a single one-row shard: the second neighbour is missing
    one = torch.empty((1, 2, 700, 2), dtype=torch.int32, device="cuda")
    idx, d = hip.knn2(dq, torch.from_numpy(t[:1]).cuda())
    one[0, 0], one[0, 1] = idx, d.contiguous().view(torch.int32)
    gi, gd = hip.knn_merge_top2(one)
    assert (gi[:, 0] == 0).all() and (gi[:, 1] == -1).all() and float(gd[:, 1].abs().sum()) == 0.0


def test_device_resident_driver_equals_the_array_form(hip):
    """run_sfm's HBM-resident form (the default on the HIP back-end) against its array-in / array-out form: the same
    kernels in the same order, so every output is identical — poses, cloud, per-frame errors, colours."""
    from sfm_mvs_amd import pipeline as pl
    K, P, feats, ids = gustav_scene(9, seed=7, pix_noise=0.2)
    rng = np.random.default_rng(0)
    images = [rng.integers(0, 256, (648, 968, 3), dtype=np.uint8) for _ in range(9)]
    a = pl.run_sfm(feats, K, images=images, device_resident=True)
    b = pl.run_sfm(feats, K, images=images, device_resident=False)
    for k in ("posearr", "Xtot", "colorstot"):
        assert a[k].shape == b[k].shape and np.array_equal(a[k], b[k]), k
    assert a["errors"] == b["errors"] and a["first_error"] == b["first_error"]


def test_mask_indices_kernel(hip):
    """sfm_mask_indices — `pts[mask.ravel() == 1]` / `pts[mask.ravel() > 0]` (sfm.py:309,313) and the complement rows of
    common_points — against NumPy on masks with the byte values OpenCV produces (0, 1, 255) and others, at sizes around the
    1024-row block boundaries, empty and large."""
    import torch
    rng = np.random.default_rng(3)
    for n in (0, 1, 5, 1023, 1024, 1025, 4096, 100_003):
        m = rng.choice(np.array([0, 0, 1, 255, 7], np.uint8), n)
        d = torch.from_numpy(m).cuda()
        assert np.array_equal(hip.mask_indices(d).cpu().numpy(), np.flatnonzero(m == 1))
        assert np.array_equal(hip.mask_indices(d.reshape(-1, 1), nonzero=True).cpu().numpy(), np.flatnonzero(m > 0))
        assert np.array_equal(hip.mask_indices(torch.from_numpy(m > 0).cuda(), nonzero=True).cpu().numpy(), np.flatnonzero(m > 0))
    with pytest.raises(hip.SfmHipError):
        hip.mask_indices(torch.zeros(4, dtype=torch.float32, device="cuda"))


@pytest.mark.gpu
def test_driver_from_pixels_on_the_reference_camera_path(hip, oracle):
    """sfm.py:301-409 from PIXELS along the reference's own camera path (pose.csv): full-size frames rendered around textured 3-D
    structure -> img_downscale -> cvtColor + SIFT -> the driver.  The HIP features of every frame equal the oracle's bit for
    bit, the free-running chain equals the oracle twin's (<= 1e-4 asserted; identical in practice), a profiled run changes
    nothing and reports every stage and the host's waits, and the planted cameras come back (pose.csv's gauge: first camera at
    the origin, unit first baseline — what recoverPose fixes)."""
    from datagen import decompose_P, gustav_views
    from oracle_backend import oracle_pipeline_backend
    from sfm_mvs_amd import pipeline as pl
    n = 5
    images, K, P = gustav_views(n, seed=5)
    assert images[0].shape == (1296, 1936, 3)
    out = pl.run_sfm_images(images, K, downscale=2)
    prof = pl.DriverProfile()
    outp = pl.run_sfm_images(images, K, downscale=2, profile=prof)
    assert np.array_equal(outp["posearr"], out["posearr"]) and np.array_equal(outp["Xtot"], out["Xtot"])
    rep = prof.report(n - 2)
    assert {"findEssentialMat (sfm.py:307)", "solvePnPRansac + inlier gathers (sfm.py:67-76)", "cvtColor + SIFT detectAndCompute (sfm.py:243-252)"} <= set(rep["stage_ms"])
    assert rep["host_syncs"]["library_side"] > 0 and rep["host_syncs"]["per_registered_camera"] < 200
    assert rep["solve_pnp_ransac_host_us_per_call"]["calls"] == n - 1          # bootstrap's discarded call + one per registered camera
    feats_o = []
    for im in images:
        kp, des = oracle.sift(oracle.bgr2gray(oracle.pyrdown(im)))
        feats_o.append((np.ascontiguousarray(kp[:, :2]), des))
    for (kp, des), (okp, odes) in zip(out["features"], feats_o):
        assert len(okp) > 2000
        assert np.array_equal(kp.cpu().numpy().view(np.int32), okp.view(np.int32)) and np.array_equal(des.cpu().numpy(), odes)
    want = pl.run_sfm(feats_o, K, images=[oracle.pyrdown(im) for im in images], be=oracle_pipeline_backend(oracle))
    assert out["posearr"].shape == want["posearr"].shape and out["Xtot"].shape == want["Xtot"].shape
    assert np.abs(out["posearr"] - want["posearr"]).max() <= 1e-4 * np.abs(want["posearr"]).max()
    assert np.abs(out["Xtot"] - want["Xtot"]).max() <= 1e-4 * np.abs(want["Xtot"]).max()
    assert np.array_equal(out["colorstot"], want["colorstot"])
    got = out["posearr"][9:].reshape(-1, 3, 4)
    for k in range(n):
        Rg, tg = decompose_P(K, got[k])
        Rp, tp = decompose_P(K, P[k])
        assert np.abs(Rg - Rp).max() < 3e-2 and np.linalg.norm(Rg.T @ tg - Rp.T @ tp) < 0.12      # (five frames of an incremental chain without bundle adjustment)
    assert max(out["errors"]) < 2.0
